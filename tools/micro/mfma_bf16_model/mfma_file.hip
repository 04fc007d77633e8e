// v_mfma_f32_16x16x32_bf16 on tiles read from files: <prefix>_A.bin / _B.bin (uint16 [T][16][32]), _C.bin (float [T][16][16]) -> <prefix>_D.bin
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_file mfma_file.hip ; run: ./mfma_file <prefix>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__global__ void k_mfma(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, const float* __restrict__ C, float* __restrict__ D) {
    const int lane = threadIdx.x, t = blockIdx.x, r = lane & 15, q = lane >> 4;
    const uint4 a = *(const uint4*)(A + ((size_t)t * 16 + r) * 32 + 8 * q);
    const uint4 b = *(const uint4*)(B + ((size_t)t * 16 + r) * 32 + 8 * q);
    f32x4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[((size_t)t * 16 + 4 * q + i) * 16 + r];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((size_t)t * 16 + 4 * q + i) * 16 + r] = c[i];
}
template <class T> static std::vector<T> rd(const std::string& n) { FILE* f = fopen(n.c_str(), "rb"); if (!f) { printf("cannot open %s\n", n.c_str()); exit(1); } fseek(f, 0, SEEK_END); long s = ftell(f); fseek(f, 0, SEEK_SET); std::vector<T> v(s / sizeof(T)); if (fread(v.data(), 1, s, f) != (size_t)s) exit(1); fclose(f); return v; }
int main(int argc, char** argv) {
    const std::string pre = argv[1];
    auto A = rd<uint16_t>(pre + "_A.bin"), B = rd<uint16_t>(pre + "_B.bin"); auto C = rd<float>(pre + "_C.bin");
    const int T = (int)(C.size() / 256);
    std::vector<float> D(C.size());
    uint16_t *dA, *dB; float *dC, *dD;
    (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, C.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(T), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    FILE* f = fopen((pre + "_D.bin").c_str(), "wb"); fwrite(D.data(), 4, D.size(), f); fclose(f);
    printf("%s: %d tiles\n", pre.c_str(), T);
    return 0;
}
