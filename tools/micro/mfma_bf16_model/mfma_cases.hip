// Targeted cases for v_mfma_f32_16x16x32_bf16: only D[0][0] of each tile is read; products p_k = a_k * 1 with a_k powers of two (bf16-exact).
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_cases mfma_cases.hip ; run: ./mfma_cases
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
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
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }     // exact for the values used here
struct Case { float c; float p[32]; const char* what; int j; };
int main() {
    std::vector<Case> cs;
    auto add = [&](float c, std::initializer_list<std::pair<int, float>> ps, const char* what, int j) { Case x; x.c = c; for (auto& v : x.p) v = 0; for (auto& kv : ps) x.p[kv.first] = kv.second; x.what = what; x.j = j; cs.push_back(x); };
    for (int j = 0; j <= 48; ++j) {
        const float s = ldexpf(1.0f, -j);
        add(0.0f, {{0, 16777216.0f}, {1, -16777216.0f}, {2, s}}, "A  c=0, p0=2^24, p1=-2^24, p2=2^-j (same lane group): expect 2^-j if kept", j);
        add(0.0f, {{0, 16777216.0f}, {1, -16777216.0f}, {8, s}}, "B  ... p8=2^-j (next lane group)", j);
        add(0.0f, {{0, 16777216.0f}, {8, -16777216.0f}, {2, s}}, "C  p0=2^24, p8=-2^24 (next group), p2=2^-j in group 0", j);
        add(16777216.0f, {{0, -16777216.0f}, {1, s}}, "D  c=2^24, p0=-2^24, p1=2^-j: accumulator in the same alignment?", j);
        add(1.0f, {{0, ldexpf(1.0f, -24)}, {1, s}}, "E  c=1, p0=2^-24 (tie), p1=2^-j: sticky?  1+2^-23 if seen", j);
        add(1.0f, {{0, ldexpf(1.0f, -24)}, {1, -s}}, "F  c=1, p0=2^-24, p1=-2^-j: 1 if seen (below tie)", j);
        add(1.0f, {{0, s}}, "G  c=1, p0=2^-j alone", j);
        add(-1.0f, {{0, ldexpf(1.0f, -24)}, {1, s}}, "H  c=-1, p0=2^-24, p1=2^-j: magnitude truncation or floor?", j);
        add(0.0f, {{0, 1.0f}, {1, ldexpf(1.0f, -24)}, {2, s}}, "I  c=0, p0=1, p1=2^-24, p2=2^-j", j);
        add(0.0f, {{0, 1.0f}, {1, ldexpf(1.0f, -24)}, {8, s}}, "J  c=0, p0=1, p1=2^-24, p8=2^-j (next group)", j);
    }
    const int T = (int)cs.size();
    std::vector<uint16_t> A((size_t)T * 16 * 32, 0), B(A.size(), 0);
    std::vector<float> C((size_t)T * 256, 0.0f), D(C.size());
    for (int t = 0; t < T; ++t) {
        for (int k = 0; k < 32; ++k) { A[((size_t)t * 16) * 32 + k] = f2bf(cs[t].p[k]); B[((size_t)t * 16) * 32 + k] = f2bf(1.0f); }
        C[(size_t)t * 256] = cs[t].c;
    }
    uint16_t *dA, *dB; float *dC, *dD;
    (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, C.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(T), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    const char* last = "";
    for (int t = 0; t < T; ++t) {
        // group the output per case family
        (void)last;
    }
    for (int fam = 0; fam < 10; ++fam) {
        printf("%s\n   j:result ", cs[fam].what);
        for (int t = fam; t < T; t += 10) {
            const float d = D[(size_t)t * 256];
            double ex = cs[t].c; for (int k = 0; k < 32; ++k) ex += cs[t].p[k];
            printf(" %d:%s", cs[t].j, d == (float)ex ? "=" : "x");
            if (d != (float)ex) printf("(%.9g)", d);
        }
        printf("\n");
    }
    return 0;
}
