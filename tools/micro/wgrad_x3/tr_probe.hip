// What does ds_read_b64_tr_b16 (gfx950) deliver?  LDS holds element index i at bf16 position i; every lane reads 8 bytes at byte offset 8 * lane.
// hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o bin/tr_probe && bin/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; (void)hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h[4 * l + j] != (l & 15) + 16 * j + 64 * (l >> 4)) ++bad;
    for (int l = 0; l < 64; l += 5) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    printf("%d of 256 differ from lds[(l & 15) + 16 j + 64 (l >> 4)]\n", bad);
    return 0;
}
