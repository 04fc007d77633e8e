// Micro-benchmark: sustained fp32 MFMA rate of v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32 with operands in registers,
// NACC independent accumulators per wave, 8 waves per workgroup, G workgroups per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flops_per_wave_iter, int iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops_per_wave_iter * iters * blocks * 8 / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 512 * 4);
    const int iters = 20000;
    for (int g = 1; g <= 2; ++g) {
        const int blocks = 256 * g;
        printf("-- %d workgroup(s) of 8 waves per CU\n", g);
        run("16x16x4, 1 acc", [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(512), 0, 0, out, iters); }, 2048.0 * 1, iters, blocks);
        run("16x16x4, 2 acc", [&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(512), 0, 0, out, iters); }, 2048.0 * 2, iters, blocks);
        run("16x16x4, 4 acc", [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(512), 0, 0, out, iters); }, 2048.0 * 4, iters, blocks);
        run("32x32x2, 1 acc", [&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(512), 0, 0, out, iters); }, 4096.0 * 1, iters, blocks);
        run("32x32x2, 2 acc", [&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(512), 0, 0, out, iters); }, 4096.0 * 2, iters, blocks);
    }
    return 0;
}
