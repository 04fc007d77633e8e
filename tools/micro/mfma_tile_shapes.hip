// Micro-benchmark (round 4, review item 2): ONE layer phase of the fused fp32 tracing MLP with its real operand paths -- activations from LDS, packed
// weights from L2, 8 waves per workgroup, 32 rows x 256 -> 256 -- in the engine's shape (2 x 2 register-blocked v_mfma_f32_16x16x4_f32: tile_engine.h with
// MT = 2, NTW = 2) against one v_mfma_f32_32x32x2_f32 tile per wave.  Answers two questions:
//   * is the 32x32x2 accumulation the same k-ascending fmaf chain?  (outputs of the two kernels compared bit for bit, and against a host fmaf loop)
//   * does the 32x32 tile feed more FLOPs per operand load?  No: a wave that owns 32 rows x 32 columns needs 2 LDS reads (ds_read_b128) and 2 weight loads
//     (16 bytes) per 16-wide k-block either way -- the 2 x 2 blocking of 16x16 tiles IS a 32 x 32 tile; only the instruction count halves (8 x 64 cycles instead of
//     16 x 32), and instruction issue is not what bounds the fp32 engine (VALU : MFMA 3.5 : 1, the softplus shares the matrix datapath).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_tile_shapes mfma_tile_shapes.hip ; run: ./mfma_tile_shapes
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int K = 256, N = 256, ROWS = 32, S = 264;          // LDS row stride (floats): 8 mod 64
__device__ __host__ inline int perm16(int c) { return (c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3); }      // tile_engine.h's 4x4 transpose inside a 16-block
__device__ __host__ inline int perm32(int c) { return (c & ~15) | ((c & 1) << 3) | ((c >> 1) & 7); }      // [even k's | odd k's] inside a 16-block

// (a) the engine's shape: wave w owns columns 32 w .. 32 w + 31 as two 16-column tiles, rows as two 16-row tiles
__global__ __launch_bounds__(512) void k_shape16(const float* __restrict__ x, const float4* __restrict__ wp, float* __restrict__ y, int layers) {
    __shared__ __attribute__((aligned(16))) float act[ROWS * S];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    for (int i = tid; i < ROWS * K; i += 512) act[(i / K) * S + perm16(i % K)] = x[(size_t)blockIdx.x * ROWS * K + i];
    for (int l = 0; l < layers; ++l) {
        __syncthreads();
        f32x4 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int t = 0; t < 2; ++t) acc[a][t] = f32x4{0, 0, 0, 0};
        const float4* wt = wp + (size_t)(2 * w) * (K / 16) * 64 + lane;
        for (int kb = 0; kb < K / 16; ++kb) {
            const float4 a0 = *(const float4*)(act + r * S + kb * 16 + 4 * q), a1 = *(const float4*)(act + (16 + r) * S + kb * 16 + 4 * q);
            const float4 b0 = wt[(size_t)kb * 64], b1 = wt[((size_t)(K / 16) + kb) * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a0)[s], ((const float*)&b0)[s], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a1)[s], ((const float*)&b0)[s], acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a0)[s], ((const float*)&b1)[s], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a1)[s], ((const float*)&b1)[s], acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
        // trivial epilogue (the real one: bias + softplus): lane holds rows 4q .. 4q+3 of column 16 t + r
        for (int a = 0; a < 2; ++a) for (int t = 0; t < 2; ++t) for (int i = 0; i < 4; ++i) {
            const int row = 16 * a + 4 * q + i, col = 32 * w + 16 * t + r;
            const float v = acc[a][t][i] * 0.0625f;
            if (l == layers - 1) y[((size_t)blockIdx.x * ROWS + row) * N + col] = acc[a][t][i];
            act[row * S + perm16(col)] = v;
        }
    }
}

// (b) one 32 x 32 tile per wave: A = activations (32 rows x 2 k), B = weights (2 k x 32 columns)
__global__ __launch_bounds__(512) void k_shape32(const float* __restrict__ x, const float4* __restrict__ wp32, float* __restrict__ y, int layers) {
    __shared__ __attribute__((aligned(16))) float act[ROWS * S];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, h = lane >> 5;
    for (int i = tid; i < ROWS * K; i += 512) act[(i / K) * S + perm32(i % K)] = x[(size_t)blockIdx.x * ROWS * K + i];
    for (int l = 0; l < layers; ++l) {
        __syncthreads();
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float4* wt = wp32 + (size_t)w * (K / 16) * 128 + lane;
        for (int kb = 0; kb < K / 16; ++kb) {
            const float4 a0 = *(const float4*)(act + n * S + kb * 16 + 8 * h), a1 = *(const float4*)(act + n * S + kb * 16 + 8 * h + 4);
            const float4 b0 = wt[(size_t)kb * 128], b1 = wt[(size_t)kb * 128 + 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(((const float*)&a0)[j], ((const float*)&b0)[j], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(((const float*)&a1)[j], ((const float*)&b1)[j], acc, 0, 0, 0);
        }
        __syncthreads();
        for (int i = 0; i < 16; ++i) {                            // C: column n, row (i & 3) + 8 (i >> 2) + 4 h
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h, col = 32 * w + n;
            const float v = acc[i] * 0.0625f;
            if (l == layers - 1) y[((size_t)blockIdx.x * ROWS + row) * N + col] = acc[i];
            act[row * S + perm32(col)] = v;
        }
    }
}

int main() {
    const int blocks = 512, layers = 8, reps = 50;
    std::vector<float> W((size_t)N * K), X((size_t)blocks * ROWS * K);
    srand(1);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    for (auto& v : X) v = rand() / (float)RAND_MAX - 0.5f;
    // packs: (a) Wp[ct][kb][lane][s] = W[16 ct + (lane & 15)][16 kb + 4 s + (lane >> 4)]; (b) Wp32[ct][kb][half][lane][j] = W[32 ct + (lane & 31)][16 kb + 2 (4 half + j) + (lane >> 5)]
    std::vector<float> P16((size_t)N * K), P32((size_t)N * K);
    for (int ct = 0; ct < N / 16; ++ct) for (int kb = 0; kb < K / 16; ++kb) for (int lane = 0; lane < 64; ++lane) for (int s = 0; s < 4; ++s)
        P16[(((size_t)ct * (K / 16) + kb) * 64 + lane) * 4 + s] = W[(size_t)(16 * ct + (lane & 15)) * K + 16 * kb + 4 * s + (lane >> 4)];
    for (int ct = 0; ct < N / 32; ++ct) for (int kb = 0; kb < K / 16; ++kb) for (int hf = 0; hf < 2; ++hf) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j)
        P32[((((size_t)ct * (K / 16) + kb) * 2 + hf) * 64 + lane) * 4 + j] = W[(size_t)(32 * ct + (lane & 31)) * K + 16 * kb + 2 * (4 * hf + j) + (lane >> 5)];
    float *dx, *dp16, *dp32, *dy16, *dy32;
    hipMalloc(&dx, X.size() * 4); hipMalloc(&dp16, P16.size() * 4); hipMalloc(&dp32, P32.size() * 4);
    hipMalloc(&dy16, (size_t)blocks * ROWS * N * 4); hipMalloc(&dy32, (size_t)blocks * ROWS * N * 4);
    hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dp16, P16.data(), P16.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp32, P32.data(), P32.size() * 4, hipMemcpyHostToDevice);
    // ---- one layer: bit-equality of the two shapes and of a host fmaf chain
    hipLaunchKernelGGL(k_shape16, dim3(blocks), dim3(512), 0, 0, dx, (const float4*)dp16, dy16, 1);
    hipLaunchKernelGGL(k_shape32, dim3(blocks), dim3(512), 0, 0, dx, (const float4*)dp32, dy32, 1);
    std::vector<float> Y16((size_t)blocks * ROWS * N), Y32(Y16.size());
    hipMemcpy(Y16.data(), dy16, Y16.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(Y32.data(), dy32, Y32.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0, diffh = 0;
    for (size_t i = 0; i < Y16.size(); ++i) diff += Y16[i] != Y32[i];
    for (int row = 0; row < 64; ++row) for (int col = 0; col < N; ++col) {
        float a = 0.f;
        for (int k = 0; k < K; ++k) a = fmaf(X[(size_t)row * K + k], W[(size_t)col * K + k], a);
        diffh += a != Y32[(size_t)row * N + col];
    }
    printf("one layer, %zu outputs: 16x16x4 (2x2) vs 32x32x2 differ in %zu; 32x32x2 vs host k-ascending fmaf chain (64 rows) differ in %zu\n", Y16.size(), diff, diffh);
    // ---- throughput of the layer phase
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        float ms = 0;
        for (int warm = 0; warm < 2; ++warm) {
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) {
                if (v == 0) hipLaunchKernelGGL(k_shape16, dim3(blocks), dim3(512), 0, 0, dx, (const float4*)dp16, dy16, layers);
                else hipLaunchKernelGGL(k_shape32, dim3(blocks), dim3(512), 0, 0, dx, (const float4*)dp32, dy32, layers);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flop = 2.0 * ROWS * K * N * layers * blocks * reps;
        printf("%s: %.3f ms per launch (%d workgroups x %d rows x %d layers), %.1f TFLOP/s, %.2f us per 32-row layer phase and CU\n",
               v == 0 ? "2 x 2 blocked 16x16x4 (engine)" : "one 32x32x2 tile per wave     ", ms / reps, blocks, ROWS, layers, flop / (ms * 1e-3) / 1e12,
               ms / reps * 1e3 / layers / (blocks / 256.0));
    }
    return 0;
}
