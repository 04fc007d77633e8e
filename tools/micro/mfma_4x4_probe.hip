// Probe of v_mfma_f32_4x4x1_16B_f32 on gfx950 (dev tool): 16 independent 4x4 outer products per instruction, K = 1.
//  (1) operand / result layout: which lane feeds which row / column of which block, where the results land;
//  (2) a k-loop of K = 1 MFMAs as an fmaf chain: bit-exact against the host's fmaf loop?
//  (3) issue rate of dependent chains with 1 / 2 / 4 accumulators per wave, one wave per SIMD and two.
// hipcc --offload-arch=gfx950 -O3 mfma_4x4_probe.hip -o mfma_4x4_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[v * 64 + l] = c[v];
}

// rows 0..3 (A, broadcast to every block), 64 columns (B, lane = column), K steps: D[row][col] = fma chain over k
__global__ void k_chain(const float* A /*[4][K]*/, const float* B /*[K][64]*/, int K, float* D /*[4][64]*/) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    for (int k = 0; k < K; ++k) c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[(l & 3) * K + k], B[k * 64 + l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[v * 64 + l] = c[v];
}

template <int NACC>
__global__ __launch_bounds__(512) void k_rate(int iters, int waves, const float* src, float* out) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w >= waves) return;
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0, 0, 0, 0};
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[l + 64 * i]; b[i] = src[256 + l + 64 * i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s], b[(s + i) & 3], c[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += c[i][0] + c[i][3];
    if (r == 123.456f) out[threadIdx.x] = r;
}

int main() {
    float ha[64], hb[64], hd[256];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i; hb[i] = 100.0f * (i + 1); }
    float *a, *b, *d;
    hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMalloc(&d, 1 << 20);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    // hypothesis: D[v][l] = A[lane 4*(l/4) + v] * B[lane l]
    int bad = 0;
    for (int v = 0; v < 4; ++v) for (int l = 0; l < 64; ++l) if (hd[v * 64 + l] != ha[(l & ~3) + v] * hb[l]) ++bad;
    printf("layout D[v][lane] == A[4*(lane/4)+v] * B[lane]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hd[l], hd[64 + l], hd[128 + l], hd[192 + l]);
    // chain exactness
    const int K = 295;
    float* hA = (float*)malloc(4 * K * 4); float* hB = (float*)malloc(K * 64 * 4); float hD[256];
    srand(3);
    for (int i = 0; i < 4 * K; ++i) hA[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * exp2(rand() % 12 - 6));
    for (int i = 0; i < K * 64; ++i) hB[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * exp2(rand() % 12 - 6));
    hipMemcpy(a, hA, 4 * K * 4, hipMemcpyHostToDevice); hipMemcpy(b, hB, K * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, a, b, K, d);
    hipMemcpy(hD, d, sizeof hD, hipMemcpyDeviceToHost);
    bad = 0;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 64; ++c) {
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = fmaf(hA[r * K + k], hB[k * 64 + c], acc);
        if (memcmp(&acc, &hD[r * 64 + c], 4)) ++bad;
    }
    printf("K = %d chain of 4x4x1 MFMAs == host fmaf chain, bit for bit: %s (%d of 256 differ)\n", K, bad ? "NO" : "yes", bad);
    // rate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
#define RATE(NACC, WAVES) do { \
        hipLaunchKernelGGL((k_rate<NACC>), dim3(256), dim3(512), 0, 0, iters, WAVES, a, d); hipDeviceSynchronize(); \
        hipEventRecord(e0); hipLaunchKernelGGL((k_rate<NACC>), dim3(256), dim3(512), 0, 0, iters, WAVES, a, d); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); \
        printf("%d accumulator(s), %d wave(s) per workgroup: %.1f cycles per MFMA per wave @2.4GHz (%.1f per SIMD-issue)\n", NACC, WAVES, ms * 1e-3 * 2.4e9 / (iters * 4.0 * NACC), \
               ms * 1e-3 * 2.4e9 / (iters * 4.0 * NACC * (WAVES > 4 ? 2 : 1))); } while (0)
    RATE(1, 4); RATE(2, 4); RATE(4, 4); RATE(1, 8); RATE(2, 8); RATE(4, 8);
    return 0;
}
