// Probe of v_mfma_f32_16x16x32_bf16 on gfx950: (1) lane -> element mapping of A / B / D, (2) how the 32 products and C are accumulated
// (which CPU model reproduces the result bit for bit).  Standalone: hipcc --offload-arch=gfx950 -O2 mfma_bf16_probe.hip -o mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_probe(const uint16_t* A /*[T][16][32]*/, const uint16_t* B /*[T][16][32] (col, k)*/, const float* C /*[T][16][16]*/, float* D, int T) {
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < T; t += gridDim.x) {
        bf8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (short)A[(t * 16 + (lane & 15)) * 32 + 8 * (lane >> 4) + i];
            b[i] = (short)B[(t * 16 + (lane & 15)) * 32 + 8 * (lane >> 4) + i];
        }
        f4 c;
        for (int i = 0; i < 4; ++i) c[i] = C[(t * 16 + 4 * (lane >> 4) + i) * 16 + (lane & 15)];
        f4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int i = 0; i < 4; ++i) D[(t * 16 + 4 * (lane >> 4) + i) * 16 + (lane & 15)] = d[i];
    }
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static double urand() { return rand() / (double)RAND_MAX; }

int main() {
    const int T = 4096;
    uint16_t* A = (uint16_t*)malloc(T * 512 * 2); uint16_t* B = (uint16_t*)malloc(T * 512 * 2);
    float* C = (float*)malloc(T * 256 * 4); float* D = (float*)malloc(T * 256 * 4);
    srand(1);
    for (int t = 0; t < T; ++t) {
        const int mode = t % 4;     // 0: small integers (layout), 1: same-scale values, 2: wide exponent spread, 3: cancellation
        for (int i = 0; i < 512; ++i) {
            float a, b;
            if (mode == 0) { a = (float)(rand() % 7 - 3); b = (float)(rand() % 5 - 2); }
            else if (mode == 1) { a = (float)(urand() * 2 - 1); b = (float)(urand() * 2 - 1); }
            else if (mode == 2) { a = (float)((urand() * 2 - 1) * exp2(rand() % 24 - 12)); b = (float)((urand() * 2 - 1) * exp2(rand() % 24 - 12)); }
            else { a = (float)(urand() * 2 - 1); b = (i & 1) ? 1.0f : -1.0f; }
            A[t * 512 + i] = f2bf(a); B[t * 512 + i] = f2bf(b);
        }
        if (mode == 3) for (int i = 0; i < 512; i += 2) A[t * 512 + i + 1] = A[t * 512 + i] ^ (rand() % 3 == 0 ? 1 : 0);   // near-cancelling pairs
        for (int i = 0; i < 256; ++i) C[t * 256 + i] = mode == 0 ? (float)(rand() % 9 - 4) : (float)((urand() * 2 - 1) * (mode == 2 ? exp2(rand() % 20 - 10) : 1.0));
    }
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, T * 512 * 2); hipMalloc(&dB, T * 512 * 2); hipMalloc(&dC, T * 256 * 4); hipMalloc(&dD, T * 256 * 4);
    hipMemcpy(dA, A, T * 512 * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B, T * 512 * 2, hipMemcpyHostToDevice); hipMemcpy(dC, C, T * 256 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, T);
    hipMemcpy(D, dD, T * 256 * 4, hipMemcpyDeviceToHost);
    // models
    const char* names[] = {"seq fmaf k=0..31 from C", "exact sum (float128) + C, one rounding", "4 groups of 8: exact group sums, then C + g0 + g1 + g2 + g3 sequential fp32",
                           "exact sum of products rounded to fp32, then + C", "groups of 8 exact, fp32 add chain g0+g1+g2+g3 then + C", "groups of 4: C + exact g0 .. g7 sequential",
                           "groups of 16: C + exact g0 + g1", "seq fmaf from 0, then + C", "truncated (toward zero) exact sum + C"};
    const int NM = 9;
    long bad[4][NM]; memset(bad, 0, sizeof bad);
    long cnt[4] = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t) for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
        const int mode = t % 4;
        float p[32];
        for (int k = 0; k < 32; ++k) p[k] = bf2f(A[(t * 16 + r) * 32 + k]) * bf2f(B[(t * 16 + c) * 32 + k]);   // exact in fp32
        const float c0 = C[(t * 16 + r) * 16 + c], d = D[(t * 16 + r) * 16 + c];
        float m[NM];
        float s = c0; for (int k = 0; k < 32; ++k) s = fmaf(bf2f(A[(t * 16 + r) * 32 + k]), bf2f(B[(t * 16 + c) * 32 + k]), s); m[0] = s;
        __float128 q = c0; for (int k = 0; k < 32; ++k) q += (__float128)p[k]; m[1] = (float)q;
        __float128 g[8]; float gs[4];
        for (int j = 0; j < 4; ++j) { __float128 qq = 0; for (int k = 0; k < 8; ++k) qq += (__float128)p[8 * j + k]; gs[j] = (float)qq; }
        m[2] = (((c0 + gs[0]) + gs[1]) + gs[2]) + gs[3];
        q = 0; for (int k = 0; k < 32; ++k) q += (__float128)p[k]; m[3] = (float)q + c0;
        m[4] = (((gs[0] + gs[1]) + gs[2]) + gs[3]) + c0;
        s = c0; for (int j = 0; j < 8; ++j) { __float128 qq = 0; for (int k = 0; k < 4; ++k) qq += (__float128)p[4 * j + k]; s = s + (float)qq; } m[5] = s;
        s = c0; for (int j = 0; j < 2; ++j) { __float128 qq = 0; for (int k = 0; k < 16; ++k) qq += (__float128)p[16 * j + k]; s = s + (float)qq; } m[6] = s;
        s = 0; for (int k = 0; k < 32; ++k) s = fmaf(bf2f(A[(t * 16 + r) * 32 + k]), bf2f(B[(t * 16 + c) * 32 + k]), s); m[7] = s + c0;
        q = c0; for (int k = 0; k < 32; ++k) q += (__float128)p[k];
        { float f = (float)q; if ((__float128)f != q && fabs((double)(__float128)f) > fabs((double)q)) f = nextafterf(f, 0.0f); m[8] = f; }
        (void)g;
        cnt[mode]++;
        for (int i = 0; i < NM; ++i) if (memcmp(&m[i], &d, 4) != 0) bad[mode][i]++;
    }
    const char* mn[] = {"small integers (layout check)", "uniform values", "wide exponent spread", "cancellation"};
    for (int mode = 0; mode < 4; ++mode) {
        printf("mode %d  %s: %ld results\n", mode, mn[mode], cnt[mode]);
        for (int i = 0; i < NM; ++i) printf("    model %d (%s): %ld mismatches\n", i, names[i], bad[mode][i]);
    }
    return 0;
}
