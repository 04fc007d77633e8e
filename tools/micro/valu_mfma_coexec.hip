// Can the fp32 VALU run a second fp32 GEMM stream beside the matrix pipe?  (dev probe, gfx950)
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) issue dependent v_mfma_f32_16x16x4_f32 chains (2 accumulators, like the
// tile engine at MT = 1), waves 4-7 (their SIMD neighbours) issue v_fmac_f32_dpp row_newbcast chains (8 accumulators: a 16-row x 32-column
// tile of the same GEMM computed as k-ascending FMA chains on the VALU).  Modes: 1 = MFMA waves only, 2 = VALU waves only, 3 = both.
// Prints the time per mode: if mode 3 ~ max(mode 1, mode 2) the two pipes overlap.
// hipcc --offload-arch=gfx950 -O3 valu_mfma_coexec.hip -o valu_mfma_coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FMAC_DPP(acc, a, w, r) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #r " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(w))

typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MF: 0 = v_mfma_f32_16x16x4_f32, 1 = v_mfma_f32_16x16x32_bf16 (same 8 passes per instruction);  VA: 0 = v_fmac_f32_dpp, 1 = v_fma_f32, 2 = v_pk_fma_f32
template <int MF, int VA>
__global__ __launch_bounds__(512) void k_coexec(int mode, int iters, const float* __restrict__ src, float* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float r = 0.0f;
    if (w < 4) {
        if (mode & 1) {
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            float a = src[lane], b0 = src[64 + lane], b1 = src[128 + lane];
            bf8 ab, bb0, bb1;
            for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3c00 + lane + i); bb0[i] = (short)(0x3d00 + lane); bb1[i] = (short)(0x3b00 + i); }
            for (int it = 0; it < iters; ++it) {                 // one iteration = one 16-wide k-block of two column tiles: 8 MFMAs
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (MF == 0) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb1, acc1, 0, 0, 0);
                    }
                }
            }
            r = acc0[0] + acc1[1];
        }
    } else if (mode & 2) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float a[4], wv[4];
        for (int i = 0; i < 4; ++i) { a[i] = src[192 + 4 * lane + i]; wv[i] = src[512 + 4 * lane + i]; }
        f2 pacc[8];
        for (int i = 0; i < 8; ++i) pacc[i] = f2{0.f, 0.f};
        for (int it = 0; it < iters; ++it) {                     // one iteration = the same k-block: 16 k x 8 rows per lane = 128 FMA instructions
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (VA == 0) {
                        FMAC_DPP(acc[0], a[kk], wv[k4], 0); FMAC_DPP(acc[1], a[kk], wv[k4], 1); FMAC_DPP(acc[2], a[kk], wv[k4], 2); FMAC_DPP(acc[3], a[kk], wv[k4], 3);
                        FMAC_DPP(acc[4], a[kk], wv[k4], 4); FMAC_DPP(acc[5], a[kk], wv[k4], 5); FMAC_DPP(acc[6], a[kk], wv[k4], 6); FMAC_DPP(acc[7], a[kk], wv[k4], 7);
                    } else if (VA == 1) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a[kk]), "v"(wv[k4]));
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(f2{a[kk], a[(kk + 1) & 3]}), "v"(f2{wv[k4], wv[(k4 + 1) & 3]}));
                    }
                }
        }
        for (int i = 0; i < 8; ++i) r += pacc[i].x + pacc[i].y;
        for (int i = 0; i < 8; ++i) r += acc[i];
    }
    if (r == 12345.678f) out[blockIdx.x * 512 + tid] = r;
}

int main() {
    float* src; float* out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)(rand() % 1000) * 1e-6f;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    typedef void (*kern_t)(int, int, const float*, float*);
    kern_t kerns[6] = {k_coexec<0, 0>, k_coexec<0, 1>, k_coexec<0, 2>, k_coexec<1, 0>, k_coexec<1, 1>, k_coexec<1, 2>};
    const char* names[6] = {"f32 MFMA + v_fmac_dpp", "f32 MFMA + v_fmac", "f32 MFMA + v_pk_fma", "bf16 MFMA + v_fmac_dpp", "bf16 MFMA + v_fmac", "bf16 MFMA + v_pk_fma"};
    for (int kk = 0; kk < 6; ++kk)
        for (int mode = 1; mode <= 3; ++mode) {
            if (mode == 1) printf("-- %s\n", names[kk]);
            hipLaunchKernelGGL(kerns[kk], dim3(256), dim3(512), 0, 0, mode, iters, src, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kerns[kk], dim3(256), dim3(512), 0, 0, mode, iters, src, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per SIMD per iteration: 8 MFMAs x 32 cycles = 256 cycles; 128 VALU FMAs x 2 cycles = 256 cycles
            const double cyc = ms * 1e-3 * 2.4e9 / iters;
            const double fl_m = (mode & 1) ? 256.0 * 4 * iters * 8 * 2048.0 : 0, fl_v = (mode & 2) ? 256.0 * 4 * iters * 128 * 64 * 2.0 : 0;
            printf("mode %d (%s): %.3f ms  %.0f cycles per k-block @2.4GHz\n", mode, mode == 1 ? "8 MFMAs per wave, waves 0-3" : mode == 2 ? "128 VALU FMAs per wave, waves 4-7" : "both", ms, cyc);
            (void)fl_m; (void)fl_v;
        }
    return 0;
}
