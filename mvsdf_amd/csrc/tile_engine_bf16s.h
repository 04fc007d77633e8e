// tile_engine_bf16s.h -- the tracing MLP with bf16 WEIGHTS and fp32-ACCURATE activations on gfx950's bf16 matrix cores
// (BASELINE configs[4], "bf16 MLP weights": the mode that is both fast and parity-checked against the reference arithmetic on
// bf16-rounded weights, idr.py:77-94 inside ray_tracing.py:27-98).
//
// tile_engine_bf16.h rounds the hidden activations to bf16 too (8 mantissa bits: masks 99.95 %, depth p99 1.5e-3 at the c5 share).  Here an
// fp32 activation a is carried as NS bf16 TERMS
//       t0 = bf16(a),  t1 = bf16(a - t0),  [t2 = bf16(a - t0 - t1)]          (every subtraction is exact in fp32)
// i.e. 16 (NS = 2) or all 24 (NS = 3: a == t0 + t1 + t2 exactly) mantissa bits, and a Linear is NS matrix instructions per k-block that SHARE
// one weight fragment:   z = bias + sum_k W16[n][k] * (t0[k] + t1[k] + t2[k]),   v_mfma_f32_16x16x32_bf16, fp32 accumulators.
// A bf16 x bf16 product is exact in fp32, so with NS = 3 the only difference to the fp32 engine on the rounded weights (trace_dtype 2, bit-exact
// vs oracle.Net(sd, bf16='weights')) is the ORDER of the fp32 additions inside the matrix core and the softplus below: accumulation noise of a
// few 1e-8 on |z| ~ 1, no 8-bit rounding anywhere.  The L2 weight stream -- what bounds the bf16 engine -- does not grow: only the LDS activation
// reads and the (16x cheaper than fp32) matrix instructions are multiplied by NS.
//   * weights: bf16 packs of tile_engine_bf16.h WITHOUT duplicated columns (nsplit = 0: the positional-encoding columns are split like
//     every other activation, into the term tiles);
//   * LDS: NS term tiles [rows][S16] bf16 behind each other (term stride rows * S16), row stride 64*KB + 16 bytes (conflict-free b128 reads);
//   * softplus: max(z, 0) + (ln 2 / 100) log2(1 + 2^(-100 log2(e) |z|)) by v_exp_f32 / v_log_f32: absolute error <= 8e-9, the accuracy class of
//     det_math's at a quarter of its instructions (mv_softplus100_acc1 below);
//   * everything else (bias as the accumulator's start value, transposed accumulators, the two weight-fetch schemes) as in tile_engine_bf16.h.
//
// Measured against an fp64 evaluation the three-term engine is CLOSER than the fp32 fmaf chain (the matrix instruction adds eight exact products per
// rounding: tools/micro/mfma_bf16_model/), so the same engine also runs the reference's fp32 arithmetic on UNROUNDED weights: MvNetBs<3, 3>
// (trace_dtype 5, "f32x3") splits the fp32 weights into three bf16 terms too and multiplies the six pairs a_s w_j with s + j <= 2
// (mv_gemm_rolling_bw / mv_gemm_carried_bw below; packs by mvsdf_pack_bf16x3_net; tests/test_gpu_f32x3.py).
#pragma once
#include "tile_engine_bf16.h"

template <int NS, int WT = 1>
struct MvNetBs : MvNetBf {};        // S = NS * (32 * KBmax + 8) / 2 floats per row (all term tiles); L[l].nsplit == 0
                                    // WT = 3 (trace_dtype 5): fp32 WEIGHTS as three bf16 terms too (packs of mvsdf_pack_bf16x3_net), see mv_gemm_rolling_bw

__host__ __device__ constexpr int mv_bs_pa(int NS) { return NS == 1 ? 4 : 2; }          // activation k-blocks in flight (LDS), CARRIED
#ifndef MV_BS_PAR3
#define MV_BS_PAR3 1
#endif
__host__ __device__ constexpr int mv_bs_par(int NS) { return NS >= 3 ? MV_BS_PAR3 : mv_bs_pa(NS); }   // ... ROLLING: three terms x two row tiles x two k-blocks of
                                                                                        // A fragments are 48 registers -- with one k-block in flight the 32-row sample kernel
                                                                                        // stays under 128 VGPRs (two workgroups per CU)

// Softplus(beta=100, threshold=20) to fp32 accuracy in six instructions per activation:
//     softplus(100 z) / 100 = max(z, 0) + (ln 2 / 100) log2(1 + t),   t = 2^(-100 log2(e) |z|)      (v_exp_f32, v_log_f32: 1 ulp each)
// Absolute error <= 8e-9 on results up to 0.3 (= the final rounding; the degree-8 polynomial form of log(1 + t) / t measured the same 7.5e-9 and costs
// 13 instructions).  For t below 6e-8 the sum 1 + t rounds to 1 and the result is max(z, 0) exactly: an absolute error below 6e-10, i.e. nothing against
// the fp32 rounding of the next layer's sums -- and 100 z > 20 gives t < 2.1e-9: the reference's threshold branch (result = z) falls out by itself.
// The engines of this family are VALU-issue-bound (PMC: ~20 VALU instructions per activation and matrix instruction, matrix pipe busy 0.17-0.24), so the
// activation's instruction count IS their speed.
__device__ __forceinline__ float mv_softplus100_acc1(float z) {
    const float t = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);
    const float lg = __builtin_amdgcn_logf(1.0f + t);
    return fmaf(lg, 0.0069314718246459961f, __builtin_amdgcn_fmed3f(z, 0.0f, 3.0e38f));
}
__device__ __forceinline__ dm_f2 mv_softplus100_acc2(dm_f2 z) { return dm_f2{mv_softplus100_acc1(z.x), mv_softplus100_acc1(z.y)}; }

// trace_dtype 5 ("f32x3") is reproduced BIT FOR BIT by the CPU oracle (oracle/oracle_mvsdf.c::sdf_row_f32x3 models the matrix instruction: tools/micro/mfma_bf16_model/),
// so its activation is det_math's softplus (IEEE operations only), not the hardware exp / log form above, and every value that enters the matrix core is first
// flushed to zero below 2^-40 (no bf16 term is ever denormal, every product stays above 2^-112: the instruction model is verified on products inside the fp32 normal
// range -- below it the hardware deviates, tools/micro/mfma_bf16_model/mfma_fuzz.hip `tiny`; 1e-12 on values up to 10)
#define MV_X3_FLUSH 9.094947017729282e-13f                          // 2^-40
__device__ __forceinline__ float mv_x3_flush(float v) { return fabsf(v) < MV_X3_FLUSH ? 0.0f : v; }
__device__ __forceinline__ dm_f2 mv_x3_flush2(dm_f2 v) { return dm_f2{mv_x3_flush(v.x), mv_x3_flush(v.y)}; }

__device__ __forceinline__ dm_f2 mv_bf_unpack2(uint32_t p) { return dm_f2{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)}; }

// two activations -> NS packed pairs of bf16 terms
template <int NS>
__device__ __forceinline__ void mv_split_pk(dm_f2 h, uint32_t (&p)[NS]) {
    p[0] = mv_f2bf_pk(h.x, h.y);
#pragma unroll
    for (int s = 1; s < NS; ++s) {
        h = h - mv_bf_unpack2(p[s - 1]);
        p[s] = mv_f2bf_pk(h.x, h.y);
    }
}
template <int NS>
__device__ __forceinline__ void mv_split_1(float v, uint16_t (&p)[NS]) {
    p[0] = mv_f2bf(v);
#pragma unroll
    for (int s = 1; s < NS; ++s) {
        v = v - mv_bf2f(p[s - 1]);
        p[s] = mv_f2bf(v);
    }
}

// positional encoding -> pe[rows][d0] (fp32, kept for the skip connection) and the layer-0 input rows of the NS term tiles (zero padded to kpad)
template <int NTHREADS, int NS, bool FLUSH = false>
__device__ __forceinline__ void mv_pe_rows_bs(const float* pts, float* pe, uint16_t* act, int S16, int TS, int rows, int multires, int kpad, int tid) {
    const int d0 = 3 + 6 * multires, T = 3 * multires + 1;
    for (int task = tid; task < rows * T; task += NTHREADS) {
        const int row = task / T, j = task - row * T;
        const float* x = pts + row * 3;
        float* pr = pe + row * d0;
        uint16_t* ar = act + row * S16;
        auto put = [&](int col, float v) {
            pr[col] = v;
            uint16_t p[NS];
            mv_split_1<NS>(FLUSH ? mv_x3_flush(v) : v, p);
#pragma unroll
            for (int s = 0; s < NS; ++s) ar[s * TS + col] = p[s];
        };
        if (j < 3 * multires) {
            const int m = j / 3, c = j - 3 * m;
            float s, co;
            dm_sincos(x[c] * (float)(1 << m), &s, &co);
            put(3 + 6 * m + c, s);
            put(3 + 6 * m + 3 + c, co);
        } else {
            for (int c = 0; c < 3; ++c) put(c, x[c]);
            for (int c = d0; c < kpad; ++c) {
#pragma unroll
                for (int s = 0; s < NS; ++s) ar[s * TS + c] = 0;
            }
        }
    }
}

// CARRIED weight fetch (k_sphere_trace), see tile_engine_bf16.h::mv_gemm_carried_bf: the same ring, NS matrix instructions per weight fragment
template <int MTc, int NTW, int PD, int PDR, int NS>
__device__ __forceinline__ void mv_gemm_carried_bs(int KB, const uint16_t* __restrict__ act, int S16, int TS, const uint4* const (&wcur)[NTW], int ntw,
                                                   f32x4 (&acc)[MTc][NTW], int lane, uint4 (&b)[PD][NTW], const uint4* const (&wnext)[NTW], int kbnext) {
    constexpr int PA = mv_bs_pa(NS);
    static_assert(PD % PA == 0, "weight register depth must be a multiple of the activation ring depth");
    const uint16_t* arow = act + (lane & 15) * S16 + 8 * (lane >> 4);
    uint4 a[PA][MTc][NS];
#pragma unroll
    for (int d = 0; d < PA; ++d) {
        const int kb = d < KB ? d : KB - 1;
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int s = 0; s < NS; ++s) a[d][r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16 + kb * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < PD; ++kb) {
        if (kb < KB) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    if (t < ntw) {
#pragma unroll
                        for (int r = 0; r < MTc; ++r)
                            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[kb][t]), __builtin_bit_cast(mv_bf8, a[kb % PA][r][s]), acc[r][t], 0, 0, 0);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kb < PDR) {
            const int kn = kb < kbnext ? kb : kbnext - 1;                                    // clamped: no branch around a load
#pragma unroll
            for (int t = 0; t < NTW; ++t) b[kb][t] = wnext[t][kn * 64];
        }
        {
            const int ka = kb + PA < KB ? kb + PA : KB - 1;
#pragma unroll
            for (int r = 0; r < MTc; ++r)
#pragma unroll
                for (int s = 0; s < NS; ++s) a[kb % PA][r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16 + ka * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int kb = PD; kb < KB; ++kb) {
        uint4 bx[NTW], ax[MTc][NS];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bx[t] = wcur[t][kb * 64];
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int s = 0; s < NS; ++s) ax[r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16 + kb * 32);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                if (t < ntw) {
#pragma unroll
                    for (int r = 0; r < MTc; ++r)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, bx[t]), __builtin_bit_cast(mv_bf8, ax[r][s]), acc[r][t], 0, 0, 0);
                }
    }
}

// ROLLING weight fetch (the row-sample kernels), see tile_engine_bf16.h::mv_gemm_rolling_bf
template <int MTc, int NT, int NTW, int PD, int NS>
__device__ __forceinline__ void mv_gemm_rolling_bs(int KB, const uint16_t* __restrict__ act, int S16, int TS, const uint4* __restrict__ wp, f32x4 (&acc)[MTc][NTW], int lane) {
    constexpr int PA = mv_bs_par(NS);
    static_assert(PD % PA == 0, "weight ring depth must be a multiple of the activation ring depth");
    const uint16_t* arow = act + (lane & 15) * S16 + 8 * (lane >> 4);
    uint4 b[PD][NT], a[PA][MTc][NS];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        const int kb = d < KB ? d : KB - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + kb) * 64];
        if (d < PA) {
#pragma unroll
            for (int r = 0; r < MTc; ++r)
#pragma unroll
                for (int s = 0; s < NS; ++s) a[d][r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16 + kb * 32);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            if (kb0 + d < KB) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < MTc; ++r)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[d][t]), __builtin_bit_cast(mv_bf8, a[d % PA][r][s]), acc[r][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kn = (kb0 + d + PD < KB) ? kb0 + d + PD : KB - 1;
            const int ka = (kb0 + d + PA < KB) ? kb0 + d + PA : KB - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + kn) * 64];
#pragma unroll
            for (int r = 0; r < MTc; ++r)
#pragma unroll
                for (int s = 0; s < NS; ++s) a[d % PA][r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16 + ka * 32);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MTc, int NTW, int NS>
__device__ __forceinline__ void mv_gemm_rolling_dispatch_bs(int KB, const uint16_t* act, int S16, int TS, const uint4* wp, int ntw, f32x4 (&acc)[MTc][NTW], int lane) {
    if (ntw == NTW) { mv_gemm_rolling_bs<MTc, NTW, NTW, 4, NS>(KB, act, S16, TS, wp, acc, lane); return; }
    if (NTW >= 4 && ntw == 3) { mv_gemm_rolling_bs<MTc, (NTW >= 4 ? 3 : 1), NTW, 4, NS>(KB, act, S16, TS, wp, acc, lane); return; }
    if (NTW >= 2 && ntw == 2) { mv_gemm_rolling_bs<MTc, (NTW >= 2 ? 2 : 1), NTW, 4, NS>(KB, act, S16, TS, wp, acc, lane); return; }
    if (ntw == 1) { mv_gemm_rolling_bs<MTc, 1, NTW, 4, NS>(KB, act, S16, TS, wp, acc, lane); return; }
    for (int t0 = 0; t0 < ntw; ++t0) {                              // 5..NTW-1 tiles (wide nets only): one by one
        f32x4 tmp[MTc][NTW];
#pragma unroll
        for (int r = 0; r < MTc; ++r) tmp[r][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mv_gemm_rolling_bs<MTc, 1, NTW, 4, NS>(KB, act, S16, TS, wp + (size_t)t0 * KB * 64, tmp, lane);
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int u = 0; u < NTW; ++u) if (u == t0) acc[r][u] += tmp[r][0];
    }
}

// ---- trace_dtype 5 ("f32x3"): fp32 weights AND fp32 activations as three bf16 terms each ----
// w = w0 + w1 + w2 and a = a0 + a1 + a2 exactly, every bf16 x bf16 product is exact in fp32, so
//     a w = sum over s + j <= 2 of a_s w_j  +  (a1 w2 + a2 w1 + a2 w2 <= 3 * 2^-24 |a w|):
// the fp32 Linear of the reference (idr.py:89) from SIX matrix instructions of 16 cycles per 32-wide k-block instead of eight of 32 cycles, accumulated by
// the matrix core (measured against an fp64 evaluation the three-term engine is CLOSER than the k-ascending fp32 fmaf chain: rms 2.0e-7 vs 4.2e-7 on
// |sdf| ~ 1, tools/micro/f32s/acc_probe.py).  Pack layout: wp[((ct * KB + kb) * WT + j) * 64 + lane] (a wave's WT fragments of a k-block are adjacent).
// ROLLING fetch only (a carried ring would hold WT x the registers); DEEP (k_sphere_trace, one workgroup per CU): four k-blocks of weights in flight, else two.
template <int MTc, int NT, int NTW, int PD, int NS, int WT>
__device__ __forceinline__ void mv_gemm_rolling_bw(int KB, const uint16_t* __restrict__ act, int S16, int TS, const uint4* __restrict__ wp, f32x4 (&acc)[MTc][NTW], int lane) {
    constexpr int TOP = (NS > WT ? NS : WT) - 1;                    // products a_s w_j with s + j <= TOP
    const uint16_t* arow = act + (lane & 15) * S16 + 8 * (lane >> 4);
    uint4 b[PD][NT][WT], a[MTc][NS];
    const uint4* wt[NT];                                            // column tile t of this wave: k-block kb's term j at wt[t][(kb * WT + j) * 64] (j * 1 KiB: an immediate offset)
#pragma unroll
    for (int t = 0; t < NT; ++t) wt[t] = wp + (size_t)t * KB * WT * 64;
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        const int kb = d < KB ? d : KB - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < WT; ++j) b[d][t][j] = wt[t][(kb * WT + j) * 64];
    }
#pragma unroll
    for (int r = 0; r < MTc; ++r)
#pragma unroll
        for (int s = 0; s < NS; ++s) a[r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16);
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            if (kb0 + d < KB) {
#pragma unroll
                for (int o = TOP; o >= 0; --o)                       // smallest products first
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int j = o - s;
                        if (j >= 0 && j < WT) {
#pragma unroll
                            for (int r = 0; r < MTc; ++r)
#pragma unroll
                                for (int t = 0; t < NT; ++t)
                                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[d][t][j]), __builtin_bit_cast(mv_bf8, a[r][s]), acc[r][t], 0, 0, 0);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kn = (kb0 + d + PD < KB) ? kb0 + d + PD : KB - 1;
            const int ka = (kb0 + d + 1 < KB) ? kb0 + d + 1 : KB - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < WT; ++j) b[d][t][j] = wt[t][(kn * WT + j) * 64];
#pragma unroll
            for (int r = 0; r < MTc; ++r)
#pragma unroll
                for (int s = 0; s < NS; ++s) a[r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16 + ka * 32);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// CARRIED form of the above (k_sphere_trace: its evaluations wait for each other, so the L2 latency of a layer's first weight fragments is exposed 9 times per
// evaluation).  `b` arrives holding this layer's k-blocks 0 .. PD-1 (slot d = k-block d); inside the ring a slot whose k-block was the layer's last for it is
// refilled with the NEXT layer's k-block of that slot (pointers wnext, count kbnext): those loads land under the last PD k-blocks' matrix instructions and the
// epilogue.  All NTW column tiles are fetched (the pointers of tiles past the layer's last are clamped by the caller), the matrix instructions of t >= ntw skipped.
// NT: the column tiles this wave multiplies (its first NT of NTW; 0: a wave without tiles in this layer only keeps the ring going)
template <int MTc, int NT, int NTW, int PD, int NS, int WT>
__device__ __forceinline__ void mv_gemm_carried_bw(int KB, const uint16_t* __restrict__ act, int S16, int TS, const uint4* const (&wcur)[NTW], f32x4 (&acc)[MTc][NTW], int lane,
                                                   uint4 (&b)[PD][NTW][WT], const uint4* const (&wnext)[NTW], int kbnext) {
    constexpr int TOP = (NS > WT ? NS : WT) - 1;
    const uint16_t* arow = act + (lane & 15) * S16 + 8 * (lane >> 4);
    uint4 a[MTc][NS];
    if constexpr (NT > 0) {
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int s = 0; s < NS; ++s) a[r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            if (NT > 0 && kb0 + d < KB) {
#pragma unroll
                for (int o = TOP; o >= 0; --o)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int j = o - s;
                        if (j >= 0 && j < WT) {
#pragma unroll
                            for (int t = 0; t < NT; ++t)
#pragma unroll
                                for (int r = 0; r < MTc; ++r)
                                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[d][t][j]), __builtin_bit_cast(mv_bf8, a[r][s]), acc[r][t], 0, 0, 0);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // slot d: this layer's k-block kb0 + d + PD if it has one, else the next layer's k-block d (clamped: no branch around a load)
                const bool more = kb0 + d + PD < KB;
                const int kn = more ? kb0 + d + PD : (d < kbnext ? d : kbnext - 1);
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const uint4* src = (more ? wcur[t] : wnext[t]) + (size_t)kn * WT * 64;
#pragma unroll
                    for (int j = 0; j < WT; ++j) b[d][t][j] = src[j * 64];
                }
                if constexpr (NT > 0) {
                    const int ka = (kb0 + d + 1 < KB) ? kb0 + d + 1 : KB - 1;
#pragma unroll
                    for (int r = 0; r < MTc; ++r)
#pragma unroll
                        for (int s = 0; s < NS; ++s) a[r][s] = *(const uint4*)(arow + s * TS + r * 16 * S16 + ka * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int MTc, int NTW, int NS, int WT, bool DEEP>
__device__ __forceinline__ void mv_gemm_rolling_dispatch_bw(int KB, const uint16_t* act, int S16, int TS, const uint4* wp, int ntw, f32x4 (&acc)[MTc][NTW], int lane) {
    constexpr int PD = (DEEP && MTc * NTW <= 4) ? 4 : 2;
    if (ntw == NTW) { mv_gemm_rolling_bw<MTc, NTW, NTW, PD, NS, WT>(KB, act, S16, TS, wp, acc, lane); return; }
    if (NTW >= 4 && ntw == 3) { mv_gemm_rolling_bw<MTc, (NTW >= 4 ? 3 : 1), NTW, 2, NS, WT>(KB, act, S16, TS, wp, acc, lane); return; }
    if (NTW >= 2 && ntw == 2) { mv_gemm_rolling_bw<MTc, (NTW >= 2 ? 2 : 1), NTW, PD, NS, WT>(KB, act, S16, TS, wp, acc, lane); return; }
    if (ntw == 1) { mv_gemm_rolling_bw<MTc, 1, NTW, 4, NS, WT>(KB, act, S16, TS, wp, acc, lane); return; }
    for (int t0 = 0; t0 < ntw; ++t0) {                              // 5..NTW-1 tiles (wide nets only): one by one
        f32x4 tmp[MTc][NTW];
#pragma unroll
        for (int r = 0; r < MTc; ++r) tmp[r][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mv_gemm_rolling_bw<MTc, 1, NTW, 4, NS, WT>(KB, act, S16, TS, wp + (size_t)t0 * KB * WT * 64, tmp, lane);
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int u = 0; u < NTW; ++u) if (u == t0) acc[r][u] += tmp[r][0];
    }
}

// Which evaluations alternate between two activation tiles (PP of mv_sdf_eval_col0 below): every engine of this family when it evaluates ONE row tile, and the
// sphere tracer's two-tile evaluations (its workgroups run one per CU: the LDS is free).  The callers provide the second region behind the first (act + rows *
// net.S floats): a kernel built for more row tiles has it free whenever it evaluates fewer, otherwise it allocates it (trace.hip: mv_act_rows).
template <class NET> struct mv_bs_pp { static constexpr bool v = false; };
#ifndef MV_BS_PP
#define MV_BS_PP 1                                                 // (-DMV_BS_PP=0: the one-tile form of round 5, for A/B builds -- tools/pp_ab.sh)
#endif
template <int NS, int WT> struct mv_bs_pp<MvNetBs<NS, WT>> { static constexpr bool v = MV_BS_PP != 0; };

// ImplicitNetwork.forward(...)[:, 0] for MTc*16 rows (points in LDS `pts`), bf16 weights x NS-term activations.  Result -> LDS out[row].
// `actf` is the activation region (rows * net.S floats) = NS term tiles of bf16 [rows][S16].  All 64*NW threads must call; ends with a barrier.
// PP (ping-pong, round 6): the layers alternate between `actf` and a second region `actf2` of the same size -- a layer's epilogue writes the tile the NEXT
// layer reads instead of overwriting the one its own matrix loop read, so the "every wave done reading" barrier between matrix loop and epilogue goes and
// ONE barrier per layer is left (a wave can only reach layer l + 1's epilogue, which writes the tile layer l read, after the barrier at the top of layer l + 1,
// i.e. after every wave's layer-l reads).  Same arithmetic, same bits.  Costs a second tile of LDS: used where one is free anyway (mv_bs_pp below).
template <int MTc, int NTW, int NW = 8, bool CARRY_ = false, int NS = 2, int WT = 1, bool PP = false>
__device__ void mv_sdf_eval_col0(const MvNetBs<NS, WT>& net, float* actf, float* pe, const float* pts, float* out, int tid, float* actf2 = nullptr) {
    constexpr bool CARRY = CARRY_ && WT == 1;                       // weight terms: ROLLING only (CARRY_ selects the deeper ring there)
    constexpr int NTHREADS = 64 * NW, PD = mv_bf_pd(NTW, CARRY), PDR = mv_bf_pdr(NTW, CARRY);
    uint16_t* act = (uint16_t*)actf;                                // the tile the current layer READS
    uint16_t* actw = PP ? (uint16_t*)actf2 : act;                   // ... and the one its epilogue WRITES
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int S16 = 2 * net.S / NS, rows = MTc * 16, d0 = 3 + 6 * net.multires, TS = rows * S16;
    const int nl = net.n_layers;
    uint4 b[CARRY ? PD : 1][NTW];                                   // CARRIED: weights of the current / coming layer's first k-blocks
    constexpr bool CARRYW = CARRY_ && WT > 1 && MTc * NTW <= 4;      // ... with weight terms (mv_gemm_carried_bw): one or two row tiles of two column tiles
#ifndef MV_X3_PDW
#define MV_X3_PDW 4
#endif
#ifndef MV_X3_PDW16
#define MV_X3_PDW16 2                                               // 16-wave workgroups (k_sphere_trace's one-tile form): four waves per SIMD hide the fetch, and 24 registers less
#endif                                                              // keep the tracer's ray state out of scratch (probe: 34.0 -> 32.8 us per evaluation; 3: 35.4, 6: 42.3)
    constexpr int PDW = MTc * NTW <= 2 ? (NW >= 16 ? MV_X3_PDW16 : MV_X3_PDW) : 2;   // k-blocks of weight fragments carried in registers (x NTW column tiles x WT terms x 4 VGPRs)
    uint4 bw[CARRYW ? PDW : 1][NTW][WT];
    f32x4 bias4[NTW];                                               // the coming layer's biases
    const uint4* wcur[NTW];                                         // the current layer's column tiles of this wave (+ lane)
    const uint4* wnext[NTW];                                        // the coming layer's, its k-block count
    int kbnext = 1;
    MV_PH_DECL                                                      // (phase stamps of tools/micro/x3_engine_rounds.hip; nothing in the product build)
    auto prep_bias = [&](int l) {
        const MvLayerBf& Ln = net.L[l];
        const int NTn = (l == nl - 1) ? 1 : Ln.NT, c0 = w * ((NTn + NW - 1) / NW);
        kbnext = Ln.KB;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tile = c0 + t < NTn ? c0 + t : NTn - 1;                                // tiles past the layer's last: clamped (loaded, unused)
            wnext[t] = Ln.wp + (size_t)tile * kbnext * 64 * WT + lane;
            bias4[t] = *(const f32x4*)(Ln.bias + tile * 16 + 4 * q);
        }
    };
    auto prep_chunk = [&](int from, int g, int G) {
        if constexpr (CARRY) {
            const int CH = (PD - from) * NTW / G;
#pragma unroll
            for (int j = 0; j < (PD - from) * NTW; ++j) {
                if (j / CH == g || (g == G - 1 && j / CH >= G)) {
                    const int idx = from * NTW + j, d = idx / NTW, t = idx % NTW;
                    const int kb = d < kbnext ? d : kbnext - 1;                              // clamped: no branch around a load
                    b[d][t] = wnext[t][kb * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    prep_bias(0);
    prep_chunk(0, 0, 1);
    if constexpr (CARRYW) {                                         // layer 0's first k-blocks, requested before the positional encoding
#pragma unroll
        for (int d = 0; d < PDW; ++d) {
            const int kb = d < kbnext ? d : kbnext - 1;
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int j = 0; j < WT; ++j) bw[d][t][j] = wnext[t][((size_t)kb * WT + j) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    mv_pe_rows_bs<NTHREADS, NS, (WT > 1)>(pts, pe, act, S16, TS, rows, net.multires, net.L[0].KB * 32, tid);
    MV_PH(0)
    for (int l = 0; l < nl - 1; ++l) {
        const MvLayerBf& L = net.L[l];
        const int NT = L.NT, KB = kbnext;
        const int per = (NT + NW - 1) / NW;
        const int ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        f32x4 acc[MTc][NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            wcur[t] = wnext[t];
#pragma unroll
            for (int a = 0; a < MTc; ++a) acc[a][t] = bias4[t];
        }
        prep_bias(l + 1);
        MV_PH(7)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // inputs of layer l complete (LDS)
        MV_PH(1)
        if constexpr (CARRY) mv_gemm_carried_bs<MTc, NTW, PD, PDR, NS>(KB, act, S16, TS, wcur, NTW, acc, lane, b, wnext, kbnext);      // (all tile slots: see mv_gemm_carried_bw)
        // (every wave multiplies all NTW tiles: the tiles past its share are clamped copies whose results the epilogue drops -- the waves move in lock step,
        // a branch per tile count costs registers at the merge (72 spills) and buys nothing)
        else if constexpr (CARRYW) mv_gemm_carried_bw<MTc, NTW, NTW, PDW, NS, WT>(KB, act, S16, TS, wcur, acc, lane, bw, wnext, kbnext);
        else if constexpr (WT > 1) { if (ntw > 0) mv_gemm_rolling_dispatch_bw<MTc, NTW, NS, WT, CARRY_>(KB, act, S16, TS, wcur[0], ntw, acc, lane); }
        else if (ntw > 0) mv_gemm_rolling_dispatch_bs<MTc, NTW, NS>(KB, act, S16, TS, wcur[0], ntw, acc, lane);
        MV_PH(6)
        if constexpr (!PP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // every wave done reading act (in-place update)
        MV_PH(3)
        {
            const float sc = mv_skip_at(net.skip_mask, l + 1) ? 0.7071067690849304f : 1.0f;   // cat([x, input]) / sqrt(2), idr.py:86-87 (x 1 is exact)
            const int N = L.N;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int col0 = (ct0 + t) * 16 + 4 * q;
#pragma unroll
                for (int a = 0; a < MTc; ++a) {
                    if (t < ntw) {
                        dm_f2 h0, h1;                                                                             // Softplus(beta=100), idr.py:91-92
                        if constexpr (WT > 1) {
                            h0 = dm2_softplus100_lean(dm_f2{acc[a][t][0], acc[a][t][1]}) * dm2_s(sc);     // >= 1.4e-11: above the 2^-40 flush by itself
                            h1 = dm2_softplus100_lean(dm_f2{acc[a][t][2], acc[a][t][3]}) * dm2_s(sc);
                        } else {
                            h0 = mv_softplus100_acc2(dm_f2{acc[a][t][0], acc[a][t][1]}) * dm2_s(sc);
                            h1 = mv_softplus100_acc2(dm_f2{acc[a][t][2], acc[a][t][3]}) * dm2_s(sc);
                        }
                        uint32_t p0[NS], p1[NS];
                        mv_split_pk<NS>(h0, p0);
                        mv_split_pk<NS>(h1, p1);
                        uint16_t* dst = actw + (a * 16 + r) * S16 + col0;
                        if ((ct0 + t) * 16 + 16 <= N) {                                                           // (wave-uniform)
#pragma unroll
                            for (int s = 0; s < NS; ++s) *(uint2*)(dst + s * TS) = uint2{p0[s], p1[s]};
                        } else {                                                                                  // the layer's last, partial tile
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                if (col0 < N) dst[s * TS] = (uint16_t)p0[s];
                                if (col0 + 1 < N) dst[s * TS + 1] = (uint16_t)(p0[s] >> 16);
                                if (col0 + 2 < N) dst[s * TS + 2] = (uint16_t)p1[s];
                                if (col0 + 3 < N) dst[s * TS + 3] = (uint16_t)(p1[s] >> 16);
                            }
                        }
                    }
                    if constexpr (CARRY) {
                        __builtin_amdgcn_sched_barrier(0);
                        prep_chunk(PDR, t * MTc + a, MTc * NTW);                                                  // outside the branch: nothing conditional writes `b`
                    }
                }
            }
            const MvLayerBf& Ln = net.L[l + 1];
            const int Kb = Ln.K, Kp = Ln.KB * 32;
            if (sc != 1.0f) {                                                     // PE part of the skip input
                for (int idx = tid; idx < rows * d0; idx += NTHREADS) {
                    const int row = idx / d0, j = idx - row * d0;
                    uint16_t p[NS];
                    mv_split_1<NS>(WT > 1 ? mv_x3_flush(dm_div_sqrt2(pe[row * d0 + j])) : dm_div_sqrt2(pe[row * d0 + j]), p);
#pragma unroll
                    for (int s = 0; s < NS; ++s) actw[s * TS + row * S16 + N + j] = p[s];
                }
            }
            if (Kp > Kb) {
                const int pad = Kp - Kb;
                for (int idx = tid; idx < rows * pad; idx += NTHREADS) {
                    const int row = idx / pad, j = idx - row * pad;
#pragma unroll
                    for (int s = 0; s < NS; ++s) actw[s * TS + row * S16 + Kb + j] = 0;
                }
            }
        }
        if constexpr (PP) { uint16_t* t_ = act; act = actw; actw = t_; }
        MV_PH(4)
    }
    {   // last layer: column 0 only (wave 0)
        f32x4 acc[MTc][NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            wcur[t] = wnext[t];
#pragma unroll
            for (int a = 0; a < MTc; ++a) acc[a][t] = bias4[t];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MV_PH(1)
        if (w == 0) {
            if constexpr (CARRY) mv_gemm_carried_bs<MTc, NTW, PD, 0, NS>(kbnext, act, S16, TS, wcur, 1, acc, lane, b, wcur, 1);
            else if constexpr (CARRYW) mv_gemm_carried_bw<MTc, 1, NTW, PDW, NS, WT>(kbnext, act, S16, TS, wcur, acc, lane, bw, wcur, 1);
            else if constexpr (WT > 1) mv_gemm_rolling_bw<MTc, 1, NTW, 4, NS, WT>(kbnext, act, S16, TS, wcur[0], acc, lane);
            else mv_gemm_rolling_bs<MTc, 1, NTW, 4, NS>(kbnext, act, S16, TS, wcur[0], acc, lane);
            if (q == 0) {
#pragma unroll
                for (int a = 0; a < MTc; ++a) out[a * 16 + r] = acc[a][0][0];
            }
        }
    }
    MV_PH(2)
    __syncthreads();
    MV_PH(5)
    MV_PH_END
}

// MTc row tiles, ping-pong activation tiles (see mv_sdf_eval_col0): NS / WT deduced from the net
template <int MTc, int NTW, int NW, bool CARRY_, int NS, int WT>
__device__ __forceinline__ void mv_sdf_eval_col0_pp(const MvNetBs<NS, WT>& net, float* actf, float* pe, const float* pts, float* out, int tid) {
    mv_sdf_eval_col0<MTc, NTW, NW, CARRY_, NS, WT, true>(net, actf, pe, pts, out, tid, actf + 16 * MTc * net.S);
}
