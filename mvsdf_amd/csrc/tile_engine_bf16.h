// tile_engine_bf16.h -- the tracing MLP on gfx950's bf16 matrix cores (BASELINE configs[4]: "bf16 MLP weights").
//
// Same row-tile x layer structure as tile_engine.h, with v_mfma_f32_16x16x32_bf16 (fp32 accumulate, 16x the fp32-input MFMA rate):
//   * weights: the folded fp32 weights rounded to bf16 (round-to-nearest-even) at pack time, stored MFMA-packed
//       Wp16[ct][kb][lane][i] = bf16(W[ct*16 + (lane & 15)][col(kb*32 + 8*(lane >> 4) + i)])          (one 16-byte load per lane and k-block)
//   * hidden activations: softplus in fp32 (mv_softplus100_bf2 below), rounded to bf16 when written to LDS (natural [row][k] order: a lane's
//     8 consecutive k are one ds_read_b128; row stride 64*KB + 16 bytes = an odd multiple of 16 bytes: conflict-free);
//   * the geometric inputs keep 16 mantissa bits: every positional-encoding column v (layer 0, and the PE part of the skip layer) enters as
//     TWO bf16 columns hi = bf16(v), lo = bf16(v - hi) that share one weight column -- rounding the ray point itself to 8 bits would move it
//     by up to 4e-3, two orders of magnitude above the tracer's 5e-5 threshold.  K grows from 39 to 78 (layer 0) and 256 to 295 (skip layer);
//   * biases, accumulation (starting from the bias), softplus, the last layer's output: fp32.
// NOT bit-exact against any CPU model: the hardware sums the 32 products of one MFMA with its own internal alignment (probed with
// tools/micro/mfma_bf16_probe.hip: no sequential / pairwise / exact-sum model reproduces it).  The oracle twin (oracle_mvsdf.c, bf16 mode) rounds
// at the same points, uses the same softplus formula and accumulates in fp32 k order; tests bound the difference and state the accuracy budget
// against the fp32 reference.
#pragma once
#include "mlp_common.h"
#include "det_math.h"
#include "det_math_pk.h"

typedef short mv_bf8 __attribute__((ext_vector_type(8)));

// phase stamps of tools/micro/bf16_engine_rounds.hip (dev probe); nothing in the product build
#ifndef MV_PH
#define MV_PH_DECL
#define MV_PH(p)
#define MV_PH_END
#endif

struct MvLayerBf {
    const uint4* wp;    // packed [NT][KB][64] x 8 bf16
    const float* bias;  // [N] fp32, READ UP TO THE NEXT MULTIPLE OF 16 ENTRIES (16-byte loads of four columns; entries past N are loaded and never used:
                        // every hipMalloc'd / torch-allocated buffer is readable that far)
    int K, N;           // true in / out
    int nsplit;         // trailing input columns that enter as hi + lo pairs (layer 0: all of them; skip layer: the PE part)
    int KB, NT;         // k-blocks of 32 over K + nsplit, column tiles of 16
};

struct MvNetBf {
    MvLayerBf L[MV_MAXL];
    int n_layers;
    unsigned skip_mask;
    int multires;
    int S;              // LDS activation row stride in FLOAT units (the bf16 row holds 2*S elements), so LDS carving matches the fp32 engine
};

__host__ __device__ static inline uint16_t mv_f2bf(float f) {     // round to nearest even (finite inputs)
    uint32_t u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float mv_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
// two fp32 -> two bf16 (low half = a) with the gfx950 conversion instruction; same rounding as mv_f2bf for finite inputs
typedef float mv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mv_f2bf_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(mv_f32x2{a, b}, mv_bf16x2));
}
__host__ __device__ static inline int mv_bf_kb(int K, int nsplit) { return (K + nsplit + 31) / 32; }
__host__ __device__ static inline size_t mv_packed_bf16_elems(int N, int K, int nsplit) { return (size_t)mv_ceil16(N) * mv_bf_kb(K, nsplit) * 32; }

// positional encoding -> pe[rows][d0] (fp32, kept for the skip connection) and the layer-0 input row [hi(d0) | lo(d0) | 0 ...] in bf16
template <int NTHREADS>
__device__ __forceinline__ void mv_pe_rows_bf(const float* pts, float* pe, uint16_t* act, int S16, int rows, int multires, int kpad, int tid) {
    const int d0 = 3 + 6 * multires, T = 3 * multires + 1;
    for (int task = tid; task < rows * T; task += NTHREADS) {
        const int row = task / T, j = task - row * T;
        const float* x = pts + row * 3;
        float* pr = pe + row * d0;
        uint16_t* ar = act + row * S16;
        auto put = [&](int col, float v) {
            pr[col] = v;
            const uint16_t hi = mv_f2bf(v);
            ar[col] = hi;
            ar[d0 + col] = mv_f2bf(v - mv_bf2f(hi));
        };
        if (j < 3 * multires) {
            const int m = j / 3, c = j - 3 * m;
            float s, co;
            dm_sincos(x[c] * (float)(1 << m), &s, &co);
            put(3 + 6 * m + c, s);
            put(3 + 6 * m + 3 + c, co);
        } else {
            for (int c = 0; c < 3; ++c) put(c, x[c]);
            for (int c = 2 * d0; c < kpad; ++c) ar[c] = 0;
        }
    }
}

// ---- the two weight-fetch schemes ----
// ROLLING (CARRY = false; the row-sample kernels): a ring of 4 k-blocks of weights per column tile, loaded inside the layer's own contraction
// (L2 latency partly exposed, twice per layer).  ~110 VGPRs: two 8-wave workgroups share a CU and hide each other's waits.
// CARRIED (CARRY = true; k_sphere_trace: one workgroup per CU, every evaluation waits for the previous one): the first 8 k-blocks -- all of
// a 256-wide layer -- of the NEXT layer are loaded while THIS layer computes: half of them inside the ring, into the registers the ring has
// just used, the rest in chunks between the groups of the epilogue.  Measured on tools/micro/bf16_engine_rounds.hip (32 rows per CU, 256
// CUs, us per evaluation): rolling 26.2 -> 24.9 with the epilogue below -> 22.9 carried.  What the probe says about this engine:
//   * the weights stream from L2 at ~58 B/clk and CU (256 CUs streaming 1.1 MB each: 10 us per evaluation = L2 bandwidth,
//     tools/micro/l2_weight_stream.hip), and a wave that issues loads faster than that stalls IN THE ISSUE, with its matrix and VALU work
//     behind it: loads have to be spread over ring and epilogue, not issued as one batch (one batch after the ring: 0.93 us per layer);
//   * no load may sit inside a branch and no branch may rewrite the weight registers: the compiler then merges register assignments with
//     copies, and a copy of a register with a load in flight waits for ALL loads (seen in the ISA: such a prefetch hides nothing);
//   * a run-time trip count around loads makes the compiler wait for everything before every k-block: the ring is unrolled over 8 k-blocks
//     with the matrix instructions (not the loads) skipped past the layer's count;
//   * without matrix instructions, without weight loads and without the exponential the evaluation still takes 18.8 of 22.9 us: LDS reads
//     of the activations (every wave reads the whole tile: 128 KB per layer = the matrix time), dependent VALU chains, barrier skew between
//     the two waves of a SIMD, layer descriptors.  The bf16 matrix pipe is busy ~20 % of the time; that is this design's plateau at 16-32
//     rows per CU.
__host__ __device__ constexpr int mv_bf_pd(int NTW, bool carry) { return (carry && NTW < 4) ? 8 : 4; }   // weight k-blocks in registers per column tile
__host__ __device__ constexpr int mv_bf_pdr(int NTW, bool carry) { return carry ? mv_bf_pd(NTW, carry) / 2 : 0; }   // ... of which re-loaded inside the ring
#define MV_BF_PDA 4                                                 // activation k-blocks in flight (LDS)

// Both rings: acc[rt][t] += (act[rt*16.., :] * Wp16[tile t, :]^T)^T -- the WEIGHTS are the matrix instruction's first operand, so a lane ends
// up with FOUR CONSECUTIVE OUTPUT COLUMNS (4q..4q+3 of the tile) of ONE row (r): the next layer's k order, one 8-byte LDS write per accumulator.

// CARRIED.  `b` holds this layer's first PD k-blocks; b[0 .. PDR) are overwritten with the next layer's (pointers `wnext`, count `kbnext`)
// right after their last use.  K-blocks past PD (the skip layer: 10; 512-wide nets) are fetched and used at the end, latency exposed.
template <int MTc, int NTW, int PD, int PDR>
__device__ __forceinline__ void mv_gemm_carried_bf(int KB, const uint16_t* __restrict__ act, int S16, const uint4* const (&wcur)[NTW], int ntw,
                                                   f32x4 (&acc)[MTc][NTW], int lane, uint4 (&b)[PD][NTW], const uint4* const (&wnext)[NTW], int kbnext) {
    constexpr int PA = MV_BF_PDA;
    static_assert(PD % PA == 0, "weight register depth must be a multiple of the activation ring depth");
    const uint16_t* arow = act + (lane & 15) * S16 + 8 * (lane >> 4);
    uint4 a[PA][MTc];
#pragma unroll
    for (int d = 0; d < PA; ++d) {
        const int kb = d < KB ? d : KB - 1;
#pragma unroll
        for (int r = 0; r < MTc; ++r) a[d][r] = *(const uint4*)(arow + r * 16 * S16 + kb * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < PD; ++kb) {
        if (kb < KB) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                if (t < ntw) {
#pragma unroll
                    for (int r = 0; r < MTc; ++r)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[kb][t]), __builtin_bit_cast(mv_bf8, a[kb % PA][r]), acc[r][t], 0, 0, 0);
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
            for (int r = 0; r < MTc; ++r) a[kb % PA][r] = *(const uint4*)(arow + r * 16 * S16 + ka * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int kb = PD; kb < KB; ++kb) {
        uint4 bx[NTW], ax[MTc];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bx[t] = wcur[t][kb * 64];
#pragma unroll
        for (int r = 0; r < MTc; ++r) ax[r] = *(const uint4*)(arow + r * 16 * S16 + kb * 32);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            if (t < ntw) {
#pragma unroll
                for (int r = 0; r < MTc; ++r)
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, bx[t]), __builtin_bit_cast(mv_bf8, ax[r]), acc[r][t], 0, 0, 0);
            }
    }
}

// ROLLING.  NT = the wave's column tiles in this layer (they are consecutive: wp = the first one's pack + lane).  The tail of the ring
// reloads the last k-block (clamped index) rather than branching around loads.
template <int MTc, int NT, int NTW, int PD>
__device__ __forceinline__ void mv_gemm_rolling_bf(int KB, const uint16_t* __restrict__ act, int S16, const uint4* __restrict__ wp, f32x4 (&acc)[MTc][NTW], int lane) {
    const uint16_t* arow = act + (lane & 15) * S16 + 8 * (lane >> 4);
    uint4 b[PD][NT], a[PD][MTc];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        const int kb = d < KB ? d : KB - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + kb) * 64];
#pragma unroll
        for (int r = 0; r < MTc; ++r) a[d][r] = *(const uint4*)(arow + r * 16 * S16 + kb * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            if (kb0 + d < KB) {
#pragma unroll
                for (int r = 0; r < MTc; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[d][t]), __builtin_bit_cast(mv_bf8, a[d][r]), acc[r][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kn = (kb0 + d + PD < KB) ? kb0 + d + PD : KB - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + kn) * 64];
#pragma unroll
            for (int r = 0; r < MTc; ++r) a[d][r] = *(const uint4*)(arow + r * 16 * S16 + kn * 32);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MTc, int NTW>
__device__ __forceinline__ void mv_gemm_rolling_dispatch_bf(int KB, const uint16_t* act, int S16, const uint4* wp, int ntw, f32x4 (&acc)[MTc][NTW], int lane) {
    if (ntw == NTW) { mv_gemm_rolling_bf<MTc, NTW, NTW, 4>(KB, act, S16, wp, acc, lane); return; }
    if (NTW >= 4 && ntw == 3) { mv_gemm_rolling_bf<MTc, (NTW >= 4 ? 3 : 1), NTW, 4>(KB, act, S16, wp, acc, lane); return; }
    if (NTW >= 2 && ntw == 2) { mv_gemm_rolling_bf<MTc, (NTW >= 2 ? 2 : 1), NTW, 4>(KB, act, S16, wp, acc, lane); return; }
    if (ntw == 1) { mv_gemm_rolling_bf<MTc, 1, NTW, 4>(KB, act, S16, wp, acc, lane); return; }
    for (int t0 = 0; t0 < ntw; ++t0) {                              // 5..NTW-1 tiles (wide nets only): one by one
        f32x4 tmp[MTc][NTW];
#pragma unroll
        for (int r = 0; r < MTc; ++r) tmp[r][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mv_gemm_rolling_bf<MTc, 1, NTW, 4>(KB, act, S16, wp + (size_t)t0 * KB * 64, tmp, lane);
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int u = 0; u < NTW; ++u) if (u == t0) acc[r][u] += tmp[r][0];
    }
}

// Softplus(beta=100, threshold=20) for activations that are rounded to bf16 right after (8 mantissa bits), with the scalings folded:
//     softplus(100 z) / 100 = max(z, 0) + ln(1 + t) / 100,   t = 2^(-|z| * 100 log2 e) in (0, 1]      (v_exp_f32: 1 ulp, deterministic on the device)
//     ln(1 + t) / 100 = t * Q(t),  Q = a degree-5 fit of ln(1 + t) / (100 t) on [0, 1] with relative error 8.5e-6 = 2^-17 -- 1/230 of the half
//     ulp of the bf16 rounding that follows (degree 4, 5.7e-5, flips that rounding for ~2 % of the activations: the mean distance to the
//     oracle's twin went from 5e-5 to 2e-4) -- evaluated for two activations per instruction (v_pk_fma_f32).
// 24 ordinary (14 of them two-wide) + 4 transcendental VALU instructions per four activations instead of 27 per activation (det_math): this
// engine is bound by issue and waits, not by the bf16 MFMAs (PMC: DESIGN.md).  Above the threshold (100 z > 20) t * Q(t) is below half an ulp
// of z: the sum IS z, no select needed.  After the bf16 rounding the result agrees with dm_softplus100's except at rounding boundaries, the
// same kind of difference as the MFMA's summation order (tests/test_gpu_bf16.py bounds both against the oracle's twin, which keeps
// dm_softplus100).
__device__ __forceinline__ dm_f2 mv_softplus100_bf2(dm_f2 z) {
    const dm_f2 t = dm_f2{__builtin_amdgcn_exp2f(fabsf(z.x) * -144.26950408889634f), __builtin_amdgcn_exp2f(fabsf(z.y) * -144.26950408889634f)};
    dm_f2 u = dm2_s(-2.3869141936302185e-4f);
    u = dm2_fma(u, t, dm2_s(1.0122226178646088e-3f));
    u = dm2_fma(u, t, dm2_s(-2.1004866063594818e-3f));
    u = dm2_fma(u, t, dm2_s(3.252066671848297e-3f));
    u = dm2_fma(u, t, dm2_s(-4.993613660335541e-3f));
    u = dm2_fma(u, t, dm2_s(9.999915957450867e-3f));
    // max(z, 0) in one instruction (fmaxf adds a canonicalising v_max)
    return dm2_fma(t, u, dm_f2{__builtin_amdgcn_fmed3f(z.x, 0.0f, 3.0e38f), __builtin_amdgcn_fmed3f(z.y, 0.0f, 3.0e38f)});
}

// ImplicitNetwork.forward(...)[:, 0] for MTc*16 rows (points in LDS `pts`) with bf16 weights / activations.  Result -> LDS out[row].
// `actf` is the activation region (rows * net.S floats), used as bf16 [rows][2*S].  All 64*NW threads must call; ends with a barrier.
// z = bias + sum: the accumulators start from the bias (16-byte loads of four consecutive columns: the bias vector is read up to the next
// multiple of 16 entries -- see MvLayerBf::bias).
template <int MTc, int NTW, int NW = 8, bool CARRY = false>
__device__ void mv_sdf_eval_col0(const MvNetBf& net, float* actf, float* pe, const float* pts, float* out, int tid) {
    constexpr int NTHREADS = 64 * NW, PD = mv_bf_pd(NTW, CARRY), PDR = mv_bf_pdr(NTW, CARRY);
    uint16_t* act = (uint16_t*)actf;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int S16 = 2 * net.S, rows = MTc * 16, d0 = 3 + 6 * net.multires;
    const int nl = net.n_layers;
    MV_PH_DECL
    uint4 b[CARRY ? PD : 1][NTW];                                   // CARRIED: weights of the current / coming layer's first k-blocks
    f32x4 bias4[NTW];                                               // the coming layer's biases
    const uint4* wcur[NTW];                                         // the current layer's column tiles of this wave (+ lane)
    const uint4* wnext[NTW];                                        // the coming layer's, its k-block count
    int kbnext = 1;
    // descriptors + biases of layer l (CARRIED: its weights follow through the ring / prep_chunk)
    auto prep_bias = [&](int l) {
        const MvLayerBf& Ln = net.L[l];
        const int NTn = (l == nl - 1) ? 1 : Ln.NT, c0 = w * ((NTn + NW - 1) / NW);
        kbnext = Ln.KB;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tile = c0 + t < NTn ? c0 + t : NTn - 1;                                // tiles past the layer's last: clamped (loaded, unused)
            wnext[t] = Ln.wp + (size_t)tile * kbnext * 64 + lane;
            bias4[t] = *(const f32x4*)(Ln.bias + tile * 16 + 4 * q);                         // not looked at before the next layer starts: stays in flight
        }
    };
    // CARRIED: chunk g of G of the coming layer's weight registers from k-block `from` on, in (k-block, tile) order
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
    mv_pe_rows_bf<NTHREADS>(pts, pe, act, S16, rows, net.multires, net.L[0].KB * 32, tid);
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
        // (CARRIED: every wave multiplies all NTW column-tile slots -- the tiles past its share are clamped copies whose results the epilogue drops; the waves
        // move in lock step, and a branch per tile count splits the matrix-instruction runs of the ring)
        if constexpr (CARRY) mv_gemm_carried_bf<MTc, NTW, PD, PDR>(KB, act, S16, wcur, NTW, acc, lane, b, wnext, kbnext);
        else if (ntw > 0) mv_gemm_rolling_dispatch_bf<MTc, NTW>(KB, act, S16, wcur[0], ntw, acc, lane);
        MV_PH(6)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // every wave done reading act (in-place update)
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
                        const dm_f2 h0 = mv_softplus100_bf2(dm_f2{acc[a][t][0], acc[a][t][1]}) * dm2_s(sc);       // Softplus(beta=100), idr.py:91-92
                        const dm_f2 h1 = mv_softplus100_bf2(dm_f2{acc[a][t][2], acc[a][t][3]}) * dm2_s(sc);
                        const uint32_t w0 = mv_f2bf_pk(h0.x, h0.y), w1 = mv_f2bf_pk(h1.x, h1.y);                   // v_cvt_pk_bf16_f32 (round to nearest even)
                        uint16_t* dst = act + (a * 16 + r) * S16 + col0;
                        if ((ct0 + t) * 16 + 16 <= N) *(uint2*)dst = uint2{w0, w1};                               // (wave-uniform)
                        else {                                                                                    // the layer's last, partial tile
                            if (col0 < N) dst[0] = (uint16_t)w0;
                            if (col0 + 1 < N) dst[1] = (uint16_t)(w0 >> 16);
                            if (col0 + 2 < N) dst[2] = (uint16_t)w1;
                            if (col0 + 3 < N) dst[3] = (uint16_t)(w1 >> 16);
                        }
                    }
                    if constexpr (CARRY) {
                        __builtin_amdgcn_sched_barrier(0);
                        prep_chunk(PDR, t * MTc + a, MTc * NTW);                                                  // outside the branch: nothing conditional writes `b`
                    }
                }
            }
            const MvLayerBf& Ln = net.L[l + 1];
            const int Kb = Ln.K + Ln.nsplit, Kp = Ln.KB * 32;
            if (sc != 1.0f) {                                                     // PE part of the skip input: hi + lo pairs
                for (int idx = tid; idx < rows * d0; idx += NTHREADS) {
                    const int row = idx / d0, j = idx - row * d0;
                    const float v = dm_div_sqrt2(pe[row * d0 + j]);
                    const uint16_t hi = mv_f2bf(v);
                    act[row * S16 + N + j] = hi;
                    act[row * S16 + N + d0 + j] = mv_f2bf(v - mv_bf2f(hi));
                }
            }
            if (Kp > Kb) {
                const int pad = Kp - Kb;
                for (int idx = tid; idx < rows * pad; idx += NTHREADS) {
                    const int row = idx / pad, j = idx - row * pad;
                    act[row * S16 + Kb + j] = 0;
                }
            }
        }
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
            if constexpr (CARRY) mv_gemm_carried_bf<MTc, NTW, PD, 0>(kbnext, act, S16, wcur, 1, acc, lane, b, wcur, 1);
            else mv_gemm_rolling_bf<MTc, 1, NTW, 4>(kbnext, act, S16, wcur[0], acc, lane);
            if (q == 0) {
#pragma unroll
                for (int a = 0; a < MTc; ++a) out[a * 16 + r] = acc[a][0][0];
            }
        }
        MV_PH(2)
    }
    __syncthreads();
    MV_PH(5)
    MV_PH_END
}
