// tile_engine_bf16.h -- the tracing MLP on gfx950's bf16 matrix cores (BASELINE configs[4]: "bf16 MLP weights").
//
// Same row-tile x layer structure as tile_engine.h, with v_mfma_f32_16x16x32_bf16 (fp32 accumulate, 16x the fp32-input MFMA rate):
//   * weights: the folded fp32 weights rounded to bf16 (round-to-nearest-even) at pack time, stored MFMA-packed
//       Wp16[ct][kb][lane][i] = bf16(W[ct*16 + (lane & 15)][col(kb*32 + 8*(lane >> 4) + i)])          (one 16-byte load per lane and k-block)
//   * hidden activations: softplus in fp32 (deterministic det_math.h), rounded to bf16 when written to LDS (natural [row][k] order: a lane's
//     8 consecutive k are one ds_read_b128; row stride 64*KB + 16 bytes = an odd multiple of 16 bytes: conflict-free);
//   * the geometric inputs keep 16 mantissa bits: every positional-encoding column v (layer 0, and the PE part of the skip layer) enters as
//     TWO bf16 columns hi = bf16(v), lo = bf16(v - hi) that share one weight column -- rounding the ray point itself to 8 bits would move it
//     by up to 4e-3, two orders of magnitude above the tracer's 5e-5 threshold.  K grows from 39 to 78 (layer 0) and 256 to 295 (skip layer);
//   * biases, accumulation, softplus, the last layer's output: fp32.
// NOT bit-exact against any CPU model: the hardware sums the 32 products of one MFMA with its own internal alignment (probed with
// tools/micro/mfma_bf16_probe.hip: no sequential / pairwise / exact-sum model reproduces it).  The oracle twin (oracle_mvsdf.c, bf16 mode) rounds
// at the same points and accumulates in fp32 k order; tests bound the difference and state the accuracy budget against the fp32 reference.
#pragma once
#include "mlp_common.h"
#include "det_math.h"
#include "det_math_pk.h"

typedef short mv_bf8 __attribute__((ext_vector_type(8)));

struct MvLayerBf {
    const uint4* wp;    // packed [NT][KB][64] x 8 bf16
    const float* bias;  // [N] fp32
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

// acc[rt][t] += act[rt*16.., :] * Wp16[(ct0+t)*16.., :]^T, ring of PD k-blocks in flight (A from LDS, B from L2)
template <int MTc, int NT, int NTW, int PD>
__device__ __forceinline__ void mv_gemm_ring_bf(const MvLayerBf& L, const uint16_t* __restrict__ act, int S16, int ct0, f32x4 (&acc)[MTc][NTW], int lane) {
    const int KB = L.KB;
    const uint4* __restrict__ wp = L.wp + (size_t)ct0 * KB * 64 + lane;
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
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, a[d][r]), __builtin_bit_cast(mv_bf8, b[d][t]), acc[r][t], 0, 0, 0);
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
__device__ __forceinline__ void mv_gemm_dispatch_bf(const MvLayerBf& L, const uint16_t* act, int S16, int ct0, int ntw, f32x4 (&acc)[MTc][NTW], int lane) {
    if (ntw == NTW) { mv_gemm_ring_bf<MTc, NTW, NTW, 4>(L, act, S16, ct0, acc, lane); return; }
    if (NTW >= 4 && ntw == 3) { mv_gemm_ring_bf<MTc, (NTW >= 4 ? 3 : 1), NTW, 4>(L, act, S16, ct0, acc, lane); return; }
    if (NTW >= 2 && ntw == 2) { mv_gemm_ring_bf<MTc, (NTW >= 2 ? 2 : 1), NTW, 4>(L, act, S16, ct0, acc, lane); return; }
    if (ntw == 1) { mv_gemm_ring_bf<MTc, 1, NTW, 4>(L, act, S16, ct0, acc, lane); return; }
    for (int t0 = 0; t0 < ntw; ++t0) {                              // 5..NTW-1 tiles (wide nets only): one by one
        f32x4 tmp[MTc][NTW];
#pragma unroll
        for (int r = 0; r < MTc; ++r) tmp[r][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        mv_gemm_ring_bf<MTc, 1, NTW, 4>(L, act, S16, ct0 + t0, tmp, lane);
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int u = 0; u < NTW; ++u) if (u == t0) acc[r][u] = tmp[r][0];
    }
}

// ImplicitNetwork.forward(...)[:, 0] for MTc*16 rows (points in LDS `pts`) with bf16 weights / activations.  Result -> LDS out[row].
// Softplus(beta=100, threshold=20) for activations that are rounded to bf16 right after (8 mantissa bits): the hardware's v_exp_f32 /
// v_log_f32 (1 ulp of fp32, deterministic on the device, not reproducible on a CPU) instead of det_math's correctly-rounded polynomial chains,
// with the scalings folded:  softplus(100 z) / 100 = max(z, 0) + log2(1 + 2^(-|z| * 100 log2 e)) * (ln 2 / 100)
// -- 5 ordinary + 2 transcendental VALU instructions per activation instead of 27, and this engine is bound by its epilogue's VALU work, not
// by the bf16 MFMAs.  Above the threshold (100 z > 20) the log term is below half an ulp of z: the sum IS z, no select needed.  Against
// dm_softplus100 the result differs by a few 1e-7 relative while the log term matters and by < 1e-9 absolute where it does not (1 + t rounds
// t away once t < 2^-24); after the bf16 rounding the two agree except at rounding boundaries, the same kind of difference as the MFMA's
// summation order (tests/test_gpu_bf16.py bounds both against the oracle's twin, which keeps dm_softplus100).
__device__ __forceinline__ float mv_softplus100_bf(float z) {
    const float t = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);            // exp(-|100 z|)
    return fmaf(__builtin_amdgcn_logf(1.0f + t), 0.006931471805599453f, fmaxf(z, 0.0f));
}

// `actf` is the activation region (rows * net.S floats), used as bf16 [rows][2*S].  All 64*NW threads must call; ends with a barrier.
template <int MTc, int NTW, int NW = 8>
__device__ void mv_sdf_eval_col0(const MvNetBf& net, float* actf, float* pe, const float* pts, float* out, int tid) {
    constexpr int NTHREADS = 64 * NW;
    uint16_t* act = (uint16_t*)actf;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int S16 = 2 * net.S, rows = MTc * 16, d0 = 3 + 6 * net.multires;
    mv_pe_rows_bf<NTHREADS>(pts, pe, act, S16, rows, net.multires, net.L[0].KB * 32, tid);
    const int nl = net.n_layers;
    for (int l = 0; l < nl; ++l) {
        const MvLayerBf& L = net.L[l];
        const bool last = (l == nl - 1);
        const int NT = last ? 1 : L.NT;
        const int per = (NT + NW - 1) / NW;
        const int ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        f32x4 acc[MTc][NTW];
#pragma unroll
        for (int a = 0; a < MTc; ++a)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bv_[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = (ct0 + t) * 16 + r;
            bv_[t] = (t < ntw && col < L.N) ? L.bias[col] : 0.0f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // inputs of layer l complete (LDS)
        if (ntw > 0) mv_gemm_dispatch_bf<MTc, NTW>(L, act, S16, ct0, ntw, acc, lane);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // every wave done reading act (in-place update)
        if (last) {
            if (w == 0 && r == 0) {
                const float b0 = bv_[0];
#pragma unroll
                for (int a = 0; a < MTc; ++a)
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[a * 16 + 4 * q + i] = acc[a][0][i] + b0;
            }
        } else {
            const bool to_skip = mv_skip_at(net.skip_mask, l + 1);
            const int N = L.N;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (t < ntw) {
                    const int col = (ct0 + t) * 16 + r;
                    if (col < N) {
                        const float bv = bv_[t];
#pragma unroll
                        for (int a = 0; a < MTc; ++a)
#pragma unroll
                            for (int i = 0; i < 4; i += 2) {
                                dm_f2 h = dm_f2{mv_softplus100_bf(acc[a][t][i] + bv), mv_softplus100_bf(acc[a][t][i + 1] + bv)};   // Softplus(beta=100), idr.py:91-92
                                if (to_skip) h = h * dm2_s(0.7071067690849304f);                              // cat([x, input]) / sqrt(2), idr.py:86-87
                                const uint32_t hb = mv_f2bf_pk(h.x, h.y);                 // one v_cvt_pk_bf16_f32 (round to nearest even)
                                act[(a * 16 + 4 * q + i) * S16 + col] = (uint16_t)hb;
                                act[(a * 16 + 4 * q + i + 1) * S16 + col] = (uint16_t)(hb >> 16);
                            }
                    }
                }
            }
            const MvLayerBf& Ln = net.L[l + 1];
            const int Kb = Ln.K + Ln.nsplit, Kp = Ln.KB * 32;
            if (to_skip) {                                                        // PE part of the skip input: hi + lo pairs
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
    }
    __syncthreads();
}
