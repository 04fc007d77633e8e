// tile_engine.h -- the MFMA "row-tile x layer" engine shared by every MLP kernel.
//
// One 256-thread workgroup (4 waves) owns up to 16*MT activation rows in LDS.  For one Linear:
//   * wave w computes column tiles [w*ntw, w*ntw + ntw) for ALL row tiles,
//   * A fragments (activations) come from LDS with one ds_read_b128 per 16-wide k-block (the row is
//     stored 4x4-transposed inside each 16-block, see mv_perm, so the lane's four k-steps are contiguous),
//   * B fragments (weights) come straight from the MFMA-packed global array (L2-resident: the whole
//     8x256 SDF net is 2.1 MB) with one coalesced 16-byte load per lane per k-block, double buffered in
//     registers -- weights are private to a wave's columns, so staging them in LDS would only add traffic;
//     every loaded weight is reused by all MT row tiles of the workgroup's rays,
//   * the product is v_mfma_f32_16x16x4_f32 issued in ascending k: a k-ordered fmaf chain (bit-exact
//     against the scalar restatement).
#pragma once
#include "mlp_common.h"
#include "det_math.h"
#include "det_math_pk.h"

#define MV_THREADS 256

// phase stamps of tools/micro/f32_engine_rounds.hip (dev probe); nothing in the product build
#ifndef MV_PH
#define MV_PH_DECL
#define MV_PH(p)
#define MV_PH_END
#endif

// dev-only ablation switches (never set in the shipped build): 1 = softplus -> identity, 2 = skip the GEMM
#ifndef MV_ABLATE
#define MV_ABLATE 0
#endif
__device__ __forceinline__ float mv_act(float z) { return (MV_ABLATE & 1) ? z * 0.5f : dm_softplus100(z); }
__device__ __forceinline__ dm_f2 mv_act2(dm_f2 z) { return (MV_ABLATE & 1) ? z * dm2_s(0.5f) : dm2_softplus100(z); }

// Workgroup barrier that orders LDS traffic only: waits for this wave's outstanding LDS operations (lgkmcnt), NOT for its global loads /
// stores.  __syncthreads() also drains vmcnt, which forces global loads issued early (to overlap a GEMM) to land before the barrier.
// Use only where the data exchanged between the waves at this point lives in LDS.
__device__ __forceinline__ void mv_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MTc, int NTW>
__device__ __forceinline__ void mv_zero_acc(f32x4 (&acc)[MTc][NTW]) {
#pragma unroll
    for (int a = 0; a < MTc; ++a)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc[rt][t] += act[rt*16.., :K] * W[(ct0+t)*16.., :K]^T    for t < NT (compile time), rt < MTc.
// Software pipelined by hand with a ring of PD stages: the A (LDS) and B (global/L2) fragments of k-blocks kb+1..kb+PD-1 are
// in flight while the 4*MTc*NT MFMAs of k-block kb issue; a stage is refilled right after its MFMAs.  KB % PD == 0.
// No guards inside: the caller dispatches on the (wave-uniform) tile count.  sched_barrier pins the order (the scheduler
// would otherwise sink the prefetches below the MFMAs to save registers).
template <int MTc, int NT, int NTW, int PD>
__device__ __forceinline__ void mv_gemm_ring(const MvLayer& L, const float* __restrict__ act, int S, int ct0,
                                             f32x4 (&acc)[MTc][NTW], int lane) {
    const int KB = L.KB;
    const float4* __restrict__ wp = L.wp + (size_t)ct0 * KB * 64 + lane;
    const float* arow = act + (lane & 15) * S + 4 * (lane >> 4);
    float4 b[PD][NT], a[PD][MTc];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
#pragma unroll
        for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + d) * 64];
#pragma unroll
        for (int r = 0; r < MTc; ++r) a[d][r] = *(const float4*)(arow + r * 16 * S + d * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int r = 0; r < MTc; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a[d][r])[s], ((const float*)&b[d][t])[s], acc[r][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int kn = (kb0 + d + PD < KB) ? kb0 + d + PD : kb0 + d;      // tail: harmless re-load of the same block
            if (!(MV_ABLATE & 8)) {
#pragma unroll
                for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + kn) * 64];
            }
            if (!(MV_ABLATE & 4)) {
#pragma unroll
                for (int r = 0; r < MTc; ++r) a[d][r] = *(const float4*)(arow + r * 16 * S + kn * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The same ring carried ACROSS layers (mv_sdf_eval_col0): the weights of a layer do not depend on the activations, so the ring stages that
// fall free while the last PD k-blocks of layer l issue are refilled with k-blocks 0..PD-1 of layer l+1 (`wn`, this wave's tiles there;
// nullptr = nothing to prefetch).  Their L2 latency then hides under those MFMAs, the barrier and the softplus epilogue instead of opening
// layer l+1's GEMM.  `primed` (wave-uniform): b already holds this layer's first PD k-blocks.  Same MFMAs in the same order as mv_gemm_ring.
template <int MTc, int NT, int NTW, int PD>
__device__ __forceinline__ void mv_gemm_ring_x(const MvLayer& L, const float* __restrict__ act, int S, int ct0, f32x4 (&acc)[MTc][NTW], int lane,
                                               float4 (&b)[PD][NT], bool primed, const float4* __restrict__ wn, int KBn) {
    const int KB = L.KB;
    const float4* __restrict__ wp = L.wp + (size_t)ct0 * KB * 64 + lane;
    const float* arow = act + (lane & 15) * S + 4 * (lane >> 4);
    float4 a[PD][MTc];
    if (!primed) {
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + d) * 64];
    }
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int r = 0; r < MTc; ++r) a[d][r] = *(const float4*)(arow + r * 16 * S + d * 16);
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < KB - PD; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int r = 0; r < MTc; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a[d][r])[s], ((const float*)&b[d][t])[s], acc[r][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int kn = kb0 + d + PD;
#pragma unroll
            for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + kn) * 64];
#pragma unroll
            for (int r = 0; r < MTc; ++r) a[d][r] = *(const float4*)(arow + r * 16 * S + kn * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // last PD k-blocks: the stages they free take the next layer's first k-blocks
#pragma unroll
    for (int d = 0; d < PD; ++d) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < MTc; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a[d][r])[s], ((const float*)&b[d][t])[s], acc[r][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (wn) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[d][t] = wn[((size_t)t * KBn + d) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MTc, int NT, int NTW>
__device__ __forceinline__ void mv_gemm_tiles(const MvLayer& L, const float* __restrict__ act, int S, int ct0,
                                              f32x4 (&acc)[MTc][NTW], int lane) {
    // deep ring when a k-block is short (few row tiles) and the register budget allows it
    if (MTc * NT <= 8 && (L.KB & 3) == 0) mv_gemm_ring<MTc, NT, NTW, 4>(L, act, S, ct0, acc, lane);
    else mv_gemm_ring<MTc, NT, NTW, 2>(L, act, S, ct0, acc, lane);
}

// wave-uniform dispatch on the number of column tiles this wave owns (1..NTW)
template <int MTc, int NTW>
__device__ __forceinline__ void mv_gemm_dispatch(const MvLayer& L, const float* __restrict__ act, int S, int ct0, int ntw,
                                                 f32x4 (&acc)[MTc][NTW], int lane) {
    if (ntw == NTW) { mv_gemm_tiles<MTc, NTW, NTW>(L, act, S, ct0, acc, lane); return; }
    if (NTW > 4) {
        if (ntw >= 4) {      // 4..NTW-1: a group of 4, then the rest one by one (rare shapes)
            mv_gemm_tiles<MTc, 4, NTW>(L, act, S, ct0, acc, lane);
            for (int t = 4; t < ntw; ++t) {
                f32x4 tmp[MTc][NTW];
#pragma unroll
                for (int r = 0; r < MTc; ++r) tmp[r][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                mv_gemm_tiles<MTc, 1, NTW>(L, act, S, ct0 + t, tmp, lane);
#pragma unroll
                for (int r = 0; r < MTc; ++r)
#pragma unroll
                    for (int u = 4; u < NTW; ++u) if (u == t) acc[r][u] = tmp[r][0];
            }
            return;
        }
    }
    if (ntw == 3) { mv_gemm_tiles<MTc, (NTW >= 3 ? 3 : 1), NTW>(L, act, S, ct0, acc, lane); return; }
    if (ntw == 2) { mv_gemm_tiles<MTc, (NTW >= 2 ? 2 : 1), NTW>(L, act, S, ct0, acc, lane); return; }
    if (ntw == 1) { mv_gemm_tiles<MTc, 1, NTW>(L, act, S, ct0, acc, lane); return; }
}

// Positional encoding of `rows` points (LDS pts[rows][3]) -> pe[rows][d0] (natural order, kept for the skip
// connection) and act[rows][S] (permuted, zero padded to ceil32(d0)).  embedder.py:10-36.
template <int NTHREADS>
__device__ __forceinline__ void mv_pe_rows(const float* pts, float* pe, float* act, int S, int rows, int multires, int tid) {
    const int d0 = 3 + 6 * multires, Kp0 = mv_kpad(d0), T = 3 * multires + 1;
    for (int task = tid; task < rows * T; task += NTHREADS) {
        const int row = task / T, j = task - row * T;
        const float* x = pts + row * 3;
        float* pr = pe + row * d0;
        float* ar = act + row * S;
        if (j < 3 * multires) {
            const int m = j / 3, c = j - 3 * m;
            float s, co;
            dm_sincos(x[c] * (float)(1 << m), &s, &co);
            const int cs = 3 + 6 * m + c, cc = cs + 3;
            pr[cs] = s; pr[cc] = co;
            ar[mv_perm(cs)] = s; ar[mv_perm(cc)] = co;
        } else {
            for (int c = 0; c < 3; ++c) { pr[c] = x[c]; ar[mv_perm(c)] = x[c]; }
            for (int c = d0; c < Kp0; ++c) ar[mv_perm(c)] = 0.0f;
        }
    }
}

// ImplicitNetwork.forward(...)[:, 0] (idr.py:77-94) for MTc*16 rows whose points sit in LDS `pts`.
// Result -> LDS out[row].  All 64*NW threads must call; ends with a barrier.  NW waves share the column tiles
// (NW = 8 puts two waves on every SIMD: one wave's LDS / L2 latency and epilogue hide behind the other's MFMAs).
// XR: carry the weight ring across layers (mv_gemm_ring_x).  It keeps 32 more registers live through the softplus epilogue: for the kernels
// that run one workgroup per CU (the sphere tracer), not for those that need 4 waves per SIMD.
template <int MTc, int NTW, int NW = 4, bool XR = false>
__device__ void mv_sdf_eval_col0(const MvNet& net, float* act, float* pe, const float* pts, float* out, int tid) {
    constexpr int NTHREADS = 64 * NW;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int S = net.S, rows = MTc * 16, d0 = 3 + 6 * net.multires;
    MV_PH_DECL
    mv_pe_rows<NTHREADS>(pts, pe, act, S, rows, net.multires, tid);
    MV_PH(0)
    const int nl = net.n_layers;
    constexpr bool XRING = XR && (MTc * NTW <= 8);                // the deep ring (PD = 4) and its carry across layers (mv_gemm_ring_x)
    float4 bring[4][NTW];
    bool primed = false;
    for (int l = 0; l < nl; ++l) {
        const MvLayer& L = net.L[l];
        const bool last = (l == nl - 1);
        const int NT = last ? 1 : L.NT;                       // tracing needs column 0 only
        const int per = (NT + NW - 1) / NW;                   // column tiles per wave
        const int ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        const bool xr = XRING && ntw == NTW && (L.KB & 3) == 0;
        // this wave's tiles of the next layer, if it runs the carried ring there too
        const float4* wn = nullptr;
        int KBn = 0;
        if (XRING && xr && l + 1 < nl) {
            const MvLayer& Ln = net.L[l + 1];
            const int NTn = (l + 2 == nl) ? 1 : Ln.NT, pern = (NTn + NW - 1) / NW, ctn = w * pern;
            int ntwn = NTn - ctn; ntwn = ntwn < 0 ? 0 : (ntwn > pern ? pern : ntwn);
            if (ntwn == NTW && (Ln.KB & 3) == 0) { wn = Ln.wp + (size_t)ctn * Ln.KB * 64 + lane; KBn = Ln.KB; }
        }
        f32x4 acc[MTc][NTW];
        mv_zero_acc<MTc, NTW>(acc);
        float bv_[NTW];                                       // biases of this wave's columns: loaded now, consumed after the GEMM
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = (ct0 + t) * 16 + r;
            bv_[t] = (t < ntw && col < L.N) ? L.bias[col] : 0.0f;
        }
        MV_PH(7)
        mv_barrier_lds();                                     // inputs of layer l complete (LDS); the bias loads stay in flight
        MV_PH(1)
        if (XRING && xr) {
            mv_gemm_ring_x<MTc, NTW, NTW, 4>(L, act, S, ct0, acc, lane, bring, primed, wn, KBn);
            primed = (wn != nullptr);
        } else if (ntw > 0 && !(MV_ABLATE & 2)) mv_gemm_dispatch<MTc, NTW>(L, act, S, ct0, ntw, acc, lane);
        MV_PH(6)
        mv_barrier_lds();                                     // every wave done reading act (in-place update)
        MV_PH(3)
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
                        const int pos = (ct0 + t) * 16 + ((r & 3) << 2) + (r >> 2);
#pragma unroll
                        for (int a = 0; a < MTc; ++a)
#pragma unroll
                            for (int i = 0; i < 4; i += 2) {                      // two activations per packed-fp32 instruction
                                dm_f2 h = mv_act2(dm_f2{acc[a][t][i] + bv, acc[a][t][i + 1] + bv});   // Softplus(beta=100), idr.py:91-92
                                if (to_skip) h = h * dm2_s(0.7071067690849304f);  // cat([x, input]) / sqrt(2), idr.py:86-87 (dm_div_sqrt2)
                                act[(a * 16 + 4 * q + i) * S + pos] = h.x;
                                act[(a * 16 + 4 * q + i + 1) * S + pos] = h.y;
                            }
                    }
                }
            }
            const int Kn = net.L[l + 1].K, Kpn = net.L[l + 1].KB * 16;
            if (to_skip) {
                for (int idx = tid; idx < rows * d0; idx += NTHREADS) {
                    const int row = idx / d0, j = idx - row * d0;
                    act[row * S + mv_perm(N + j)] = dm_div_sqrt2(pe[row * d0 + j]);
                }
            }
            if (Kpn > Kn) {
                const int pad = Kpn - Kn;
                for (int idx = tid; idx < rows * pad; idx += NTHREADS) {
                    const int row = idx / pad, j = idx - row * pad;
                    act[row * S + mv_perm(Kn + j)] = 0.0f;
                }
            }
        }
        MV_PH(4)
    }
    __syncthreads();
    MV_PH(5)
    MV_PH_END
}
