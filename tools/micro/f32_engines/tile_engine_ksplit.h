// tile_engine_ksplit.h -- the K-split staggered evaluation that lived at the end of csrc/tile_engine.h until round 4 (mvsdf_sdf_col0 with mt = 17):
// measured 62-63 us per isolated 16-row evaluation against 58 us for the barrier engine, bit-exact.  Not in the product any more;
// to rebuild: #include after tile_engine.h, paste basic_hip_kernels.inc into csrc/basic.hip.
#pragma once
#include "tile_engine.h"

// ---------------------------------------------------------------------------------------------------------------
// K-split staggered evaluation of ONE 16-row tile by 8 waves (the latency regime of the sphere tracer and the secant chains: one tile
// per CU, a chain of dependent evaluations).  mv_sdf_eval_col0 runs every layer as barrier -> GEMM (all waves) -> barrier -> softplus
// (all waves): the matrix pipe idles during every epilogue, barrier and ring fill.  Here the waves form two groups, A = waves 0-3 and
// B = waves 4-7 (one wave of each on every SIMD); A owns the low column tiles of every layer, B the high ones.  Because the GEMM of
// layer l+1 consumes its k-blocks in ASCENDING order (the bit-exact fmaf chain), it needs A's columns of layer l first and B's columns
// only for its second half.  So the groups alternate on the matrix pipe:
//     A: GEMM(l) | softplus(l)           | GEMM(l+1)  | ...
//     B:  (wait) | GEMM(l)  | softplus(l)             | GEMM(l+1) ...
// and each group's epilogue (softplus, LDS writes, bias loads, ring fill of the next GEMM) hides under the other group's MFMAs.  No
// workgroup barriers inside: per-layer LDS counters (gemm done / activations published, per group), activations ping-pong between two
// buffers.  Same arithmetic in the same order as mv_sdf_eval_col0: bit-identical.
// The flags live in LDS: the accesses are made through explicit LDS (address space 3) pointers -- a generic pointer would compile to
// flat loads / flat atomics, which are tracked by vmcnt too: every poll would then wait for the weight prefetches in flight.
typedef __attribute__((address_space(3))) int mv_lds_int;
__device__ __forceinline__ void mv_ks_wait(const int* f, int target) {
    const volatile mv_lds_int* p = (const volatile mv_lds_int*)f;
    while (__builtin_amdgcn_readfirstlane(*p) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");                              // later LDS reads stay behind the wait
}
// The LDS operations of one wave are processed in issue order, so the counter update lands after the wave's earlier LDS writes; the
// asm only keeps the compiler from reordering (and drains this wave's LDS queue, NOT its outstanding global loads).
__device__ __forceinline__ void mv_ks_signal(int* f, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((mv_lds_int*)f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// mv_gemm_ring with a wait before the first LDS read of k-block kA (the weight loads of the ring keep flowing across the wait)
template <int NT, int NTW, int PD>
__device__ __forceinline__ void mv_gemm_ring_ks(const MvLayer& L, const float* __restrict__ act, int S, int ct0, f32x4 (&acc)[1][NTW], int lane,
                                                int kA, const int* flag0, const int* flagA) {
    const int KB = L.KB;
    const float4* __restrict__ wp = L.wp + (size_t)ct0 * KB * 64 + lane;
    const float* arow = act + (lane & 15) * S + 4 * (lane >> 4);
    float4 b[PD][NT], a[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + d) * 64];
    if (flag0) mv_ks_wait(flag0, 4);                            // inputs of k-blocks [0, kA) published (and the matrix pipe handed over)
    if (kA < PD && flagA) mv_ks_wait(flagA, 4);
#pragma unroll
    for (int d = 0; d < PD; ++d) a[d] = *(const float4*)(arow + d * 16);
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&a[d])[s], ((const float*)&b[d][t])[s], acc[0][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int kn = (kb0 + d + PD < KB) ? kb0 + d + PD : kb0 + d;
#pragma unroll
            for (int t = 0; t < NT; ++t) b[d][t] = wp[((size_t)t * KB + kn) * 64];
            if (kn == kA && kA >= PD && flagA) mv_ks_wait(flagA, 4);
            a[d] = *(const float4*)(arow + kn * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NTW>
__device__ __forceinline__ void mv_gemm_ks_dispatch(const MvLayer& L, const float* act, int S, int ct0, int ntw, f32x4 (&acc)[1][NTW], int lane, int kA,
                                                    const int* flag0, const int* flagA) {
    const bool deep = (L.KB & 3) == 0 && (kA & 3) == 0;
    if (!deep && (kA & 1)) kA = 0;                              // odd split point: wait for everything up front
#define MV_KS(NT_) do { if (deep) mv_gemm_ring_ks<NT_, NTW, 4>(L, act, S, ct0, acc, lane, kA, flag0, flagA); \
                        else mv_gemm_ring_ks<NT_, NTW, 2>(L, act, S, ct0, acc, lane, kA, flag0, flagA); } while (0)
    if (ntw == NTW) { MV_KS(NTW); return; }
    if (NTW >= 4 && ntw == 3) { MV_KS((NTW >= 4 ? 3 : 1)); return; }
    if (NTW >= 4 && ntw == 2) { MV_KS((NTW >= 4 ? 2 : 1)); return; }
    if (ntw == 1) { MV_KS(1); return; }
#undef MV_KS
}

// LDS: act0 / act1 [16][S] (ping-pong), pe [16][d0], pts [16][3], out [16], flags [4 * n_layers] ints.  512 threads.
template <int NTW>
__device__ void mv_sdf_eval_col0_ks(const MvNet& net, float* act0, float* act1, float* pe, const float* pts, float* out, int* flags, int tid) {
    constexpr int NTHREADS = 512, NW = 8;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int grp = w >> 2;                                     // 0: group A (low columns, leads), 1: group B
    const int S = net.S, d0 = 3 + 6 * net.multires, nl = net.n_layers;
    mv_pe_rows<NTHREADS>(pts, pe, act0, S, 16, net.multires, tid);
    if (tid < 4 * nl) flags[tid] = 0;
    __syncthreads();
    int per_prev = 0, n_prev = 0;
    for (int l = 0; l < nl; ++l) {
        const MvLayer& L = net.L[l];
        const bool last = (l == nl - 1);
        const int NT = last ? 1 : L.NT;
        const int per = (NT + NW - 1) / NW;
        const int ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        const float* xin = (l & 1) ? act1 : act0;
        float* xout = (l & 1) ? act0 : act1;
        int* fl = flags + 4 * l;                                // {A gemm done, B gemm done, A activations published, B activations published}
        const int* fp = fl - 4;                                 // previous layer's
        // k-blocks below kA hold only columns group A wrote in the previous epilogue (complete 16-blocks of its tiles)
        int kA = L.KB;
        if (l > 0) { const int ca = 64 * per_prev < n_prev ? 64 * per_prev : n_prev; kA = ca >> 4; if (kA > L.KB) kA = L.KB; }
        f32x4 acc[1][NTW];
        mv_zero_acc<1, NTW>(acc);
        float bv_[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = (ct0 + t) * 16 + r;
            bv_[t] = (t < ntw && col < L.N) ? L.bias[col] : 0.0f;
        }
        // Hand-over of the matrix pipe + first-half inputs: A waits for B's GEMM(l-1) and A's activations(l-1); B waits for A's GEMM(l)
        // (which itself waited for A's activations(l-1)).  Second-half inputs: B's activations(l-1).  Waves without column tiles in this
        // layer take the same entry wait, so that no wave runs ahead of the layer the others are reading (B writes the skip columns).
#ifndef MV_KS_EXCL
#define MV_KS_EXCL 1
#endif
        if (grp == 0) {
            if (l > 0 && (MV_KS_EXCL || ntw == 0 || kA == L.KB)) mv_ks_wait(fp + 1, 4);
            if (ntw > 0) mv_gemm_ks_dispatch<NTW>(L, xin, S, ct0, ntw, acc, lane, kA, l > 0 ? fp + 2 : nullptr, l > 0 ? fp + 3 : nullptr);
        } else {
            if (ntw > 0) mv_gemm_ks_dispatch<NTW>(L, xin, S, ct0, ntw, acc, lane, kA, (MV_KS_EXCL || l == 0) ? fl + 0 : fp + 2, l > 0 ? fp + 3 : nullptr);
            else mv_ks_wait(fl + 0, 4);
        }
        mv_ks_signal(fl + grp, lane);                           // this wave no longer needs the matrix pipe for layer l
        if (last) {
            if (w == 0 && r == 0) {
                const float b0 = bv_[0];
#pragma unroll
                for (int i = 0; i < 4; ++i) out[4 * q + i] = acc[0][0][i] + b0;
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
                        for (int i = 0; i < 4; i += 2) {
                            dm_f2 h = mv_act2(dm_f2{acc[0][t][i] + bv, acc[0][t][i + 1] + bv});
                            if (to_skip) h = h * dm2_s(0.7071067690849304f);
                            xout[(4 * q + i) * S + pos] = h.x;
                            xout[(4 * q + i + 1) * S + pos] = h.y;
                        }
                    }
                }
            }
            if (grp == 1) {                                     // skip concatenation / K padding: columns beyond N, published with group B's
                const int Kn = net.L[l + 1].K, Kpn = net.L[l + 1].KB * 16, gt = tid - 256;
                if (to_skip) {
                    for (int idx = gt; idx < 16 * d0; idx += 256) {
                        const int row = idx / d0, j = idx - row * d0;
                        xout[row * S + mv_perm(N + j)] = dm_div_sqrt2(pe[row * d0 + j]);
                    }
                }
                if (Kpn > Kn) {
                    const int pad = Kpn - Kn;
                    for (int idx = gt; idx < 16 * pad; idx += 256) {
                        const int row = idx / pad, j = idx - row * pad;
                        xout[row * S + mv_perm(Kn + j)] = 0.0f;
                    }
                }
            }
            mv_ks_signal(fl + 2 + grp, lane);                   // LDS operations of a wave complete in order: the writes above precede this add
        }
        per_prev = per; n_prev = L.N;
    }
    __syncthreads();
}
