// chain_x3.h -- the fused differentiable chains of the SDF network (layer_kernels.h: k_chain_fwd, k_chain_bwd / k_chain_bwd2) on gfx950's bf16 matrix
// cores in the THREE-TERM fp32 arithmetic of tile_engine_bf16s.h ("f32x3"): every fp32 weight and every fp32 activation / adjoint enters the matrix core
// as three bf16 terms (w = w0 + w1 + w2, a = a0 + a1 + a2 exactly), a Linear is the six exact products a_s w_j, s + j <= 2, of v_mfma_f32_16x16x32_bf16
// accumulated in fp32 -- fp32-accurate (measured closer to an fp64 evaluation than the k-ascending fmaf chain of v_mfma_f32_16x16x4_f32) at 2.6x the
// matrix rate.  Same passes, same saved tensors (the fp32 context / workspace layout of diff_mlp.hip), same formulas as the fp32 chains;
// what differs from them:
//   * the running tile lives in LDS as three bf16 term tiles [rows][S16] (a phase's inputs are split ONCE, by the phase that produces them);
//   * the accumulators come out of the matrix core with a row per lane and four consecutive columns per register quad (lane = (q, r): row r, columns
//     16 tile + 4 q ..+3), so a phase's epilogue already holds what the next phase's prologue multiplied in the fp32 chains: s_l = sigma_l . u_{l+1}
//     (normal chain) and zbar_l = sigma_l . hbar_{l+1} + zbar2_l (E.2) are formed there -- no separate prologue pass, one barrier pair per phase;
//   * Softplus(100) / sigmoid(100 z) by v_exp_f32 / v_log_f32 / v_rcp_f32 (mv_softplus100_acc1's form, absolute error <= 8e-9): these passes are compared with
//     the reference within a tolerance (tests/test_gpu_diff.py, tests/test_gpu_idr.py), not bit for bit against an instruction-level oracle like the tracer.
// Reference code being replaced: idr.py:77-107 (forward / gradient) and autograd's (double) backward of both.
#pragma once
#include "layer_kernels.h"
#include "tile_engine_bf16s.h"

typedef FwdArgsT<MvNetBf> FwdArgsX3;            // net / netT: three-term packs of W_l / W_l^T (mvsdf_pack_bf16x3_net's layout); S = bf16 elements per LDS row of one term tile
typedef ChainArgsT<MvNetBf> ChainArgsX3;

__device__ __forceinline__ void mv_softplus_sigmoid100_fast(float z, float* h, float* sg) {
    const float t = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);       // exp(-|100 z|)
    const float one_t = 1.0f + t;
    *h = fmaf(__builtin_amdgcn_logf(one_t), 0.0069314718246459961f, fmaxf(z, 0.0f));
    *sg = (z >= 0.0f ? 1.0f : t) * __builtin_amdgcn_rcpf(one_t);
}

// four consecutive floats of a row; `vec`: the address is 16-byte aligned and all four are inside the row (wave-uniform)
__device__ __forceinline__ f32x4 mv_ld4(const float* __restrict__ p, bool vec, int nvalid) {
    if (vec) return *(const f32x4*)p;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (nvalid > 0) v[0] = p[0];
    if (nvalid > 1) v[1] = p[1];
    if (nvalid > 2) v[2] = p[2];
    if (nvalid > 3) v[3] = p[3];
    return v;
}
__device__ __forceinline__ void mv_st4(float* __restrict__ p, f32x4 v, bool vec, int nvalid) {
    if (vec) { *(f32x4*)p = v; return; }
    if (nvalid > 0) p[0] = v[0];
    if (nvalid > 1) p[1] = v[1];
    if (nvalid > 2) p[2] = v[2];
    if (nvalid > 3) p[3] = v[3];
}

// four consecutive columns [col0, col0 + 4) of LDS row rr <- the three bf16 terms of v (columns >= nvalid are not written)
__device__ __forceinline__ void mv_x3_put4(uint16_t* act, int S16, int TS, int rr, int col0, f32x4 v, int nvalid) {
    uint32_t p0[3], p1[3];
    mv_split_pk<3>(dm_f2{v[0], v[1]}, p0);
    mv_split_pk<3>(dm_f2{v[2], v[3]}, p1);
    uint16_t* dst = act + rr * S16 + col0;
    if (nvalid >= 4) {
#pragma unroll
        for (int s = 0; s < 3; ++s) *(uint2*)(dst + s * TS) = uint2{p0[s], p1[s]};
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (nvalid > 0) dst[s * TS] = (uint16_t)p0[s];
            if (nvalid > 1) dst[s * TS + 1] = (uint16_t)(p0[s] >> 16);
            if (nvalid > 2) dst[s * TS + 2] = (uint16_t)p1[s];
        }
    }
}
__device__ __forceinline__ void mv_x3_put1(uint16_t* act, int S16, int TS, int rr, int col, float v) {
    uint16_t p[3];
    mv_split_1<3>(v, p);
#pragma unroll
    for (int s = 0; s < 3; ++s) act[s * TS + rr * S16 + col] = p[s];
}
// zero the columns [c0, c1) of every row and term (the k padding a phase's matrix instructions read)
template <int ROWS, int NTH>
__device__ __forceinline__ void mv_x3_zero_cols(uint16_t* act, int S16, int TS, int c0, int c1, int tid) {
    const int pad = c1 - c0;
    if (pad <= 0) return;
    for (int idx = tid; idx < ROWS * pad; idx += NTH) {
        const int rr = idx / pad, j = idx - rr * pad;
#pragma unroll
        for (int s = 0; s < 3; ++s) act[s * TS + rr * S16 + c0 + j] = 0;
    }
}

// this wave's column tiles of a phase: [ct0, ct0 + ntw)
#define MV_X3_TILES(L_) \
    const int NT = (L_).NT, per = (NT + NW - 1) / NW, ct0 = w * per; \
    int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw); ntw = ntw > NTW ? NTW : ntw;

template <int MT, int NTW>
__device__ __forceinline__ void mv_x3_gemm(const MvLayerBf& L, const uint16_t* act, int S16, int TS, int ct0, int ntw, f32x4 (&acc)[MT][NTW], int lane) {
    if (ntw > 0) mv_gemm_rolling_dispatch_bw<MT, NTW, 3, 3, (NTW == 1)>(L.KB, act, S16, TS, L.wp + (size_t)ct0 * L.KB * 3 * 64 + lane, ntw, acc, lane);
}

// Measured (tools/micro/chain_x3/fwd_probe.py, 3100 rows of the 8x256 network, one 16-row tile per CU).  With the rolling weight fetch a phase's matrix loop is the 393 KB
// weight stream at ~85 GB/s per CU (half of what the L2 delivers to the tracer's continuously running ring: a chain's ring drains at every phase end): 4.6 us per phase,
// 111 us per launch.  With tile_engine_bf16s.h's CARRIED ring (below) as 16 waves x 1 tile, four k-blocks deep: 3.4 us per phase, 97 us; two k-blocks 99, three 102; as 8
// waves x 2 tiles 103 (a wave's epilogue -- its loads / stores of the saved tensors -- is latency-bound per wave: 16 waves halve it).  Not kept: non-temporal stores of the
// saved tensors (112 -> 129 us: sigma_l is re-read by the normal chain of the same launch), non-temporal weight loads (164 us).  Without any store: 100 us.

// The weight fetch of a chain of phases.  PDW > 0 (hidden width <= 256, one column tile per wave): tile_engine_bf16s.h's CARRIED ring -- the first PDW k-blocks of
// the NEXT phase's weight fragments are requested by the slots of the current phase's matrix loop that have no k-block of their own left, and land under its
// epilogue: a phase starts with PDW of its KB k-blocks on the CU.
template <int MT, int NTW, int NW, int PDW>
struct MvX3Ring {
    static constexpr bool CARRY = PDW > 0;
    uint4 bw[CARRY ? PDW : 1][NTW][3];
    const uint4* wcur[NTW];
    const uint4* wnext[NTW];
    int kbnext;
    // (layers are handed over as (pack, k-blocks, column tiles), never as pointers into the kernel arguments: taking such an address makes the compiler copy
    // the whole argument block to scratch memory -- 1.5 KB per lane, and the kernel twice as slow)
    __device__ __forceinline__ void prep(const uint4* wp, int KB, int NT, int w, int lane) {
        const int per = (NT + NW - 1) / NW, c0 = w * per;
        kbnext = KB;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tile = c0 + t < NT ? c0 + t : NT - 1;
            wnext[t] = wp + (size_t)tile * kbnext * 3 * 64 + lane;
        }
    }
    __device__ __forceinline__ void fill() {                        // the very first phase: nothing to hide the request behind
        if constexpr (CARRY) {
#pragma unroll
            for (int d = 0; d < PDW; ++d) {
                const int kb = d < kbnext ? d : kbnext - 1;
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int j = 0; j < 3; ++j) bw[d][t][j] = wnext[t][((size_t)kb * 3 + j) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // one phase: acc += act x L (the layer prep()'d last); (nwp, nkb, nnt): the layer of the phase after it (nwp null: none -- the ring re-requests this layer's
    // k-block 0, unused)
    __device__ __forceinline__ void gemm(const MvLayerBf& L, const uint4* nwp, int nkb, int nnt, const uint16_t* act, int S16, int TS, int ct0, int ntw,
                                         f32x4 (&acc)[MT][NTW], int w, int lane) {
        if constexpr (CARRY) {
            const int KB = kbnext;
#pragma unroll
            for (int t = 0; t < NTW; ++t) wcur[t] = wnext[t];
            if (nwp) prep(nwp, nkb, nnt, w, lane); else kbnext = 1;
            mv_gemm_carried_bw<MT, NTW, NTW, PDW, 3, 3>(KB, act, S16, TS, wcur, acc, lane, bw, wnext, kbnext);
        } else {
            mv_x3_gemm<MT, NTW>(L, act, S16, TS, ct0, ntw, acc, lane);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Forward value + normal (k_chain_fwd's passes and outputs)
template <int MT, int NTW, int NW, int PDW = 0>
__global__ __launch_bounds__(64 * NW) void k_chain_fwd_x3(FwdArgsX3 a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = a.row_base + blockIdx.x * ROWS, S16 = a.S, TS = ROWS * S16, nl = a.net.n_layers, d0 = 3 + 6 * a.net.multires;
    const unsigned skm = a.net.skip_mask;
    uint16_t* act = (uint16_t*)smem;                                // three term tiles [ROWS][S16]
    float* pe = smem + (3 * TS) / 2;                                // [ROWS][d0] natural order (fp32)
    float* padj = pe + ((ROWS * d0 + 3) & ~3);                      // [ROWS][d0] PE adjoint of the skip layer(s), then g_0
    float* pts = padj + ((ROWS * d0 + 3) & ~3);                     // [ROWS][3]
    CH_PH_DECL
    MvX3Ring<MT, NTW, NW, PDW> ring;
    ring.prep(a.net.L[0].wp, a.net.L[0].KB, a.net.L[0].NT, w, lane);
    ring.fill();                                                    // layer 0's first k-blocks, requested before the positional encoding
    for (int i = tid; i < ROWS * 3; i += NTH) {
        const int row = row0 + i / 3, c = i - 3 * (i / 3);
        float v = 0.0f;
        if (row < a.M) {
            if (!a.g.pts) v = a.x[3 * (size_t)row0 + i];
            else {
                const int E = a.g.n_eik + 2 * a.g.n_ds;
                if (row < a.g.n_eik) v = a.g.eik[3 * (size_t)row + c];
                else if (row < a.g.n_eik + a.g.n_ds) v = a.g.on[3 * (size_t)(row - a.g.n_eik) + c];
                else if (row < E) v = a.g.jit[3 * (size_t)(row - a.g.n_eik - a.g.n_ds) + c];
                else v = a.g.pts[3 * (size_t)a.g.perm[row - E] + c];
                a.g.x_out[3 * (size_t)row + c] = v;
            }
        }
        pts[i] = v;
    }
    __syncthreads();
    mv_pe_rows_bs<NTH, 3, false>(pts, pe, act, S16, TS, ROWS, a.net.multires, a.net.L[0].KB * 32, tid);
    __syncthreads();
    for (int idx = tid; idx < ROWS * a.ld0; idx += NTH) {
        const int rr = idx / a.ld0, k = idx - rr * a.ld0, row = row0 + rr;
        if (row < a.M) a.H0[(size_t)row * a.ld0 + k] = k < d0 ? pe[rr * d0 + k] : 0.0f;
    }
    CH_PH(0)
    const bool top_skip = mv_skip_at(skm, nl - 1);
    f32x4 stop[MT][NTW];                                            // s_{L-2} = sigma_{L-2} . u_{L-1} of this lane's elements: the normal chain's first input
    // ---- value chain
    for (int l = 0; l < nl - 1; ++l) {
        const MvLayerBf& L = a.net.L[l];
        const int N = L.N;
        MV_X3_TILES(L)
        const bool to_skip = mv_skip_at(skm, l + 1), top = (l == nl - 2);
        const int Kn = a.net.L[l + 1].K, Kpn = a.net.L[l + 1].KB * 32;
        f32x4 acc[MT][NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tile = ct0 + t < NT ? ct0 + t : NT - 1;
            const f32x4 b4 = *(const f32x4*)(L.bias + tile * 16 + 4 * q);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][t] = b4;
        }
        mv_barrier_lds();
        CH_PH(1)
        // (after the last hidden layer the ring prefetches for the normal chain's first phase: the last Linear in between fetches its own)
        {
            const bool inner = l + 1 < nl - 1;
            const MvLayerBf& Ln = inner ? a.net.L[l + 1] : a.netT.L[nl - 2];
            ring.gemm(L, (inner || row0 < a.Mg) ? Ln.wp : nullptr, Ln.KB, Ln.NT, act, S16, TS, ct0, ntw, acc, w, lane);
        }
        CH_PH(2)
        mv_barrier_lds();
        CH_PH(3)
        const bool vz = (N & 3) == 0, va = (Kn & 3) == 0;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col0 = (ct0 + t) * 16 + 4 * q, nv = N - col0;             // valid columns of this lane's four
                f32x4 wl = {0.f, 0.f, 0.f, 0.f};
                if (top) {
                    wl = mv_ld4(a.w_last_row0 + col0, false, nv);
                    if (top_skip) { for (int i = 0; i < 4; ++i) wl[i] = dm_div_sqrt2(wl[i]); }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int rr = m * 16 + r, row = row0 + rr;
                    f32x4 h, sg;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float hh, ss;
                        mv_softplus_sigmoid100_fast(acc[m][t][i], &hh, &ss);
                        h[i] = to_skip ? dm_div_sqrt2(hh) : hh;
                        sg[i] = ss;
                    }
                    mv_x3_put4(act, S16, TS, rr, col0, h, nv);
                    if (row < a.M) {
                        mv_st4(a.Z[l] + (size_t)row * N + col0, sg, vz && nv >= 4, nv);
                        mv_st4(a.A[l + 1] + (size_t)row * Kn + col0, h, va && nv >= 4, nv);
                    }
                    if (top) {
                        f32x4 s4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) s4[i] = (row < a.Mg && i < nv) ? sg[i] * wl[i] : 0.0f;
                        stop[m][t] = s4;
                        if (row < a.Mg) mv_st4(a.Sg[l] + (size_t)row * N + col0, s4, vz && nv >= 4, nv);
                    }
                }
            }
        }
        if (to_skip)
            for (int idx = tid; idx < ROWS * d0; idx += NTH) {
                const int rr = idx / d0, j = idx - rr * d0, row = row0 + rr;
                const float v = dm_div_sqrt2(pe[rr * d0 + j]);
                mv_x3_put1(act, S16, TS, rr, N + j, v);
                if (row < a.M) a.A[l + 1][(size_t)row * Kn + N + j] = v;
            }
        mv_x3_zero_cols<ROWS, NTH>(act, S16, TS, Kn, Kpn, tid);
        CH_PH(4)
    }
    {   // last layer: every output column, groups of NTW column tiles per wave
        const MvLayerBf& L = a.net.L[nl - 1];
        const int N = L.N, NT = L.NT, per = (NT + NW - 1) / NW;
        mv_barrier_lds();
        for (int g0 = 0; g0 < per; g0 += NTW) {
            const int ct0 = w * per + g0;
            int ntw = min(per - g0, NT - ct0);
            ntw = ntw < 0 ? 0 : (ntw > NTW ? NTW : ntw);
            f32x4 acc[MT][NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                int tile = ct0 + t < NT ? ct0 + t : NT - 1;
                tile = tile < 0 ? 0 : tile;
                const f32x4 b4 = *(const f32x4*)(L.bias + tile * 16 + 4 * q);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][t] = b4;
            }
            if constexpr (PDW > 0 && NTW == 1) {                      // (two k-blocks in flight here: the carried ring's registers stay live across this layer)
                if (ntw > 0) mv_gemm_rolling_bw<MT, 1, NTW, 2, 3, 3>(L.KB, act, S16, TS, L.wp + (size_t)ct0 * L.KB * 3 * 64 + lane, acc, lane);
            } else mv_x3_gemm<MT, NTW>(L, act, S16, TS, ct0, ntw, acc, lane);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (t < ntw) {
                    const int col0 = (ct0 + t) * 16 + 4 * q, nv = N - col0;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int row = row0 + m * 16 + r;
                        if (row < a.M) mv_st4(a.y + (size_t)row * a.ldy + col0, acc[m][t], false, nv);
                    }
                }
            }
        }
    }
    CH_PH(5)
    if (row0 >= a.Mg) return;                                       // workgroup-uniform: no normals for these rows
    // ---- normal chain (rows >= Mg inside the tile carry zeros).  u_L = W_L[0, :]; with a skip into the last Linear its PE part starts the PE adjoint.
    __syncthreads();                                                // every wave done reading the last layer's input -- and (vmcnt(0)) every sigma_l of this tile stored: the
                                                                    // normal chain reads them back (each lane its own elements, but nothing here should rest on that)
    {
        const MvLayerBf& L = a.net.L[nl - 2];
        MV_X3_TILES(L)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            if (t < ntw) {
                const int col0 = (ct0 + t) * 16 + 4 * q;
#pragma unroll
                for (int m = 0; m < MT; ++m) mv_x3_put4(act, S16, TS, m * 16 + r, col0, stop[m][t], L.N - col0);
            }
        mv_x3_zero_cols<ROWS, NTH>(act, S16, TS, L.N, a.netT.L[nl - 2].KB * 32, tid);
    }
    for (int i = tid; i < ROWS * d0; i += NTH) padj[i] = top_skip ? dm_div_sqrt2(a.w_last_row0[a.net.L[nl - 1].K - d0 + (i % d0)]) : 0.0f;
    CH_PH(6)
    for (int l = nl - 2; l >= 0; --l) {
        const MvLayerBf& L = a.netT.L[l];                           // contraction over out_l (K), produces in_l columns (N)
        const int N = L.N;
        MV_X3_TILES(L)
        const bool sk = mv_skip_at(skm, l);
        const int Nh = sk ? N - d0 : N;                             // the hidden part = out_{l-1} columns (l > 0)
        const bool vh = (Nh & 3) == 0;
        // sigma_{l-1} of this lane's output elements: requested before the matrix instructions, consumed after them (clamped, unconditional)
        f32x4 zz[MT][NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col0 = (ct0 + t) * 16 + 4 * q, nv = Nh - col0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = row0 + m * 16 + r;
                const bool ok = l > 0 && t < ntw && row < a.Mg && nv > 0;
                zz[m][t] = mv_ld4(a.Z[l > 0 ? l - 1 : 0] + (ok ? (size_t)row * Nh + col0 : 0), ok && vh && nv >= 4, ok ? nv : 0);
            }
        }
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        CH_PH(7)
        mv_barrier_lds();
        CH_PH(8)
        {
            const MvLayerBf& Ln = a.netT.L[l > 0 ? l - 1 : 0];
            ring.gemm(L, l > 0 ? Ln.wp : nullptr, Ln.KB, Ln.NT, act, S16, TS, ct0, ntw, acc, w, lane);
        }
        CH_PH(9)
        mv_barrier_lds();
        CH_PH(10)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col0 = (ct0 + t) * 16 + 4 * q;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int rr = m * 16 + r, row = row0 + rr;
                    f32x4 v = acc[m][t];
                    if (sk) { for (int i = 0; i < 4; ++i) v[i] = dm_div_sqrt2(v[i]); }
                    if (l == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int col = col0 + i;
                            if (col < N) {
                                const float g = padj[rr * d0 + col] + v[i];
                                padj[rr * d0 + col] = g;
                                if (row < a.Mg) a.G0[(size_t)row * a.ld0 + col] = g;
                            }
                        }
                    } else {
                        const int nv = Nh - col0;                       // hidden columns among this lane's four (<= 0: all of them are PE columns)
                        if (nv > 0) {
                            f32x4 s4;
#pragma unroll
                            for (int i = 0; i < 4; ++i) s4[i] = (row < a.Mg && i < nv) ? zz[m][t][i] * v[i] : 0.0f;
                            mv_x3_put4(act, S16, TS, rr, col0, s4, nv);
                            if (row < a.Mg) {
                                mv_st4(a.U[l] + (size_t)row * Nh + col0, v, vh && nv >= 4, nv);
                                mv_st4(a.Sg[l - 1] + (size_t)row * Nh + col0, s4, vh && nv >= 4, nv);
                            }
                        }
                        if (sk) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int col = col0 + i;
                                if (col >= Nh && col < N) padj[rr * d0 + (col - Nh)] += v[i];
                            }
                        }
                    }
                }
            }
        }
        if (l > 0) mv_x3_zero_cols<ROWS, NTH>(act, S16, TS, Nh, a.netT.L[l - 1].KB * 32, tid);
        CH_PH(11)
    }
    __syncthreads();
    for (int idx = tid; idx < ROWS * 3; idx += NTH) {               // n = J_PE^T g_0
        const int rr = idx / 3, c = idx - 3 * rr, row = row0 + rr;
        if (row >= a.Mg) continue;
        const float* h = pe + rr * d0;
        const float* g = padj + rr * d0;
        float v = g[c];
        for (int m = 0; m < a.net.multires; ++m) {
            const float f = (float)(1 << m);
            v += f * (h[6 + 6 * m + c] * g[3 + 6 * m + c] - h[3 + 6 * m + c] * g[6 + 6 * m + c]);
        }
        a.nrm[(size_t)row * 3 + c] = v;
    }
    CH_PH(12)
    CH_PH_END
}

// ---------------------------------------------------------------------------------------------------------------
// One backward pass per row tile (mv_chain_bwd_body's passes and outputs): gbar_0 = J_PE nbar, the ascending E.1 chain, the descending E.2 chain, the
// input adjoint.  dn_in == NULL: E.2 only.
template <int MT, int NTW, int NW, int PDW>
__device__ __forceinline__ void mv_chain_bwd_body_x3(const ChainArgsX3& a, int blk, float* smem) {
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    constexpr bool PF = MT * NTW <= 4;                              // side inputs of an epilogue requested before the phase's matrix instructions
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = blk * ROWS, S16 = a.S, TS = ROWS * S16, nl = a.net.n_layers, d0 = 3 + 6 * a.net.multires;
    const unsigned skm = a.net.skip_mask;
    const int Mr = mv_chain_rows(a);                             // (deferred step: the true row count comes from device memory, the grid covers its upper bound)
    if (row0 >= Mr) return;
    uint16_t* act = (uint16_t*)smem;
    float* g0s = smem + (3 * TS) / 2;                               // [ROWS][d0]: gbar_0 (E.1), then the PE adjoint (E.2)
    float* pe_adj = g0s;
    const bool top_skip = mv_skip_at(skm, nl - 1);
    MvX3Ring<MT, NTW, NW, PDW> ring;
    {
        const MvLayerBf& L0 = a.dn_in ? a.net.L[0] : a.netT.L[nl - 1];
        ring.prep(L0.wp, L0.KB, L0.NT, w, lane);
        ring.fill();                                                // the first phase's first k-blocks, requested before its input is built
    }
    if (a.dn_in) {
        {   // gbar_0 = J_PE nbar
            const int Kp0 = a.net.L[0].KB * 32;
            for (int idx = tid; idx < ROWS * Kp0; idx += NTH) {
                const int rr = idx / Kp0, k = idx - rr * Kp0, row = row0 + rr;
                float v = 0.0f;
                if (row < Mr && k < d0) {
                    const float* h = a.H0 + (size_t)row * a.row_ld0;
                    const float* nb = a.dn_in + (size_t)row * 3;
                    if (k < 3) v = nb[k];
                    else {
                        const int jj = k - 3, m = jj / 6, rem = jj - 6 * m, c = rem % 3;
                        const float f = (float)(1 << m);
                        v = rem < 3 ? f * h[6 + 6 * m + c] * nb[c] : -f * h[3 + 6 * m + c] * nb[c];
                    }
                }
                if (row < Mr && k < a.row_ld0) a.VB0w[(size_t)row * a.row_ld0 + k] = v;    // vbar_0 for the weight gradient
                mv_x3_put1(act, S16, TS, rr, k, v);
                if (k < d0) g0s[rr * d0 + k] = v;
            }
        }
        for (int l = 0; l < nl - 1; ++l) {
            const MvLayerBf& L = a.net.L[l];
            const int N = L.N;
            MV_X3_TILES(L)
            const bool top = (l == nl - 2), to_skip = mv_skip_at(skm, l + 1);
            const bool vn = (N & 3) == 0;
            const int ldn = a.net.L[l + 1].K;                        // row length of vbar_{l+1}
            const bool vv = (ldn & 3) == 0;
            // side inputs of the epilogue (sigma_l, u_{l+1}): requested BEFORE the matrix instructions (PF), or -- the 2-row-tile x 4-column-tile wide form,
            // whose accumulators and weight ring leave no registers for them -- inside the epilogue
            f32x4 zz[PF ? MT : 1][PF ? NTW : 1], uu[PF ? MT : 1][PF ? NTW : 1];
            auto side = [&](int t, int m, f32x4& z, f32x4& u) {
                const int col0 = (ct0 + t) * 16 + 4 * q, nv = N - col0;
                const int row = row0 + m * 16 + r;
                const bool ok = t < ntw && row < Mr && nv > 0;
                z = mv_ld4(a.Z[l] + (ok ? (size_t)row * N + col0 : 0), ok && vn && nv >= 4, ok ? nv : 0);
                if (top) {
                    f32x4 wl = mv_ld4(a.w_last_row0 + (ok ? col0 : 0), false, ok ? nv : 0);
                    if (to_skip) { for (int i = 0; i < 4; ++i) wl[i] = dm_div_sqrt2(wl[i]); }
                    u = wl;
                } else u = mv_ld4(a.U[l + 1] + (ok ? (size_t)row * N + col0 : 0), ok && vn && nv >= 4, ok ? nv : 0);
            };
            if constexpr (PF) {
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m) side(t, m, zz[m][t], uu[m][t]);
            }
            f32x4 acc[MT][NTW];
            mv_zero_acc<MT, NTW>(acc);
            mv_barrier_lds();
            {
                const MvLayerBf& Ln = l + 1 < nl - 1 ? a.net.L[l + 1] : a.netT.L[nl - 1];     // (after E.1's last phase: E.2's first)
                ring.gemm(L, Ln.wp, Ln.KB, Ln.NT, act, S16, TS, ct0, ntw, acc, w, lane);
            }
            mv_barrier_lds();
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (t < ntw) {
                    const int col0 = (ct0 + t) * 16 + 4 * q, nv = N - col0;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int rr = m * 16 + r, row = row0 + rr;
                        f32x4 ub, z2, zs, us;
                        if constexpr (PF) { zs = zz[m][t]; us = uu[m][t]; } else side(t, m, zs, us);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float sb = acc[m][t][i], sig = zs[i];
                            const float u = sig * sb;
                            ub[i] = to_skip ? dm_div_sqrt2(u) : u;
                            z2[i] = us[i] * sb * dm_sigmoid_prime100(sig);
                        }
                        mv_x3_put4(act, S16, TS, rr, col0, ub, nv);
                        if (row < Mr && nv > 0) {
                            mv_st4(a.VB[l + 1] + (size_t)row * ldn + col0, ub, vv && nv >= 4, nv);
                            mv_st4(a.ZB2o[l] + (size_t)row * N + col0, z2, vn && nv >= 4, nv);
                        }
                    }
                }
            }
            if (to_skip)
                for (int idx = tid; idx < ROWS * d0; idx += NTH) {
                    const int rr = idx / d0, j = idx - rr * d0, row = row0 + rr;
                    const float tv = dm_div_sqrt2(g0s[rr * d0 + j]);
                    mv_x3_put1(act, S16, TS, rr, N + j, tv);
                    if (row < Mr) a.VB[l + 1][(size_t)row * ldn + N + j] = tv;     // PE tail of the skip layer's vbar
                }
            mv_x3_zero_cols<ROWS, NTH>(act, S16, TS, ldn, a.net.L[l + 1].KB * 32, tid);
        }
        __syncthreads();                                            // zbar2 of this tile written (global) before E.2 reads it
    }
    for (int i = tid; i < ROWS * d0; i += NTH) pe_adj[i] = 0.0f;
    {   // E.2's first input: dy (or the delta pass's one scalar per row on output column 0)
        const int K = a.netT.L[nl - 1].K, Kp = a.netT.L[nl - 1].KB * 32;
        for (int idx = tid; idx < ROWS * Kp; idx += NTH) {
            const int rr = idx / Kp, k = idx - rr * Kp, row = row0 + rr;
            float v = 0.0f;
            if (row < Mr && k < K) v = a.dy_col0 ? (k == 0 ? a.dy_col0[row] : 0.0f) : a.dy[(size_t)row * a.ld_dy + k];
            mv_x3_put1(act, S16, TS, rr, k, v);
        }
    }
    for (int l = nl - 1; l >= 0; --l) {
        const MvLayerBf& L = a.netT.L[l];                           // contraction over out_l (K), produces in_l columns (N)
        const int N = L.N;
        MV_X3_TILES(L)
        const bool sk = mv_skip_at(skm, l);
        const int Nh = sk ? N - d0 : N;
        const bool vh = (Nh & 3) == 0;
        // sigma_{l-1}, zbar2_{l-1} (and, accumulating, zbar_{l-1}) of this lane's output elements: requested before the matrix instructions
        f32x4 zz[PF ? MT : 1][PF ? NTW : 1], z2[PF ? MT : 1][PF ? NTW : 1], zo[PF ? MT : 1][PF ? NTW : 1];
        const int lm = l > 0 ? l - 1 : 0;
        const bool has2 = a.ZB2[lm] != nullptr;
        auto side = [&](int t, int m, f32x4& z, f32x4& zb2, f32x4& zold) {
            const int col0 = (ct0 + t) * 16 + 4 * q, nv = Nh - col0;
            const int row = row0 + m * 16 + r;
            const bool ok = l > 0 && t < ntw && row < Mr && nv > 0;
            const size_t off = ok ? (size_t)row * Nh + col0 : 0;
            z = mv_ld4(a.Z[lm] + off, ok && vh && nv >= 4, ok ? nv : 0);
            zb2 = mv_ld4((has2 ? a.ZB2[lm] : a.Z[lm]) + off, ok && vh && nv >= 4, ok ? nv : 0);
            zold = a.accum ? mv_ld4(a.ZB[lm] + off, ok && vh && nv >= 4, ok ? nv : 0) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        if constexpr (PF) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) side(t, m, zz[m][t], z2[m][t], zo[m][t]);
        }
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        mv_barrier_lds();
        {
            const MvLayerBf& Ln = a.netT.L[l > 0 ? l - 1 : 0];
            ring.gemm(L, l > 0 ? Ln.wp : nullptr, Ln.KB, Ln.NT, act, S16, TS, ct0, ntw, acc, w, lane);
        }
        mv_barrier_lds();
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col0 = (ct0 + t) * 16 + 4 * q;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int rr = m * 16 + r, row = row0 + rr;
                    f32x4 v = acc[m][t];
                    if (sk) { for (int i = 0; i < 4; ++i) v[i] = dm_div_sqrt2(v[i]); }
                    if (l == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int col = col0 + i;
                            if (col < N) {
                                const float hb0 = pe_adj[rr * d0 + col] + v[i];
                                pe_adj[rr * d0 + col] = hb0;                       // kept for the input adjoint below
                                if (row < Mr && a.H0B) a.H0B[(size_t)row * a.row_ld0 + col] = hb0;
                            }
                        }
                    } else {
                        const int nv = Nh - col0;
                        if (nv > 0) {
                            f32x4 zb, zs, z2s, zos;
                            if constexpr (PF) { zs = zz[m][t]; z2s = z2[m][t]; zos = zo[m][t]; } else side(t, m, zs, z2s, zos);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float x = zs[i] * v[i];
                                if (has2) x += z2s[i];
                                zb[i] = (row < Mr && i < nv) ? x : 0.0f;
                            }
                            mv_x3_put4(act, S16, TS, rr, col0, zb, nv);             // the chain continues with THIS pass's zbar
                            if (row < Mr) {
                                f32x4 st = zb;
                                if (a.accum) { for (int i = 0; i < 4; ++i) st[i] = zos[i] + zb[i]; }
                                mv_st4(a.ZB[l - 1] + (size_t)row * Nh + col0, st, vh && nv >= 4, nv);
                            }
                        }
                        if (sk) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int col = col0 + i;
                                if (col >= Nh && col < N) pe_adj[rr * d0 + (col - Nh)] += v[i];
                            }
                        }
                    }
                }
            }
        }
        if (l > 0) mv_x3_zero_cols<ROWS, NTH>(act, S16, TS, Nh, a.netT.L[l - 1].KB * 32, tid);
    }
    if (a.dx) {
        __syncthreads();
        for (int idx = tid; idx < ROWS * 3; idx += NTH) {           // xbar = J_PE^T hbar_0 + sum_k PE''_k g0[k] nbar[c(k)]
            const int rr = idx / 3, c = idx - 3 * rr, row = row0 + rr;
            if (row >= Mr) continue;
            const float* h = a.H0 + (size_t)row * a.row_ld0;
            const float* hb = pe_adj + rr * d0;
            float v = hb[c], second = 0.0f;
            for (int m = 0; m < a.net.multires; ++m) {
                const float f = (float)(1 << m);
                const float sn = h[3 + 6 * m + c], co = h[6 + 6 * m + c];
                v += f * (co * hb[3 + 6 * m + c] - sn * hb[6 + 6 * m + c]);
                if (a.dn_in) {
                    const float* g = a.G0 + (size_t)row * a.row_ld0;
                    second -= f * f * (sn * g[3 + 6 * m + c] + co * g[6 + 6 * m + c]);
                }
            }
            if (a.dn_in) v += second * a.dn_in[(size_t)row * 3 + c];
            a.dx[(size_t)row * 3 + c] = v;
        }
    }
}

template <int MT, int NTW, int NW, int PDW = 0>
__global__ __launch_bounds__(64 * NW) void k_chain_bwd_x3(ChainArgsX3 a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    mv_chain_bwd_body_x3<MT, NTW, NW, PDW>(a, blockIdx.x, smem);
}
static_assert(2 * sizeof(ChainArgsX3) + 16 <= 4096, "kernel arguments of k_chain_bwd2_x3 exceed 4 KiB");
template <int MT, int NTW, int NW, int PDW = 0>
__global__ __launch_bounds__(64 * NW) void k_chain_bwd2_x3(ChainArgsX3 a, ChainArgsX3 b, int na) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < na) mv_chain_bwd_body_x3<MT, NTW, NW, PDW>(a, blockIdx.x, smem);
    else mv_chain_bwd_body_x3<MT, NTW, NW, PDW>(b, blockIdx.x - na, smem);
}
