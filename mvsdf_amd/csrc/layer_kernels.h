// layer_kernels.h -- the differentiable MLP passes as per-layer MFMA kernels built on tile_engine.h.
//
//   k_layer<PRO, EPI, MT> : out[rows, N] = EPI( PRO(inputs)[rows, K] x Wpacked )   for one Linear (or its transpose),
//                           16*MT rows per workgroup; the prologue functor builds the A tile in LDS from global
//                           tensors (fusing sigmoid/softplus-derivative products), the epilogue functor consumes the
//                           MFMA accumulators (bias, softplus, masks, splits at the skip connection) and stores.
//   k_wgrad_net / k_reduce_net : dW[No, Ki] = sum_rows P[row, No]^T Q[row, Ki]  (+ column sums of P for the bias) of EVERY layer,
//                           split over row chunks into slabs, reduced in a fixed order (deterministic).
//   k_chain_* / k_render_chain_* : whole forward / backward chains of a row tile in one launch (the running tile stays in LDS).
//
// Formulas: SURVEY.md Appendix E (value + normal forward, first- and second-order backward), verified there
// against torch.autograd double backward.  Reference code being replaced: idr.py:77-107 (forward / gradient),
// idr.py:145-167 (rendering net) and autograd's backward of both.
#pragma once
#include "tile_engine.h"
#include "step_internal.h"

enum {
    PRO_PLAIN = 0,      // A[row][k]
    PRO_SIG_MUL,        // Z[row][k] * U[row][k], Z = the saved sigma_l = sigmoid100(z_l)   (s_l = sigma_l . u_{l+1})
    PRO_SIG_BCAST,      // Z[row][k] * bcast[k]                         (l = 7: u_8 = W_8[0, :])
    PRO_ZBAR,           // zb = Z*U + (row < Mg ? A[row][k] : 0); also stored to out2     (E.2)
    PRO_TANH_BWD        // dz = A[row][k] * (1 - U[row][k]^2); also stored to out2
};
enum {
    EPI_SOFTPLUS = 0,   // z = acc + b;  sigmoid100(z) -> out1 ("Z": the backward and the normal chain need sigma, never z itself);
                        // h = softplus100(z) (/sqrt2 if skip_next) -> out0 (+ PE tail if skip_next)
    EPI_BIAS,           // out0 = acc + b
    EPI_SPLIT,          // col < csplit: out0[row][col] = f(acc) ; else out1[row][col - csplit] = f(acc); f = /sqrt2 if scale; + add[row][col] if add
    EPI_SBAR,           // sbar = acc: out0 = sigma*sbar (/sqrt2 if skip_next), out1 = u*sbar*sigma'            (E.1)
    EPI_RELU,           // out0 = max(acc + b, 0)
    EPI_TANH,           // out0 = tanh(acc + b)
    EPI_RELU_MASK       // out0 = acc * (add[row][col] > 0)
};

struct LayerArgs {
    MvLayer L;                 // packed weights (W or W^T pack) + bias
    int S;                     // LDS row stride
    int M, Mg;                 // rows; rows that carry second-order terms (PRO_ZBAR)
    const float* A; int lda;
    const float* Z; int ldz;
    const float* U; int ldu;
    const float* bcast; int bcast_sqrt2;   // bcast_sqrt2: the last Linear is a skip layer, u_L reaches sigma_{L-1} as W_L[0, :N] / sqrt(2)
    const float* add; int ldadd;
    float* out0; int ld0;
    float* out1; int ld1;
    float* out2; int ld2;
    int csplit;                // EPI_SPLIT
    int scale_sqrt2;           // divide by sqrt(2)
    int skip_next;             // EPI_SOFTPLUS / EPI_SBAR: next layer is the skip layer
    int d0;                    // PE width (skip tail) / view-PE multires for PRO_RENDER_IN
    const float* pe; int ldpe; // PE rows for the skip tail
};

template <int PRO>
__device__ __forceinline__ float mv_prologue(const LayerArgs& a, int row, int k) {
    if (PRO == PRO_PLAIN) return a.A[(size_t)row * a.lda + k];
    if (PRO == PRO_SIG_MUL || PRO == PRO_SIG_BCAST) {
        const float u = PRO == PRO_SIG_MUL ? a.U[(size_t)row * a.ldu + k] : (a.bcast_sqrt2 ? dm_div_sqrt2(a.bcast[k]) : a.bcast[k]);
        const float sv = a.Z[(size_t)row * a.ldz + k] * u;
        if (a.out2) a.out2[(size_t)row * a.ld2 + k] = sv;                       // s_l, kept for the weight gradient (E.1)
        return sv;
    }
    if (PRO == PRO_ZBAR) {
        float zb = a.Z[(size_t)row * a.ldz + k] * a.U[(size_t)row * a.ldu + k];
        if (row < a.Mg && a.A) zb += a.A[(size_t)row * a.lda + k];
        a.out2[(size_t)row * a.ld2 + k] = zb;
        return zb;
    }
    if (PRO == PRO_TANH_BWD) {
        const float y = a.U[(size_t)row * a.ldu + k];
        const float dz = a.A[(size_t)row * a.lda + k] * (1.0f - y * y);
        a.out2[(size_t)row * a.ld2 + k] = dz;
        return dz;
    }
    return 0.0f;
}

// side inputs of an epilogue element, loaded for ALL of a lane's outputs before any store (stores to possibly-aliasing
// pointers would otherwise serialise every load behind the previous store)
struct EpiIn { float z, u, add; };
template <int EPI>
__device__ __forceinline__ EpiIn mv_epilogue_load(const LayerArgs& a, int row, int col) {
    EpiIn e = {0.f, 0.f, 0.f};
    if (EPI == EPI_SPLIT) { if (a.add) e.add = a.add[(size_t)row * a.ldadd + col]; }
    else if (EPI == EPI_SBAR) { e.z = a.Z[(size_t)row * a.ldz + col]; e.u = a.U ? a.U[(size_t)row * a.ldu + col] : (a.bcast_sqrt2 ? dm_div_sqrt2(a.bcast[col]) : a.bcast[col]); }
    else if (EPI == EPI_RELU_MASK) e.add = a.add[(size_t)row * a.ldadd + col];
    return e;
}

template <int EPI>
__device__ __forceinline__ void mv_epilogue(const LayerArgs& a, int row, int col, float acc, const EpiIn& in) {
    if (EPI == EPI_SOFTPLUS) {
        const float z = acc + a.L.bias[col];
        float h, sg;
        dm_softplus_sigmoid100(z, &h, &sg);
        a.out1[(size_t)row * a.ld1 + col] = sg;
        if (a.skip_next) h = dm_div_sqrt2(h);
        a.out0[(size_t)row * a.ld0 + col] = h;
    } else if (EPI == EPI_BIAS) {
        a.out0[(size_t)row * a.ld0 + col] = acc + a.L.bias[col];
    } else if (EPI == EPI_SPLIT) {
        float v = acc;
        if (a.add) v += in.add;
        if (a.scale_sqrt2) v = dm_div_sqrt2(v);
        if (col < a.csplit) a.out0[(size_t)row * a.ld0 + col] = v;
        else a.out1[(size_t)row * a.ld1 + (col - a.csplit)] = v;
    } else if (EPI == EPI_SBAR) {
        const float sig = in.z;                                                 // the saved sigma
        const float u = in.u;
        float ub = sig * acc;
        if (a.skip_next) ub = dm_div_sqrt2(ub);
        a.out0[(size_t)row * a.ld0 + col] = ub;
        a.out1[(size_t)row * a.ld1 + col] = u * acc * dm_sigmoid_prime100(sig);
    } else if (EPI == EPI_RELU) {
        a.out0[(size_t)row * a.ld0 + col] = fmaxf(acc + a.L.bias[col], 0.0f);
    } else if (EPI == EPI_TANH) {
        a.out0[(size_t)row * a.ld0 + col] = tanhf(acc + a.L.bias[col]);
    } else if (EPI == EPI_RELU_MASK) {
        a.out0[(size_t)row * a.ld0 + col] = in.add > 0.0f ? acc : 0.0f;
    }
}

template <int PRO, int EPI, int MT, int NTW>
__global__ __launch_bounds__(MV_THREADS) void k_layer(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * ROWS, S = a.S, K = a.L.K, Kp = a.L.KB * 16, N = a.L.N;
    float* act = smem;
    for (int base = 0; base < ROWS * Kp; base += MV_THREADS * 8) {          // 8 independent elements per thread in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * MV_THREADS + tid;
            const int rr = idx / Kp, k = idx - rr * Kp, row = row0 + rr;
            v[u] = 0.0f;
            if (idx < ROWS * Kp && row < a.M && k < K) v[u] = mv_prologue<PRO>(a, row, k);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * MV_THREADS + tid;
            const int rr = idx / Kp, k = idx - rr * Kp;
            if (idx < ROWS * Kp) act[rr * S + mv_perm(k)] = v[u];
        }
    }
    __syncthreads();
    const int NT = a.L.NT;
    const int per = (NT + 3) >> 2;
    for (int g0 = 0; g0 < per; g0 += NTW) {                    // column-tile groups of NTW per wave (wide layers)
        const int ct0 = w * per + g0;
        int ntw = min(per - g0, NT - ct0);
        ntw = ntw < 0 ? 0 : (ntw > NTW ? NTW : ntw);
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        if (ntw > 0) mv_gemm_dispatch<MT, NTW>(a.L, act, S, ct0, ntw, acc, lane);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col = (ct0 + t) * 16 + r;
                if (col < N) {
                    EpiIn in[MT][4];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = row0 + m * 16 + 4 * q + i;
                            in[m][i] = EpiIn{0.f, 0.f, 0.f};
                            if (row < a.M) in[m][i] = mv_epilogue_load<EPI>(a, row, col);
                        }
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = row0 + m * 16 + 4 * q + i;
                            if (row < a.M) mv_epilogue<EPI>(a, row, col, acc[m][t][i], in[m][i]);
                        }
                }
            }
        }
    }
    if (EPI == EPI_SOFTPLUS || EPI == EPI_SBAR) {
        if (EPI == EPI_SOFTPLUS && a.skip_next) {              // a_{l+1}[:, N:N+d0] = PE / sqrt(2)   (idr.py:86-87)
            for (int idx = tid; idx < ROWS * a.d0; idx += MV_THREADS) {
                const int rr = idx / a.d0, j = idx - rr * a.d0, row = row0 + rr;
                if (row < a.M) a.out0[(size_t)row * a.ld0 + N + j] = dm_div_sqrt2(a.pe[(size_t)row * a.ldpe + j]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients dW[No][Ki] = P1^T Q1 (+ P2^T Q2) are GEMMs whose contraction runs over ROWS: split over 128-row chunks, partial 64x64 blocks go
// to slabs that a second launch sums in a fixed order (deterministic, no atomics); tiles are staged through LDS with 16-byte loads (wg_stage).
// The kernels are the network-wide k_wgrad_net / k_reduce_net below.
__device__ __forceinline__ void wg_stage(const float* __restrict__ src, int ld, int row_lo, int row_hi, int c0, int ncols, float* __restrict__ dst,
                                         int LD, int tid) {
    // 64 rows x 64 cols -> LDS; 16-byte global loads when the source allows it
    const bool vec = ((ld & 3) == 0) && ((c0 & 3) == 0) && ((((size_t)src) & 15) == 0);
    if (vec) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = u * MV_THREADS + tid, rr = idx >> 4, c4 = (idx & 15) * 4, row = row_lo + rr;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < row_hi) {
                if (c0 + c4 + 3 < ncols) v[u] = *(const float4*)(src + (size_t)row * ld + c0 + c4);
                else {
                    float t[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int e = 0; e < 4; ++e) if (c0 + c4 + e < ncols) t[e] = src[(size_t)row * ld + c0 + c4 + e];
                    v[u] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = u * MV_THREADS + tid, rr = idx >> 4, c4 = (idx & 15) * 4;
            *(float4*)(dst + rr * LD + c4) = v[u];
        }
    } else {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = u * MV_THREADS + tid, rr = idx >> 6, c = idx & 63, row = row_lo + rr;
            v[u] = (row < row_hi && c0 + c < ncols) ? src[(size_t)row * ld + c0 + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = u * MV_THREADS + tid, rr = idx >> 6, c = idx & 63;
            dst[rr * LD + c] = v[u];
        }
    }
}

// dev probe (side build with -DMV_CHAIN_PROBE, tools/chain_probe.py): wave w of workgroup 0 accumulates the 100 MHz clock between the marks of
// k_chain_fwd (tools/chain_probe.py) or of one 256 x 256 workgroup of k_wgrad_net (tools/chain_probe.py wgrad) into g_chain_ph[w][mark]; nothing in
// the product build
#ifdef MV_CHAIN_PROBE
__device__ unsigned long long g_chain_ph[16][16];
#define CH_PH_DECL unsigned long long chp_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long cht_ = wall_clock64();
#define CH_PH(i) { const unsigned long long t_ = wall_clock64(); chp_[i] += t_ - cht_; cht_ = t_; }
#define CH_PH_END if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_chain_ph[threadIdx.x >> 6][i_], chp_[i_]); }
#define WG_PH_END(blk) if ((int)blockIdx.x == (blk) && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_chain_ph[8 + (threadIdx.x >> 6)][i_], chp_[i_]); }
#else
#define CH_PH_DECL
#define CH_PH(i)
#define CH_PH_END
#define WG_PH_END(blk)
#endif

// ---- network-wide variants: the weight gradients of EVERY layer in one launch, then one reduction launch ----
// Workgroup = (layer, 128-row chunk, 64x64 output block); both pairs of a layer (E.2 term zbar^T a and E.1 term s^T vbar) accumulate
// in the same registers, so a layer needs one slab per chunk.
#define MV_WG_MAXL 16                      // layers of one launch: both networks of a training step (9 + 5)
// rows per weight-gradient chunk (one slab per chunk and layer).  128 -> 256: k_wgrad_net unchanged, k_reduce_net reads half the slabs (c2 20 -> 13 us,
// c5share 41 -> 23 us).  Measured and NOT kept for k_wgrad_net itself (it runs at 0.32-0.37 matrix-pipe busy): register-prefetching the next operand tiles
// under the MFMA loop (63.6 vs 61.6 us at c2) and 128 x 128 output tiles with 8 waves, i.e. half the L2 traffic per flop (62.7 vs 61.5 us at c2, 120 vs 110
// at the c5 share) -- neither exposed load latency nor L2 bandwidth is its bound.  512 rows: k_reduce_net 13.6 -> 10.3 / 24 -> 14 us, k_wgrad_net
// 94 -> 91 us at c2 (825 workgroups: one round of the 1024 slots) but 154 -> 166 us at the c5 share and 311 -> 323 us at c3: not taken.
#define MV_WG_CHUNK 256
struct WgradLayer {
    const float* P1; const float* Q1; const float* P2; const float* Q2;   // [M, No], [M, Ki]; P2 null: single pair
    int ldp1, ldq1, ldp2, ldq2;
    int No, Ki, nbx, nby;
    int M;                         // rows of this layer's operands (deferred step, WgradNetArgs.cnt set: mbase + cnt[0]; M then bounds the slabs)
    int mbase;
    int nchunks;                   // 128-row chunks of M = slabs of this layer
    int ch0, nch;                  // the chunks THIS launch computes (a caller may split a layer's chunks over two launches)
    int blk0;                      // first workgroup of this layer in this launch
    float* slab; float* bslab;     // [nchunks][No * Ki], [nchunks][No]
    float* dW; float* db;          // reduction targets [No * Ki], [No]
    unsigned woff, boff;           // prefix offsets of this layer in k_reduce_net's flattened index space
};
struct WgradNetArgs {
    int n_layers, chunk;
    WgradLayer L[MV_WG_MAXL];
    // optional column sums: colslab[ch][c] = sum over chunk ch's rows of colX[row][c] (extra workgroups of k_wgrad_net from col_blk0 on);
    // k_reduce_net adds them to row 0 of layer col_layer (E.1 end: W_last[0, :] += sum_rows ubar_last)
    const float* colX; int col_ld, col_M, col_n, col_layer, col_nchunks, col_ch0, col_nch, col_blk0;
    float* colslab;
    unsigned wtotal, btotal;
    const long long* cnt; int col_mbase;   // deferred step: device-side row counts (step_internal.h); the chunks beyond the true rows are neither computed nor summed
    int xcd_runs, nblocks;             // xcd_runs: runs of 16 consecutive logical blocks per XCD (the launch grid is padded to a multiple of 128); nblocks: logical blocks
};

__device__ __forceinline__ int mv_wg_rows(const WgradNetArgs& a, const WgradLayer& L) { return a.cnt ? L.mbase + (int)a.cnt[0] : L.M; }
__device__ __forceinline__ int mv_wg_col_rows(const WgradNetArgs& a) { return a.cnt ? a.col_mbase + (int)a.cnt[0] : a.col_M; }

// `scratch`: 256 floats of LDS (the caller's operand tile: a static array of its own here would push k_wgrad_net from 40 960 to 41 984 bytes of LDS, i.e.
// from FOUR to THREE workgroups per CU)
__device__ __forceinline__ void mv_colsum_block(const WgradNetArgs& a, int local, float* scratch) {
    float (*red)[64] = (float (*)[64])scratch;
    const int nbx = (a.col_n + 63) / 64;
    const int ch = a.col_ch0 + local / nbx, bx = local - (local / nbx) * nbx;
    const int c = bx * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int rbeg = ch * a.chunk, rend = min(mv_wg_col_rows(a), rbeg + a.chunk);
    if (rbeg >= rend && a.cnt) return;                            // (deferred step: a chunk beyond the true rows; k_reduce_net does not read it)
    float s = 0.0f;
    if (c < a.col_n)
        for (int row = rbeg + g; row < rend; row += 4) s += a.colX[(size_t)row * a.col_ld + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < a.col_n) a.colslab[(size_t)ch * a.col_n + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(MV_THREADS) void k_wgrad_net(WgradNetArgs a) {
    constexpr int LD = 80;
    __shared__ __attribute__((aligned(16))) float Pt[64 * LD];
    __shared__ __attribute__((aligned(16))) float Qt[64 * LD];
    int bid = blockIdx.x;
    if (a.xcd_runs) {
        // The dispatcher places workgroup b on XCD b % 8.  The 16 output blocks of one (layer, 256-row chunk) share their operand tiles: in launch order
        // every tile is pulled into several L2s.  Here each XCD takes RUNS of 16 consecutive logical blocks: workgroup b = 8 i + x  ->  logical block
        // ((i >> 4) * 8 + x) * 16 + (i & 15).
        const int x = bid & 7, i = bid >> 3;
        bid = (((i >> 4) << 3) + x) * 16 + (i & 15);
        if (bid >= a.nblocks) return;
    }
    if (a.colX && bid >= a.col_blk0) { mv_colsum_block(a, bid - a.col_blk0, Pt); return; }
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    int l = 0;
    while (l + 1 < a.n_layers && bid >= a.L[l + 1].blk0) ++l;
    const WgradLayer& L = a.L[l];
    const int local = bid - L.blk0, nb = L.nbx * L.nby;
    const int chl = local / nb, rem = local - chl * nb, by = rem / L.nbx, bx = rem - by * L.nbx;
    const int ch = L.ch0 + chl;
    const int i0 = bx * 64, o0 = by * 64, No = L.No, Ki = L.Ki;
    const int rbeg = ch * a.chunk, rend = min(mv_wg_rows(a, L), rbeg + a.chunk);
    if (rbeg >= rend && a.cnt) return;                            // (deferred step: a chunk beyond the true rows; k_reduce_net does not read its slab)
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const bool do_bias = bx == 0;
    CH_PH_DECL
    const int npairs = L.P2 ? 2 : 1;
    // Fast path (full 64-column tiles, 16-byte aligned rows: every 256-wide layer): the operand tiles of stage s + 1 are requested into
    // registers BEFORE the MFMA loop of stage s and written to LDS after it -- the probe (tools/chain_probe.py) showed a workgroup spending half
    // of its life waiting for the tiles it had just asked for.  The loads are unconditional (rows clamped, masked when stored): a load inside a
    // branch cannot stay in flight across the loop (the compiler waits for it at the merge) -- why the same idea measured nothing in round 3's
    // first attempt.
    const int nrb = (rend - rbeg + 63) / 64, nst = npairs * nrb;
    const bool fast = ((L.ldp1 & 3) == 0) && ((L.ldq1 & 3) == 0) && ((((size_t)L.P1) & 15) == 0) && ((((size_t)L.Q1) & 15) == 0) && o0 + 64 <= No && i0 + 64 <= Ki &&
                      (!L.P2 || (((L.ldp2 & 3) == 0) && ((L.ldq2 & 3) == 0) && ((((size_t)L.P2) & 15) == 0) && ((((size_t)L.Q2) & 15) == 0))) && nst > 0;
    if (fast) {
        float4 pv[4], qv[4];
        auto issue = [&](int st) {
            const int pair = st >= nrb ? 1 : 0, rb = rbeg + (st - pair * nrb) * 64;
            const float* P = pair ? L.P2 : L.P1;
            const float* Q = pair ? L.Q2 : L.Q1;
            const int ldp = pair ? L.ldp2 : L.ldp1, ldq = pair ? L.ldq2 : L.ldq1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = u * MV_THREADS + tid, rr = idx >> 4, c4 = (idx & 15) * 4;
                const int row = min(rb + rr, rend - 1);
                pv[u] = *(const float4*)(P + (size_t)row * ldp + o0 + c4);
                qv[u] = *(const float4*)(Q + (size_t)row * ldq + i0 + c4);
            }
        };
        auto store = [&](int st) {
            const int pair = st >= nrb ? 1 : 0, rb = rbeg + (st - pair * nrb) * 64;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = u * MV_THREADS + tid, rr = idx >> 4, c4 = (idx & 15) * 4;
                const bool in = rb + rr < rend;
                *(float4*)(Pt + rr * LD + c4) = in ? pv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4*)(Qt + rr * LD + c4) = in ? qv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto compute = [&](int st) {
            if (do_bias && st < nrb && tid < 64) {
                for (int rr = 0; rr < 64; ++rr) bsum += Pt[rr * LD + tid];
            }
            // operands of k-step s + 1 are read from LDS while the four MFMAs of k-step s issue (the plain loop reads, waits ~100 cycles, issues two
            // MFMAs, reads again: each wave kept the matrix pipe busy 40 % of its own time)
            float av[2], bq[2][4];
            const float* pa = Pt + q * LD + 16 * w + r;
            const float* pb = Qt + q * LD + r;
            av[0] = pa[0];
#pragma unroll
            for (int t = 0; t < 4; ++t) bq[0][t] = pb[16 * t];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) {
                    av[(s + 1) & 1] = pa[4 * (s + 1) * LD];
#pragma unroll
                    for (int t = 0; t < 4; ++t) bq[(s + 1) & 1][t] = pb[4 * (s + 1) * LD + 16 * t];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1], bq[s & 1][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        issue(0);
        for (int st = 0; st + 1 < nst; ++st) {
            __syncthreads();
            store(st);
            __syncthreads();
            issue(st + 1);
            compute(st);
        }
        __syncthreads();
        store(nst - 1);
        __syncthreads();
        compute(nst - 1);
    } else
    for (int pair = 0; pair < npairs; ++pair) {
        const float* P = pair ? L.P2 : L.P1;
        const float* Q = pair ? L.Q2 : L.Q1;
        const int ldp = pair ? L.ldp2 : L.ldp1, ldq = pair ? L.ldq2 : L.ldq1;
        for (int rb = rbeg; rb < rend; rb += 64) {
            CH_PH(0)
            __syncthreads();
            CH_PH(1)
            wg_stage(P, ldp, rb, rend, o0, No, Pt, LD, tid);
            wg_stage(Q, ldq, rb, rend, i0, Ki, Qt, LD, tid);
            CH_PH(2)
            __syncthreads();
            CH_PH(3)
            if (do_bias && pair == 0 && tid < 64) {
                for (int rr = 0; rr < 64; ++rr) bsum += Pt[rr * LD + tid];
            }
            CH_PH(4)
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const float av = Pt[(4 * s + q) * LD + 16 * w + r];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Qt[(4 * s + q) * LD + 16 * t + r], acc[t], 0, 0, 0);
            }
        }
    }
    CH_PH(5)
    float* slab = L.slab + (size_t)ch * No * Ki;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = i0 + 16 * t + r;
        if (i < Ki) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = o0 + 16 * w + 4 * q + e;
                if (o < No) slab[(size_t)o * Ki + i] = acc[t][e];
            }
        }
    }
    if (do_bias && tid < 64 && o0 + tid < No) L.bslab[(size_t)ch * No + o0 + tid] = bsum;
    CH_PH(6)
    WG_PH_END(a.L[1].blk0 + 1)                                   // a bx = 1 block (no bias sum) of the first chunk of layer 1 (256 x 256)
}

// dW / db of every layer = sum over its chunks of the slabs, fixed order (deterministic)
__global__ void k_reduce_net(WgradNetArgs a) {
    const unsigned total = a.wtotal + a.btotal;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const bool isb = i >= a.wtotal;
        const unsigned k = isb ? i - a.wtotal : i;
        int l = 0;
        if (isb) { while (l + 1 < a.n_layers && k >= a.L[l + 1].boff) ++l; }
        else { while (l + 1 < a.n_layers && k >= a.L[l + 1].woff) ++l; }
        const WgradLayer& L = a.L[l];
        float v = 0.0f;
        if (isb) {
            const unsigned j = k - L.boff;
            const float* sp = L.bslab + j;
            const int nchb = a.cnt ? (mv_wg_rows(a, L) + a.chunk - 1) / a.chunk : L.nchunks;
            for (int c = 0; c < nchb; ++c) v += sp[(size_t)c * L.No];
            L.db[j] = v;
        } else {
            const unsigned j = k - L.woff;
            const size_t nk = (size_t)L.No * L.Ki;
            const float* sp = L.slab + j;
            const int nch = a.cnt ? (mv_wg_rows(a, L) + a.chunk - 1) / a.chunk : L.nchunks;
            for (int c = 0; c < nch; c += 8) {                    // eight slabs requested at once (clamped), added in slab order
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = sp[(size_t)(c + u < nch ? c + u : nch - 1) * nk];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (c + u < nch) v += t[u];
            }
            if (a.colslab && l == a.col_layer && j < (unsigned)a.col_n) {
                const int ncc = a.cnt ? (mv_wg_col_rows(a) + a.chunk - 1) / a.chunk : a.col_nchunks;
                for (int c = 0; c < ncc; ++c) v += a.colslab[(size_t)c * a.col_n + j];
            }
            L.dW[j] = v;
        }
    }
}

// part[chunk][c] = sum over the chunk's rows of X[row][c]   (summed by k_reduce_net)
__global__ void k_colsum(const float* __restrict__ X, int ld, int M, int n, int chunk, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, ch = blockIdx.y;
    const int rbeg = ch * chunk, rend = min(M, rbeg + chunk);
    float s = 0.0f;
    if (c < n)
        for (int row = rbeg + g; row < rend; row += 4) s += X[(size_t)row * ld + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < n) part[(size_t)ch * n + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused chains: all layers of one backward chain in ONE launch, the running activation tile never leaves LDS.
// 16*MT rows per workgroup, NW waves.  Same arithmetic as the per-layer kernels (k_layer) they replace.
template <class NET>
struct ChainArgsT {
    NET net;                   // packs of W_l   (E.1)
    NET netT;                  // packs of W_l^T (E.2)
    int S, M, row_ld0;         // LDS stride; rows; padded PE row length (ld0)
    const float* dy; int ld_dy;                // E.2: upstream of the outputs [M][Nout]
    const float* Z[MV_MAXL]; const float* U[MV_MAXL];       // forward context (already offset to the first row): Z_l [M][N_l], u_l [M][N_{l-1}]
    const float* w_last_row0;                  // u_L = W_L[0, :]
    const float* ZB2[MV_MAXL];                 // E.1 -> E.2: second-order terms [M][N_l] (null: none)
    float* ZB[MV_MAXL];                        // E.2 out: zbar_l [M][N_l]
    float* H0B;                                // E.2 out: adjoint of the PE input [M][ld0]
    const float* VB0;                          // E.1 in: gbar_0 [M][ld0]
    float* VB[MV_MAXL];                        // E.1 out: vbar_l [M][K_l] (l >= 1)
    float* ZB2o[MV_MAXL];                      // E.1 out
    // k_chain_bwd (whole pass in one launch)
    const float* H0; const float* G0;          // PE values / forward g_0 [M][row_ld0]
    const float* dn_in;                        // [M][3] upstream of the normals (null: E.2 only)
    float* VB0w;                               // vbar_0 out [M][row_ld0]
    float* dx;                                 // [M][3] input adjoint (null: not needed)
    // delta pass (E.2 only, dn_in == NULL): the upstream is one scalar per row on output column 0, and zbar_l is ADDED to the stored one
    const float* dy_col0;                      // [M] (null: the full dy)
    int accum;                                 // 1: ZB[l] += zbar_l (H0B may be null: not written)
    // deferred step (step_internal.h, "device-side counts"): rows = cnt_base + cnt[0] instead of M (M then is the upper bound the grid was sized for)
    const long long* cnt; int cnt_base;
};
typedef ChainArgsT<MvNet> ChainArgs;           // (ChainArgsT<MvNetBf>: the three-term bf16 chains of chain_x3.h)
template <class NET>
__device__ __forceinline__ int mv_chain_rows(const ChainArgsT<NET>& a) { return a.cnt ? a.cnt_base + (int)a.cnt[0] : a.M; }

// E.2 (descending): hb_L = dy W_L;  for l = L-1..0: zb_l = sigma_l . hb_{l+1} + zb2_l (stored), ab_l = zb_l W_l, split at the skip layer.
template <int MT, int NTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_e2(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * ROWS, S = a.S, nl = a.net.n_layers, d0 = 3 + 6 * a.net.multires;
    const unsigned skm = a.net.skip_mask;
    float* act = smem;
    float* pe_adj = smem + ROWS * S;                             // [ROWS][d0]: PE adjoint contributed by the skip layer, added at l = 0
    for (int i = tid; i < ROWS * d0; i += NTH) pe_adj[i] = 0.0f;
    for (int l = nl - 1; l >= 0; --l) {
        const MvLayer& L = a.netT.L[l];                          // contraction over out_l (K), produces in_l columns (N)
        const int K = L.K, Kp = L.KB * 16, N = L.N;
        // ---- prologue: build the A tile (zbar_l, or dy for the last layer) in LDS
        for (int base = 0; base < ROWS * Kp; base += NTH * 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTH + tid;
                const int rr = idx / Kp, k = idx - rr * Kp, row = row0 + rr;
                v[u] = 0.0f;
                if (idx < ROWS * Kp && row < a.M && k < K) {
                    if (l == nl - 1) v[u] = a.dy_col0 ? (k == 0 ? a.dy_col0[row] : 0.0f) : a.dy[(size_t)row * a.ld_dy + k];
                    else {
                        float zb = a.Z[l][(size_t)row * K + k] * act[rr * S + mv_perm(k)];
                        if (a.ZB2[l]) zb += a.ZB2[l][(size_t)row * K + k];
                        a.ZB[l][(size_t)row * K + k] = a.accum ? a.ZB[l][(size_t)row * K + k] + zb : zb;   // the GEMM below continues with this pass's zbar
                        v[u] = zb;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTH + tid;
                const int rr = idx / Kp, k = idx - rr * Kp;
                if (idx < ROWS * Kp) act[rr * S + mv_perm(k)] = v[u];
            }
        }
        __syncthreads();
        // ---- GEMM
        const int NT = L.NT, per = (NT + NW - 1) / NW, ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
        __syncthreads();
        // ---- epilogue: hb for the next (lower) layer stays in LDS; PE adjoint goes to global
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col = (ct0 + t) * 16 + r;
                if (col < N) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                            float v = acc[m][t][i];
                            if (mv_skip_at(skm, l)) {
                                v = dm_div_sqrt2(v);
                                if (col < N - d0) act[rr * S + mv_perm(col)] = v;
                                else pe_adj[rr * d0 + (col - (N - d0))] += v;              // several skip layers: their PE adjoints add up (zeroed above)
                            } else if (l == 0) {
                                if (row < a.M) a.H0B[(size_t)row * a.row_ld0 + col] = pe_adj[rr * d0 + col] + v;
                            } else {
                                act[rr * S + mv_perm(col)] = v;
                            }
                        }
                }
            }
        }
        __syncthreads();
    }
}

// E.1 (ascending): vbar_0 = gbar_0;  for l = 0..L-2: sbar_l = vbar_l W_l^T;  vbar_{l+1} = sigma_l . sbar_l (stored; /sqrt2 + PE tail at the
// skip layer);  zb2_l = u_{l+1} . sbar_l . sigma'_l (stored).
template <int MT, int NTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_e1(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * ROWS, S = a.S, nl = a.net.n_layers, d0 = 3 + 6 * a.net.multires;
    const unsigned skm = a.net.skip_mask;
    float* act = smem;
    float* g0s = smem + ROWS * S;                                // [ROWS][d0]: gbar_0, re-enters at the skip layer
    {
        const int Kp0 = a.net.L[0].KB * 16;
        for (int idx = tid; idx < ROWS * Kp0; idx += NTH) {
            const int rr = idx / Kp0, k = idx - rr * Kp0, row = row0 + rr;
            const float v = (row < a.M && k < d0) ? a.VB0[(size_t)row * a.row_ld0 + k] : 0.0f;
            act[rr * S + mv_perm(k)] = v;
            if (k < d0) g0s[rr * d0 + k] = v;
        }
    }
    for (int l = 0; l < nl - 1; ++l) {
        const MvLayer& L = a.net.L[l];
        const int N = L.N;
        const bool top = (l == nl - 2), to_skip = mv_skip_at(skm, l + 1);
        const int NT = L.NT, per = (NT + NW - 1) / NW, ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        // side inputs of the epilogue (z_l, u_{l+1}): requested BEFORE the GEMM so their latency hides behind it
        float zz[NTW][MT][4], uu[NTW][MT][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = (ct0 + t) * 16 + r;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + m * 16 + 4 * q + i;
                    zz[t][m][i] = 0.f; uu[t][m][i] = 0.f;
                    if (t < ntw && col < N && row < a.M) {
                        zz[t][m][i] = a.Z[l][(size_t)row * N + col];
                        uu[t][m][i] = top ? (to_skip ? dm_div_sqrt2(a.w_last_row0[col]) : a.w_last_row0[col]) : a.U[l + 1][(size_t)row * N + col];
                    }
                }
        }
        mv_barrier_lds();                                        // the tile lives in LDS; the side loads above stay in flight
        if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
        mv_barrier_lds();
        const int ldn = a.net.L[l + 1].K;                        // row length of vbar_{l+1}
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col = (ct0 + t) * 16 + r;
                if (col < N) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                            const float sb = acc[m][t][i];
                            const float sig = zz[t][m][i];                           // the saved sigma
                            float ub = sig * sb;
                            if (to_skip) ub = dm_div_sqrt2(ub);
                            act[rr * S + mv_perm(col)] = ub;
                            if (row < a.M) {
                                a.VB[l + 1][(size_t)row * ldn + col] = ub;
                                a.ZB2o[l][(size_t)row * N + col] = uu[t][m][i] * sb * dm_sigmoid_prime100(sig);
                            }
                        }
                }
            }
        }
        if (l + 1 < nl - 1 || true) {
            const int Kn = a.net.L[l + 1].K, Kpn = a.net.L[l + 1].KB * 16;
            if (to_skip)
                for (int idx = tid; idx < ROWS * d0; idx += NTH) {
                    const int rr = idx / d0, j = idx - rr * d0;
                    act[rr * S + mv_perm(N + j)] = dm_div_sqrt2(g0s[rr * d0 + j]);
                }
            if (Kpn > Kn) {
                const int pad = Kpn - Kn;
                for (int idx = tid; idx < ROWS * pad; idx += NTH) {
                    const int rr = idx / pad, j = idx - rr * pad;
                    act[rr * S + mv_perm(Kn + j)] = 0.0f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One backward pass of the SDF network per row tile in ONE launch: gbar_0 = J_PE nbar (k_pe_normal_bwd), the ascending E.1 chain, the
// descending E.2 chain and the input adjoint xbar = J_PE^T hbar_0 + second-order PE term (k_pe_input_bwd, App. E.3).  Same arithmetic
// as those four launches; z_l bar 2 (E.1 -> E.2) goes through global memory of the same workgroup (L2-hot).  dn_in == NULL: E.2 only.
template <int MT, int NTW, int NW>
__device__ __forceinline__ void mv_chain_bwd_body(const ChainArgs& a, int blk, float* smem) {
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = blk * ROWS, S = a.S, nl = a.net.n_layers, d0 = 3 + 6 * a.net.multires;
    const unsigned skm = a.net.skip_mask;
    const int Mr = mv_chain_rows(a);                             // (deferred step: the true row count comes from device memory, the grid covers its upper bound)
    if (row0 >= Mr) return;
    float* act = smem;
    float* g0s = smem + ROWS * S;                                // [ROWS][d0]: gbar_0 (E.1), then the PE adjoint (E.2)
    float* pe_adj = g0s;
    if (a.dn_in) {
        {   // gbar_0 = J_PE nbar
            const int Kp0 = a.net.L[0].KB * 16;
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
                act[rr * S + mv_perm(k)] = v;
                if (k < d0) g0s[rr * d0 + k] = v;
            }
        }
        for (int l = 0; l < nl - 1; ++l) {
            const MvLayer& L = a.net.L[l];
            const int N = L.N;
            const bool top = (l == nl - 2), to_skip = mv_skip_at(skm, l + 1);
            const int NT = L.NT, per = (NT + NW - 1) / NW, ct0 = w * per;
            int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
            f32x4 acc[MT][NTW];
            mv_zero_acc<MT, NTW>(acc);
            // side inputs of the epilogue (z_l, u_{l+1}): requested BEFORE the GEMM so their latency hides behind it
            float zz[NTW][MT][4], uu[NTW][MT][4];
    #pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int col = (ct0 + t) * 16 + r;
    #pragma unroll
                for (int m = 0; m < MT; ++m)
    #pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = row0 + m * 16 + 4 * q + i;
                        zz[t][m][i] = 0.f; uu[t][m][i] = 0.f;
                        if (t < ntw && col < N && row < Mr) {
                            zz[t][m][i] = a.Z[l][(size_t)row * N + col];
                            uu[t][m][i] = top ? (to_skip ? dm_div_sqrt2(a.w_last_row0[col]) : a.w_last_row0[col]) : a.U[l + 1][(size_t)row * N + col];
                        }
                    }
            }
            mv_barrier_lds();                                        // the tile lives in LDS; the side loads above stay in flight
            if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
            mv_barrier_lds();
            const int ldn = a.net.L[l + 1].K;                        // row length of vbar_{l+1}
    #pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (t < ntw) {
                    const int col = (ct0 + t) * 16 + r;
                    if (col < N) {
    #pragma unroll
                        for (int m = 0; m < MT; ++m)
    #pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                                const float sb = acc[m][t][i];
                                const float sig = zz[t][m][i];                           // the saved sigma
                                float ub = sig * sb;
                                if (to_skip) ub = dm_div_sqrt2(ub);
                                act[rr * S + mv_perm(col)] = ub;
                                if (row < Mr) {
                                    a.VB[l + 1][(size_t)row * ldn + col] = ub;
                                    a.ZB2o[l][(size_t)row * N + col] = uu[t][m][i] * sb * dm_sigmoid_prime100(sig);
                                }
                            }
                    }
                }
            }
            if (l + 1 < nl - 1 || true) {
                const int Kn = a.net.L[l + 1].K, Kpn = a.net.L[l + 1].KB * 16;
                if (to_skip)
                    for (int idx = tid; idx < ROWS * d0; idx += NTH) {
                        const int rr = idx / d0, j = idx - rr * d0, row = row0 + rr;
                        const float tv = dm_div_sqrt2(g0s[rr * d0 + j]);
                        act[rr * S + mv_perm(N + j)] = tv;
                        if (row < Mr) a.VB[l + 1][(size_t)row * ldn + N + j] = tv;     // PE tail of the skip layer's vbar
                    }
                if (Kpn > Kn) {
                    const int pad = Kpn - Kn;
                    for (int idx = tid; idx < ROWS * pad; idx += NTH) {
                        const int rr = idx / pad, j = idx - rr * pad;
                        act[rr * S + mv_perm(Kn + j)] = 0.0f;
                    }
                }
            }
        }
        __syncthreads();                                         // zbar2 of this tile written (global) before E.2 reads it
    }
    for (int i = tid; i < ROWS * d0; i += NTH) pe_adj[i] = 0.0f;
    // sigma_l and zbar2_l of this thread's elements of a layer's prologue, requested one layer ahead (before the GEMM of the layer above), like
    // k_chain_fwd's normal chain: unconditional loads (clamped), the first ZPF passes of a prologue
    constexpr int ZPF = 2;
    float zpre[ZPF][4], z2pre[ZPF][4];
    auto z_prefetch = [&](int lq) {
        const int Kq = a.netT.L[lq].K, Kpq = a.netT.L[lq].KB * 16;
        const float* z2 = a.ZB2[lq] ? a.ZB2[lq] : a.Z[lq];      // (no second-order term in this pass: any valid address, the value is not used)
#pragma unroll
        for (int it = 0; it < ZPF; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = it * NTH * 4 + u * NTH + tid;
                const int rr = idx / Kpq, k = idx - rr * Kpq, row = row0 + rr;
                const bool ok = idx < ROWS * Kpq && row < Mr && k < Kq;
                const size_t off = ok ? (size_t)row * Kq + k : 0;
                zpre[it][u] = a.Z[lq][off];
                z2pre[it][u] = z2[off];
            }
    };
    for (int l = nl - 1; l >= 0; --l) {
        const MvLayer& L = a.netT.L[l];                          // contraction over out_l (K), produces in_l columns (N)
        const int K = L.K, Kp = L.KB * 16, N = L.N;
        // ---- prologue: build the A tile (zbar_l, or dy for the last layer) in LDS
        auto prologue_pass = [&](int base, const float* zp, const float* z2p) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTH + tid;
                const int rr = idx / Kp, k = idx - rr * Kp, row = row0 + rr;
                v[u] = 0.0f;
                if (idx < ROWS * Kp && row < Mr && k < K) {
                    if (l == nl - 1) v[u] = a.dy_col0 ? (k == 0 ? a.dy_col0[row] : 0.0f) : a.dy[(size_t)row * a.ld_dy + k];
                    else {
                        float zb = (zp ? zp[u] : a.Z[l][(size_t)row * K + k]) * act[rr * S + mv_perm(k)];
                        if (a.ZB2[l]) zb += z2p ? z2p[u] : a.ZB2[l][(size_t)row * K + k];
                        a.ZB[l][(size_t)row * K + k] = a.accum ? a.ZB[l][(size_t)row * K + k] + zb : zb;   // the GEMM below continues with this pass's zbar
                        v[u] = zb;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTH + tid;
                const int rr = idx / Kp, k = idx - rr * Kp;
                if (idx < ROWS * Kp) act[rr * S + mv_perm(k)] = v[u];
            }
        };
        if (l == nl - 1) {
            for (int base = 0; base < ROWS * Kp; base += NTH * 4) prologue_pass(base, nullptr, nullptr);
        } else {
#pragma unroll
            for (int it = 0; it < ZPF; ++it)
                if (it * NTH * 4 < ROWS * Kp) prologue_pass(it * NTH * 4, zpre[it], z2pre[it]);
            for (int base = ZPF * NTH * 4; base < ROWS * Kp; base += NTH * 4) prologue_pass(base, nullptr, nullptr);
        }
        __syncthreads();
        z_prefetch(l > 0 ? l - 1 : 0);                           // (l == 0: loaded, unused)
        // ---- GEMM
        const int NT = L.NT, per = (NT + NW - 1) / NW, ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
        __syncthreads();
        // ---- epilogue: hb for the next (lower) layer stays in LDS; PE adjoint goes to global
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col = (ct0 + t) * 16 + r;
                if (col < N) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                            float v = acc[m][t][i];
                            if (mv_skip_at(skm, l)) {
                                v = dm_div_sqrt2(v);
                                if (col < N - d0) act[rr * S + mv_perm(col)] = v;
                                else pe_adj[rr * d0 + (col - (N - d0))] += v;              // several skip layers: their PE adjoints add up (zeroed above)
                            } else if (l == 0) {
                                const float hb0 = pe_adj[rr * d0 + col] + v;
                                pe_adj[rr * d0 + col] = hb0;                       // kept for the input adjoint below
                                if (row < Mr && a.H0B) a.H0B[(size_t)row * a.row_ld0 + col] = hb0;
                            } else {
                                act[rr * S + mv_perm(col)] = v;
                            }
                        }
                }
            }
        }
        __syncthreads();
    }

    if (a.dx) {
        __syncthreads();
        for (int idx = tid; idx < ROWS * 3; idx += NTH) {        // xbar = J_PE^T hbar_0 + sum_k PE''_k g0[k] nbar[c(k)]
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

template <int MT, int NTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_bwd(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    mv_chain_bwd_body<MT, NTW, NW>(a, blockIdx.x, smem);
}

// Two independent passes in one grid (workgroups [0, na) run `a`, the rest run `b`): the full pass over [samples | hit rays] and the
// input-adjoint-only pass over the hit rays do not depend on each other (see functional._IdrStep.backward) and together fill the chip.
static_assert(2 * sizeof(ChainArgs) + 16 <= 4096, "kernel arguments of k_chain_bwd2 exceed 4 KiB");
template <int MT, int NTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_bwd2(ChainArgs a, ChainArgs b, int na) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < na) mv_chain_bwd_body<MT, NTW, NW>(a, blockIdx.x, smem);
    else mv_chain_bwd_body<MT, NTW, NW>(b, blockIdx.x - na, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// Forward value + normal of the SDF network for 16*MT rows per workgroup in ONE launch (replaces 1 + 9 + 8 + 1 launches):
//   value  (ascending, idr.py:77-94):  PE -> [Linear, Softplus(100)] x (L-1) -> Linear; stores H0, A_l, sigma_l = sigmoid(100 z_l), y;
//   normal (descending, idr.py:96-107 = VJP of output 0): u_L = W_L[0,:]; s_l = sigma(100 z_l) . u_{l+1}; u_l = s_l W_l (split and
//           /sqrt2 at the skip layer); g_0 = u_0 (+ PE part of the skip layer); n = J_PE^T g_0; stores Sg_l, U_l, G0, n.
// The running activation / adjoint tile never leaves LDS.  Same arithmetic as the per-layer kernels it replaces.
template <class NET>
struct FwdArgsT {
    NET net, netT;
    int S, M, Mg, ld0;                         // M / Mg: END of the rows this launch evaluates / gives normals to ...
    int row_base;                              // ... which start at row_base (workgroup b owns the rows row_base + 16 MT b ..)
    const float* x;                            // [M][3]
    float* H0;                                 // [M][ld0]
    float* A[MV_MAXL]; float* Z[MV_MAXL];      // A_l [M][K_l] (l >= 1), Z_l [M][N_l] = sigma_l = sigmoid100(z_l) (what every later pass needs of z_l)
    float* U[MV_MAXL]; float* Sg[MV_MAXL];     // u_l [Mg][N_{l-1}] (1 <= l <= L-2), s_l [Mg][N_l]
    float* G0;                                 // [Mg][ld0]
    float* y; int ldy;                         // [M][Nout]
    const float* w_last_row0;                  // W_L[0, :]
    float* nrm;                                // [Mg][3]
    FwdGather g;                               // g.pts != null: x is gathered (x is ignored)
};
typedef FwdArgsT<MvNet> FwdArgs;

template <int MT, int NTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_fwd(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = a.row_base + blockIdx.x * ROWS, S = a.S, nl = a.net.n_layers, d0 = 3 + 6 * a.net.multires;
    const unsigned skm = a.net.skip_mask;
    float* act = smem;
    float* pe = act + ROWS * S;                                  // [ROWS][d0] natural order
    float* padj = pe + ((ROWS * d0 + 3) & ~3);                   // [ROWS][d0] PE adjoint of the skip layer, then g_0
    float* pts = padj + ((ROWS * d0 + 3) & ~3);                  // [ROWS][3]
    CH_PH_DECL
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
    mv_pe_rows<NTH>(pts, pe, act, S, ROWS, a.net.multires, tid);
    __syncthreads();
    for (int idx = tid; idx < ROWS * a.ld0; idx += NTH) {
        const int rr = idx / a.ld0, k = idx - rr * a.ld0, row = row0 + rr;
        if (row < a.M) a.H0[(size_t)row * a.ld0 + k] = k < d0 ? pe[rr * d0 + k] : 0.0f;
    }
    CH_PH(0)
    // ---- value chain
    for (int l = 0; l < nl - 1; ++l) {
        const MvLayer& L = a.net.L[l];
        const int N = L.N, NT = L.NT, per = (NT + NW - 1) / NW, ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        const bool to_skip = mv_skip_at(skm, l + 1);
        const int Kn = a.net.L[l + 1].K, Kpn = a.net.L[l + 1].KB * 16;
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        float bv_[NTW];                                          // biases of this wave's columns: requested now, consumed after the GEMM (clamped: no branch)
#pragma unroll
        for (int t = 0; t < NTW; ++t) { const int col = (ct0 + t) * 16 + r; bv_[t] = L.bias[col < N ? col : N - 1]; }
        __syncthreads();
        CH_PH(1)
        if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
        CH_PH(2)
        __syncthreads();
        CH_PH(3)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col = (ct0 + t) * 16 + r;
                if (col < N) {
                    const float bv = bv_[t];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                            const float z = acc[m][t][i] + bv;
                            float h, sg;
                            dm_softplus_sigmoid100(z, &h, &sg);
                            if (to_skip) h = dm_div_sqrt2(h);
                            act[rr * S + mv_perm(col)] = h;
                            if (row < a.M) {
                                a.Z[l][(size_t)row * N + col] = sg;
                                a.A[l + 1][(size_t)row * Kn + col] = h;
                            }
                        }
                }
            }
        }
        if (to_skip)
            for (int idx = tid; idx < ROWS * d0; idx += NTH) {
                const int rr = idx / d0, j = idx - rr * d0, row = row0 + rr;
                const float v = dm_div_sqrt2(pe[rr * d0 + j]);
                act[rr * S + mv_perm(N + j)] = v;
                if (row < a.M) a.A[l + 1][(size_t)row * Kn + N + j] = v;
            }
        if (Kpn > Kn) {
            const int pad = Kpn - Kn;
            for (int idx = tid; idx < ROWS * pad; idx += NTH) {
                const int rr = idx / pad, j = idx - rr * pad;
                act[rr * S + mv_perm(Kn + j)] = 0.0f;
            }
        }
        CH_PH(4)
    }
    {   // last layer: every output column, groups of NTW column tiles per wave
        const MvLayer& L = a.net.L[nl - 1];
        const int N = L.N, NT = L.NT, per = (NT + NW - 1) / NW;
        __syncthreads();
        for (int g0 = 0; g0 < per; g0 += NTW) {
            const int ct0 = w * per + g0;
            int ntw = min(per - g0, NT - ct0);
            ntw = ntw < 0 ? 0 : (ntw > NTW ? NTW : ntw);
            f32x4 acc[MT][NTW];
            mv_zero_acc<MT, NTW>(acc);
            float bv_[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) { const int col = (ct0 + t) * 16 + r; bv_[t] = L.bias[(col >= 0 && col < N) ? col : N - 1]; }
            if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (t < ntw) {
                    const int col = (ct0 + t) * 16 + r;
                    if (col < N) {
                        const float bv = bv_[t];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int row = row0 + m * 16 + 4 * q + i;
                                if (row < a.M) a.y[(size_t)row * a.ldy + col] = acc[m][t][i] + bv;
                            }
                    }
                }
            }
        }
    }
    CH_PH(5)
    if (row0 >= a.Mg) return;                                    // workgroup-uniform: no normals for these rows
    // ---- normal chain (rows >= Mg inside the tile carry zeros)
    // (a skip connection into the LAST Linear, idr.py:46-49,86: u_L = W_L[0, :] splits like any skip layer's adjoint -- its PE part starts the PE adjoint)
    const bool top_skip = mv_skip_at(skm, nl - 1);
    for (int i = tid; i < ROWS * d0; i += NTH) padj[i] = top_skip ? dm_div_sqrt2(a.w_last_row0[a.net.L[nl - 1].K - d0 + (i % d0)]) : 0.0f;
    // sigma_l of this thread's elements of a layer's prologue, requested one layer ahead (before the GEMM of the layer above): the prologue is a
    // load, a multiply and two stores per element, and with the load issued on the spot it was 2 us of exposed L2 latency per layer (probe: 16-18
    // of the kernel's 124 us at c2).  Covers the first ZPF passes of a prologue (all of it up to 32 rows x 256 columns).  No branch around the
    // loads and no branch that rewrites zpre (the compiler would copy registers with loads in flight, i.e. wait for them): clamped addresses.
    constexpr int ZPF = 2;
    float zpre[ZPF][4];
    auto z_prefetch = [&](int lq) {
        const int Kq = a.netT.L[lq].K, Kpq = a.netT.L[lq].KB * 16;
#pragma unroll
        for (int it = 0; it < ZPF; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = it * NTH * 4 + u * NTH + tid;
                const int rr = idx / Kpq, k = idx - rr * Kpq, row = row0 + rr;
                const bool ok = idx < ROWS * Kpq && row < a.Mg && k < Kq;
                zpre[it][u] = a.Z[lq][ok ? (size_t)row * Kq + k : 0];
            }
    };
    z_prefetch(nl - 2);
    for (int l = nl - 2; l >= 0; --l) {
        const MvLayer& L = a.netT.L[l];                          // contraction over out_l (K), produces in_l columns (N)
        const int K = L.K, Kp = L.KB * 16, N = L.N;
        const bool top = (l == nl - 2);
        __syncthreads();                                         // Z_l of this tile written (value chain) / previous epilogue done
        CH_PH(6)
        auto prologue_pass = [&](int base, const float* zp) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTH + tid;
                const int rr = idx / Kp, k = idx - rr * Kp, row = row0 + rr;
                v[u] = 0.0f;
                if (idx < ROWS * Kp && row < a.Mg && k < K) {
                    const float uu = top ? (top_skip ? dm_div_sqrt2(a.w_last_row0[k]) : a.w_last_row0[k]) : act[rr * S + mv_perm(k)];
                    v[u] = (zp ? zp[u] : a.Z[l][(size_t)row * K + k]) * uu;
                    a.Sg[l][(size_t)row * K + k] = v[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTH + tid;
                const int rr = idx / Kp, k = idx - rr * Kp;
                if (idx < ROWS * Kp) act[rr * S + mv_perm(k)] = v[u];
            }
        };
#pragma unroll
        for (int it = 0; it < ZPF; ++it)
            if (it * NTH * 4 < ROWS * Kp) prologue_pass(it * NTH * 4, zpre[it]);
        for (int base = ZPF * NTH * 4; base < ROWS * Kp; base += NTH * 4) prologue_pass(base, nullptr);
        CH_PH(7)
        __syncthreads();
        CH_PH(8)
        z_prefetch(l > 0 ? l - 1 : 0);                           // (l == 0: loaded, unused)
        const int NT = L.NT, per = (NT + NW - 1) / NW, ct0 = w * per;
        int ntw = NT - ct0; ntw = ntw < 0 ? 0 : (ntw > per ? per : ntw);
        f32x4 acc[MT][NTW];
        mv_zero_acc<MT, NTW>(acc);
        if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
        CH_PH(9)
        __syncthreads();
        CH_PH(10)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < ntw) {
                const int col = (ct0 + t) * 16 + r;
                if (col < N) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                            float v = acc[m][t][i];
                            if (mv_skip_at(skm, l)) {
                                v = dm_div_sqrt2(v);
                                if (col < N - d0) {
                                    act[rr * S + mv_perm(col)] = v;
                                    if (row < a.Mg) a.U[l][(size_t)row * (N - d0) + col] = v;
                                } else padj[rr * d0 + (col - (N - d0))] += v;
                            } else if (l == 0) {
                                const float g = padj[rr * d0 + col] + v;
                                padj[rr * d0 + col] = g;
                                if (row < a.Mg) a.G0[(size_t)row * a.ld0 + col] = g;
                            } else {
                                act[rr * S + mv_perm(col)] = v;
                                if (row < a.Mg) a.U[l][(size_t)row * N + col] = v;
                            }
                        }
                }
            }
        }
        CH_PH(11)
    }
    __syncthreads();
    for (int idx = tid; idx < ROWS * 3; idx += NTH) {            // n = J_PE^T g_0
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
// Rendering network (idr.py:145-167, mode 'idr') as one forward and one backward launch per 16*MT rows.
//   forward : a_0 = cat[points, view, PE(view), normals, feat]; a_{l+1} = relu(a_l W_l^T + b_l); rgb = tanh(a_L W_L^T + b_L); stores a_l, rgb
//   backward: zb_L = drgb (1 - rgb^2); ab_l = zb_l W_l; zb_{l-1} = ab_l [a_l > 0]; stores zb_l (for the weight gradients) and din = ab_0
struct RenderChainArgs {
    MvNet net, netT;
    int S, N, mv, K0;
    const float* points; const float* view; const float* normals; const float* feat; int ldfeat;   // forward inputs
    float* A[MV_MAXL]; float* rgb_ctx; float* rgb;                                                   // forward outputs
    const float* drgb; const float* Ac[MV_MAXL]; const float* rgbc; float* ZB[MV_MAXL]; float* din;  // backward
    const long long* drgb_rows;                // backward: row r reads drgb[drgb_rows[r]] (null: drgb[r]) -- the step's upstream arrives in ray order, the net ran on sorted rows
    const long long* cnt;                      // backward, deferred step: the row count is cnt[0] (N then is the upper bound the grid was sized for)
};

// `mv` packs the input layout of RenderingNetwork.forward (idr.py:145-154): low 8 bits = multires_view; bit 8 set = mode 'no_view_dir'
// (cat[points, normals, feat]); bit 9 set = mode 'no_normal' (cat[points, PE(view), feat]); neither = mode 'idr'.
__host__ __device__ static inline int mv_render_dv(int mv) { return (mv & 0x100) ? 0 : 3 + 6 * (mv & 0xff); }
__host__ __device__ static inline int mv_render_dn(int mv) { return (mv & 0x200) ? 0 : 3; }
__device__ __forceinline__ float mv_render_input_raw(const float* points, const float* view, const float* normals, const float* feat, int ldfeat,
                                                     int mv, int row, int k) {
    const int dv = mv_render_dv(mv), dn = mv_render_dn(mv);
    if (k < 3) return points[(size_t)row * 3 + k];
    if (k < 3 + dv) {
        const int j = k - 3;
        if (j < 3) return view[(size_t)row * 3 + j];
        const int jj = j - 3, m = jj / 6, rem = jj - 6 * m, c = rem % 3;
        float sn, co;
        dm_sincos(view[(size_t)row * 3 + c] * (float)(1 << m), &sn, &co);
        return rem < 3 ? sn : co;
    }
    if (k < 3 + dv + dn) return normals[(size_t)row * 3 + (k - 3 - dv)];
    return feat[(size_t)row * ldfeat + (k - 3 - dv - dn)];
}
__device__ __forceinline__ float mv_render_input(const RenderChainArgs& a, int row, int k) {
    return mv_render_input_raw(a.points, a.view, a.normals, a.feat, a.ldfeat, a.mv, row, k);
}

template <int MT, int NTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_render_chain_fwd(RenderChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * ROWS, S = a.S, nl = a.net.n_layers;
    float* act = smem;
    {
        const int K0 = a.K0, Kp0 = a.net.L[0].KB * 16;
        for (int idx = tid; idx < ROWS * Kp0; idx += NTH) {
            const int rr = idx / Kp0, k = idx - rr * Kp0, row = row0 + rr;
            float v = 0.0f;
            if (row < a.N && k < K0) { v = mv_render_input(a, row, k); a.A[0][(size_t)row * K0 + k] = v; }
            act[rr * S + mv_perm(k)] = v;
        }
    }
    for (int l = 0; l < nl; ++l) {
        const MvLayer& L = a.net.L[l];
        const bool last = (l == nl - 1);
        const int N = L.N, NT = L.NT, per = (NT + NW - 1) / NW;
        __syncthreads();
        for (int g0 = 0; g0 < per; g0 += NTW) {
            const int ct0 = w * per + g0;
            int ntw = min(per - g0, NT - ct0);
            ntw = ntw < 0 ? 0 : (ntw > NTW ? NTW : ntw);
            f32x4 acc[MT][NTW];
            mv_zero_acc<MT, NTW>(acc);
            float bv_[NTW];                                      // requested before the GEMM, consumed after it
#pragma unroll
            for (int t = 0; t < NTW; ++t) { const int col = (ct0 + t) * 16 + r; bv_[t] = L.bias[(col >= 0 && col < N) ? col : N - 1]; }
            if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
            if (per > NTW || !last) __syncthreads();             // every wave done reading act before the in-place update
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (t < ntw) {
                    const int col = (ct0 + t) * 16 + r;
                    if (col < N) {
                        const float bv = bv_[t];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                                const float z = acc[m][t][i] + bv;
                                if (last) {
                                    const float y = tanhf(z);
                                    if (row < a.N) { a.rgb[(size_t)row * N + col] = y; a.rgb_ctx[(size_t)row * N + col] = y; }
                                } else {
                                    const float h = fmaxf(z, 0.0f);
                                    if (row < a.N) a.A[l + 1][(size_t)row * N + col] = h;
                                    act[rr * S + mv_perm(col)] = h;
                                }
                            }
                    }
                }
            }
        }
        if (!last) {
            const int Kn = a.net.L[l + 1].K, Kpn = a.net.L[l + 1].KB * 16;
            if (Kpn > Kn) {
                const int pad = Kpn - Kn;
                for (int idx = tid; idx < ROWS * pad; idx += NTH) {
                    const int rr = idx / pad, j = idx - rr * pad;
                    act[rr * S + mv_perm(Kn + j)] = 0.0f;
                }
            }
        }
    }
}

template <int MT, int NTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_render_chain_bwd(RenderChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, NTH = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * ROWS, S = a.S, nl = a.net.n_layers;
    const int Nr = a.cnt ? (int)a.cnt[0] : a.N;
    if (row0 >= Nr) return;
    float* act = smem;
    {   // zb_L = drgb (1 - rgb^2)
        const int K = a.net.L[nl - 1].N, Kp = a.netT.L[nl - 1].KB * 16;
        for (int idx = tid; idx < ROWS * Kp; idx += NTH) {
            const int rr = idx / Kp, k = idx - rr * Kp, row = row0 + rr;
            float v = 0.0f;
            if (row < Nr && k < K) {
                const float y = a.rgbc[(size_t)row * K + k];
                v = a.drgb[(size_t)(a.drgb_rows ? a.drgb_rows[row] : row) * K + k] * (1.0f - y * y);
                a.ZB[nl - 1][(size_t)row * K + k] = v;
            }
            act[rr * S + mv_perm(k)] = v;
        }
    }
    for (int l = nl - 1; l >= 0; --l) {
        const MvLayer& L = a.netT.L[l];                          // contraction over out_l (K), produces in_l columns (N)
        const int N = L.N, NT = L.NT, per = (NT + NW - 1) / NW;
        __syncthreads();
        // the layer's outputs overwrite the tile the GEMM reads: with several column-tile groups per wave they are staged in registers
        // group by group only when one group suffices; otherwise (first layer, N = K0 > 256) the outputs go to global memory only.
        for (int g0 = 0; g0 < per; g0 += NTW) {
            const int ct0 = w * per + g0;
            int ntw = min(per - g0, NT - ct0);
            ntw = ntw < 0 ? 0 : (ntw > NTW ? NTW : ntw);
            f32x4 acc[MT][NTW];
            mv_zero_acc<MT, NTW>(acc);
            // the ReLU masks of this wave's outputs (post-activation values): requested before the GEMM, consumed after it (clamped: no branch)
            float am[NTW][MT][4];
            {
                const float* Acl = l > 0 ? a.Ac[l] : a.rgbc;     // (l == 0: any valid address, the values are not used)
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const int col = (ct0 + t) * 16 + r;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = row0 + m * 16 + 4 * q + i;
                            const bool ok = l > 0 && t < ntw && col >= 0 && col < N && row < Nr;
                            am[t][m][i] = Acl[ok ? (size_t)row * N + col : 0];
                        }
                }
            }
            if (ntw > 0) mv_gemm_dispatch<MT, NTW>(L, act, S, ct0, ntw, acc, lane);
            if (l > 0) __syncthreads();                          // (l > 0 has a single group: per <= NTW, checked by the launcher)
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (t < ntw) {
                    const int col = (ct0 + t) * 16 + r;
                    if (col < N) {
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int rr = m * 16 + 4 * q + i, row = row0 + rr;
                                const float ab = acc[m][t][i];
                                if (l > 0) {
                                    float zb = 0.0f;
                                    if (row < Nr) {
                                        zb = am[t][m][i] > 0.0f ? ab : 0.0f;                          // relu mask: stored post-activation > 0
                                        a.ZB[l - 1][(size_t)row * N + col] = zb;
                                    }
                                    act[rr * S + mv_perm(col)] = zb;
                                } else if (row < Nr) a.din[(size_t)row * N + col] = ab;
                            }
                    }
                }
            }
        }
        if (l > 0) {
            const int Kn = a.netT.L[l - 1].K, Kpn = a.netT.L[l - 1].KB * 16;
            if (Kpn > Kn) {
                const int pad = Kpn - Kn;
                for (int idx = tid; idx < ROWS * pad; idx += NTH) {
                    const int rr = idx / pad, j = idx - rr * pad;
                    act[rr * S + mv_perm(Kn + j)] = 0.0f;
                }
            }
        }
    }
}
