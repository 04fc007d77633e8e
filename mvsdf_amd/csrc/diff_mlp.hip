// diff_mlp.hip -- differentiable passes of the SDF network (value + normal, first/second-order backward) and of the
// rendering network, orchestrated on one stream from C++ (no host sync, no allocation).  Math: SURVEY.md App. E.
// Replaces ImplicitNetwork.forward/.gradient (idr.py:77-107), RenderingNetwork.forward (idr.py:145-167) and what
// torch.autograd does behind loss.backward() for both (idr_train.py:287).
#include <stdlib.h>
#include "layer_kernels.h"
#include "chain_x3.h"
#include "capi_util.h"
#include "step_internal.h"

// ------------------------------------------------------------------------------------------------ small kernels
// H0[row][0..d0) = PE(x[row]); H0[row][d0..ld) = 0       (embedder.py:10-36)
__global__ void k_pe_global(const float* __restrict__ x, int M, int multires, float* __restrict__ H0, int ld) {
    const int T = 3 * multires + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * T) return;
    const int row = idx / T, j = idx - row * T, d0 = 3 + 6 * multires;
    const float* xr = x + (size_t)row * 3;
    float* h = H0 + (size_t)row * ld;
    if (j < 3 * multires) {
        const int m = j / 3, c = j - 3 * m;
        float s, co;
        dm_sincos(xr[c] * (float)(1 << m), &s, &co);
        h[3 + 6 * m + c] = s;
        h[6 + 6 * m + c] = co;
    } else {
        for (int c = 0; c < 3; ++c) h[c] = xr[c];
        for (int c = d0; c < ld; ++c) h[c] = 0.0f;
    }
}

// n = J0^T g0   (App. E forward normal, last step)
__global__ void k_pe_normal(const float* __restrict__ H0, int ldh, const float* __restrict__ G0, int ldg, int Mg, int multires,
                            float* __restrict__ n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mg * 3) return;
    const int row = idx / 3, c = idx - 3 * row;
    const float* h = H0 + (size_t)row * ldh;
    const float* g = G0 + (size_t)row * ldg;
    float v = g[c];
    for (int m = 0; m < multires; ++m) {
        const float f = (float)(1 << m);
        v += f * (h[6 + 6 * m + c] * g[3 + 6 * m + c] - h[3 + 6 * m + c] * g[6 + 6 * m + c]);
    }
    n[idx] = v;
}

// gbar0 = J0 nbar  -> VB0[row][0..d0) (pad 0), and the PE tail of the skip layer's vbar: VBs[row][tail0 + j] = gbar0[j]/sqrt(2)
__global__ void k_pe_normal_bwd(const float* __restrict__ H0, int ldh, const float* __restrict__ dn, int Mb, int multires,
                                float* __restrict__ VB0, int ld0, float* __restrict__ VBs, int lds_, int tail0) {
    const int T = 3 * multires + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mb * T) return;
    const int row = idx / T, j = idx - row * T, d0 = 3 + 6 * multires;
    const float* h = H0 + (size_t)row * ldh;
    const float* nb = dn + (size_t)row * 3;
    float* o = VB0 + (size_t)row * ld0;
    float* t = VBs ? VBs + (size_t)row * lds_ + tail0 : nullptr;
    if (j < 3 * multires) {
        const int m = j / 3, c = j - 3 * m;
        const float f = (float)(1 << m);
        const float a = f * h[6 + 6 * m + c] * nb[c], b = -f * h[3 + 6 * m + c] * nb[c];
        o[3 + 6 * m + c] = a;
        o[6 + 6 * m + c] = b;
        if (t) { t[3 + 6 * m + c] = dm_div_sqrt2(a); t[6 + 6 * m + c] = dm_div_sqrt2(b); }
    } else {
        for (int c = 0; c < 3; ++c) { o[c] = nb[c]; if (t) t[c] = dm_div_sqrt2(nb[c]); }
        for (int c = d0; c < ld0; ++c) o[c] = 0.0f;
    }
}

// xbar = J0^T hbar0 + sum_k PE''_k g0[k] nbar[c(k)]        (App. E.3)
__global__ void k_pe_input_bwd(const float* __restrict__ H0, int ldh, const float* __restrict__ H0B, int ldb, const float* __restrict__ G0,
                               int ldg, const float* __restrict__ dn, int Mb, int multires, float* __restrict__ dx) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mb * 3) return;
    const int row = idx / 3, c = idx - 3 * row;
    const float* h = H0 + (size_t)row * ldh;
    const float* hb = H0B + (size_t)row * ldb;
    float v = hb[c];
    float second = 0.0f;
    for (int m = 0; m < multires; ++m) {
        const float f = (float)(1 << m);
        const float s = h[3 + 6 * m + c], co = h[6 + 6 * m + c];
        v += f * (co * hb[3 + 6 * m + c] - s * hb[6 + 6 * m + c]);
        if (dn) {
            const float* g = G0 + (size_t)row * ldg;
            second -= f * f * (s * g[3 + 6 * m + c] + co * g[6 + 6 * m + c]);
        }
    }
    if (dn) v += second * dn[(size_t)row * 3 + c];
    dx[idx] = v;
}

// E[row][j] = W_L[0, tail0 + j] / sqrt(2): the PE adjoint a skip connection into the LAST Linear starts the normal chain with (per-layer route)
__global__ void k_pe_adj_top(const float* __restrict__ w_last_row0, int tail0, int d0, int Mg, float* __restrict__ E, int ld) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mg * d0) return;
    const int row = idx / d0, j = idx - row * d0;
    E[(size_t)row * ld + j] = dm_div_sqrt2(w_last_row0[tail0 + j]);
}

__global__ void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// ------------------------------------------------------------------------------------------------ launch helpers
template <int PRO, int EPI, int MT>
static hipError_t launch_layer_mt(LayerArgs& a, hipStream_t s) {
    const size_t lds = (size_t)16 * MT * a.S * sizeof(float);
    static size_t lds_set = 0;                                 // per instantiation: raise the dynamic-LDS cap once per size
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_layer<PRO, EPI, MT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL((k_layer<PRO, EPI, MT, 4>), dim3((a.M + 16 * MT - 1) / (16 * MT)), dim3(MV_THREADS), lds, s, a);
    return hipGetLastError();
}

template <int PRO, int EPI>
static hipError_t launch_layer(LayerArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    // rows per workgroup: 16 while that still leaves the chip under-subscribed (one workgroup per CU), else 32 (weights
    // reused by two row tiles).  MVSDF_LAYER_MT overrides (dev).
    static int mt_env = -1;
    if (mt_env < 0) { const char* e = mv_dev_env("MVSDF_LAYER_MT"); mt_env = e ? atoi(e) : 0; }
    const bool small = mt_env ? (mt_env == 1) : (a.M <= 16 * 512);
    return small ? launch_layer_mt<PRO, EPI, 1>(a, s) : launch_layer_mt<PRO, EPI, 2>(a, s);
}

static LayerArgs base_args(const MvLayer& L, int S, int M) {
    LayerArgs a;
    memset(&a, 0, sizeof(a));
    a.L = L; a.S = S; a.M = M;
    return a;
}

static bool mv_chain_w8() {
    static int v = -1;
    if (v < 0) { const char* e = mv_dev_env("MVSDF_CHAIN_W8"); v = e ? atoi(e) : 0; }
    return v != 0;
}

static int stride_for(const MvNet& a, const MvNet& b) { return a.S > b.S ? a.S : b.S; }

// layout of the forward context / backward workspace (floats)
struct SdfLayout {
    int nl, d0, ld0;                 // layers, PE width, padded PE row
    size_t H0, A[MV_MAXL], Z[MV_MAXL], U[MV_MAXL], Sg[MV_MAXL], E, G0, total;
};
static SdfLayout sdf_ctx_layout(const MvNet& net, int M, int Mg) {
    SdfLayout o;
    memset(&o, 0, sizeof(o));
    o.nl = net.n_layers; o.d0 = 3 + 6 * net.multires; o.ld0 = (o.d0 + 3) & ~3;
    size_t p = 0;
    o.H0 = p; p += (size_t)M * o.ld0;
    for (int l = 1; l < o.nl; ++l) { o.A[l] = p; p += (size_t)M * net.L[l].K; }
    for (int l = 0; l < o.nl - 1; ++l) { o.Z[l] = p; p += (size_t)M * net.L[l].N; }
    for (int l = 1; l < o.nl - 1; ++l) { o.U[l] = p; p += (size_t)Mg * net.L[l - 1].N; }   // u_l, width out_{l-1}
    for (int l = 0; l < o.nl - 1; ++l) { o.Sg[l] = p; p += (size_t)Mg * net.L[l].N; }      // s_l = sigma_l . u_{l+1}
    o.E = p; p += (size_t)Mg * o.ld0;
    o.G0 = p; p += (size_t)Mg * o.ld0;
    o.total = p;
    return o;
}
struct SdfBwdLayout {
    size_t VB[MV_MAXL], ZB2[MV_MAXL], ZB[MV_MAXL], HB[2], H0B, slabA, slabB, bslab, total;
    int nchunks, chunk;
    size_t maxnk;
};
static size_t wgrad_net_slab_floats(const MvNet& net, int nchunks) {
    size_t t = 0;
    for (int l = 0; l < net.n_layers; ++l) t += (size_t)nchunks * net.L[l].N * net.L[l].K;
    return t;
}
static size_t wgrad_net_bslab_floats(const MvNet& net, int nchunks) {
    size_t t = 0;
    for (int l = 0; l < net.n_layers; ++l) t += (size_t)nchunks * net.L[l].N;
    return t;
}
static SdfBwdLayout sdf_bwd_layout(const MvNet& net, int Mb) {
    SdfBwdLayout o;
    memset(&o, 0, sizeof(o));
    const int nl = net.n_layers, ld0 = ((3 + 6 * net.multires) + 3) & ~3;
    size_t p = 0;
    int maxw = 0;
    for (int l = 0; l < nl; ++l) { maxw = net.L[l].K > maxw ? net.L[l].K : maxw; maxw = net.L[l].N > maxw ? net.L[l].N : maxw; }
    for (int l = 0; l < nl; ++l) { o.VB[l] = p; p += (size_t)Mb * (l == 0 ? ld0 : net.L[l].K); }
    for (int l = 0; l < nl - 1; ++l) { o.ZB2[l] = p; p += (size_t)Mb * net.L[l].N; }
    for (int l = 0; l < nl - 1; ++l) { o.ZB[l] = p; p += (size_t)Mb * net.L[l].N; }
    o.HB[0] = p; p += (size_t)Mb * maxw;
    o.HB[1] = p; p += (size_t)Mb * maxw;
    o.H0B = p; p += (size_t)Mb * ld0;
    o.chunk = MV_WG_CHUNK;
    o.nchunks = (Mb + o.chunk - 1) / o.chunk;
    if (o.nchunks < 1) o.nchunks = 1;
    o.maxnk = 0;
    for (int l = 0; l < nl; ++l) { const size_t nk = (size_t)net.L[l].N * net.L[l].K; o.maxnk = nk > o.maxnk ? nk : o.maxnk; }
    o.slabA = p; p += wgrad_net_slab_floats(net, o.nchunks);     // one slab per (layer, chunk)
    o.slabB = p; p += (size_t)o.nchunks * maxw;                   // column sums of ubar_last (E.1 end)
    o.bslab = p; p += wgrad_net_bslab_floats(net, o.nchunks);
    o.total = p;
    return o;
}

// ---- weight / bias gradients: k_wgrad_net (+ folded column sums) and k_reduce_net over a list of layers ----
// wgrad_net_layers: per-layer bookkeeping of one network's layers [l0, l0 + n) inside `a` (No / Ki / operands filled by the caller): M rows in
// 128-row chunks, slabs carved from `slab` / `bslab`, reduction targets carved from dW_cat / db_cat.
static void wgrad_net_layers(WgradNetArgs& a, int l0, int n, int M, int nchunks, float* slab, float* bslab, float* dW_cat, float* db_cat) {
    size_t so = 0, bo = 0, wo = 0, b2 = 0;
    for (int l = l0; l < l0 + n; ++l) {
        WgradLayer& L = a.L[l];
        L.M = M; L.nchunks = nchunks; L.ch0 = 0; L.nch = nchunks;
        L.nbx = (L.Ki + 63) / 64; L.nby = (L.No + 63) / 64;
        L.slab = slab + so; so += (size_t)nchunks * L.No * L.Ki;
        L.bslab = bslab + bo; bo += (size_t)nchunks * L.No;
        L.dW = dW_cat + wo; wo += (size_t)L.No * L.Ki;
        L.db = db_cat + b2; b2 += L.No;
    }
}
// one k_wgrad_net launch over the chunk ranges [ch0, ch0 + nch) of the layers (+ the column-sum workgroups when a.colX is set)
static hipError_t wgrad_launch(WgradNetArgs& a, hipStream_t s) {
    int blk = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        WgradLayer& L = a.L[l];
        L.blk0 = blk; blk += L.nbx * L.nby * L.nch;
    }
    if (a.colX) { a.col_blk0 = blk; blk += ((a.col_n + 63) / 64) * a.col_nch; }
    // XCD-aware block order (layer_kernels.h; same blocks, same arithmetic, another placement): on by default, MVSDF_WG_XCD=0 turns it off.  Round 4, rocprofv3:
    // 92.9 -> 85.0 us at c2, 154.6 -> 153.9 at the c5 share, 310.7 -> 304.0 at c3 (round 2's two orders -- a contiguous range per XCD, and this one at c2 on the
    // older kernel -- measured nothing)
    static int xcd_env = -2;
    if (xcd_env == -2) { const char* e = mv_dev_env("MVSDF_WG_XCD"); xcd_env = (e && *e) ? atoi(e) : -1; }
    a.nblocks = blk;
    a.xcd_runs = xcd_env >= 0 ? (xcd_env != 0) : 1;
    const int grid = a.xcd_runs ? ((blk + 127) / 128) * 128 : blk;
    if (blk > 0) hipLaunchKernelGGL(k_wgrad_net, dim3(grid), dim3(MV_THREADS), 0, s, a);
    return hipGetLastError();
}
static hipError_t wgrad_reduce(WgradNetArgs& a, hipStream_t s) {
    size_t wo = 0, b2 = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        WgradLayer& L = a.L[l];
        L.woff = (unsigned)wo; wo += (size_t)L.No * L.Ki;
        L.boff = (unsigned)b2; b2 += L.No;
    }
    if (wo + b2 >= 0xffffffffull) return hipErrorInvalidValue;
    a.wtotal = (unsigned)wo; a.btotal = (unsigned)b2;
    const unsigned total = a.wtotal + a.btotal;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(k_reduce_net, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}
static hipError_t launch_wgrad_net(WgradNetArgs& a, hipStream_t s) {
    hipError_t e = wgrad_launch(a, s);
    return e != hipSuccess ? e : wgrad_reduce(a, s);
}
static void wgrad_colsum(WgradNetArgs& a, const float* X, int ld, int M, int n, int layer, int nchunks, float* colslab) {
    a.colX = X; a.col_ld = ld; a.col_M = M; a.col_n = n; a.col_layer = layer; a.col_nchunks = nchunks; a.col_ch0 = 0; a.col_nch = nchunks; a.colslab = colslab;
}

// the skip layer of a network with at most one (-1: none); -2: several (only the fused chain kernels handle those)
static int mv_single_skip(const MvNet& net) {
    const unsigned m = net.skip_mask;
    if (!m) return -1;
    if (m & (m - 1)) return -2;
    return __builtin_ctz(m);
}

// Row tiles per workgroup of the fused chain kernels (the W <= 256, 16-wave instantiations).  A workgroup with two 16-row tiles takes
// ~1.76x one tile (measured at c3: the phases are bound by the per-row loads / stores of the saved activations, not by the shared weight
// fragments), so two tiles pay only when they save a round of the 256 CUs: 257..512 tiles (c5's per-GPU share: 248 -> 215 us), not
// 513..768 (c3).  Four tiles never pay (3.8x).  MVSDF_CHAIN_MT=1|2 overrides (dev A/B).
static int mv_chain_mt(int tiles16) {
    static const int env = [] { const char* e = mv_dev_env("MVSDF_CHAIN_MT"); return e ? atoi(e) : 0; }();
    if (env == 1 || env == 2) return env;
    const int rounds1 = (tiles16 + 255) / 256, rounds2 = (tiles16 + 511) / 512;
    return 1.76 * rounds2 < rounds1 ? 2 : 1;
}
/* Does the fused forward chain over the rows [E, M) alone need fewer / shorter rounds of workgroups than over [0, M)?  (the same model as
 * mv_chain_mt: a round of two-tile workgroups costs 1.76 rounds of one-tile workgroups)  The step asks before it moves the E sample rows beside the tracer. */
int mv_chain_split_pays(const MvsdfNetDesc* d, int E, int M) {
    MvNet net;
    if (E < 16 || E >= M || mv_make_net(d, &net) || mv_chain_ntw(net) != 2 || mv_chain_w8()) return 0;
    auto cost = [](int tiles16) { const int r1 = (tiles16 + 255) / 256, r2 = (tiles16 + 511) / 512; return 1.76 * r2 < r1 ? 1.76 * r2 : 1.0 * r1; };
    return cost((M - E + 15) / 16) < cost((M + 15) / 16) ? 1 : 0;
}

// The fused chains in the three-term bf16 arithmetic (chain_x3.h) run when both descriptors carry the packs (MvsdfNetDesc.wx3); the dev library's
// MVSDF_CHAIN_X3=0 keeps the fp32-input MFMA chains (A/B, tests/test_gpu_alt_paths.py).
static bool mv_chain_x3_on() {
    static const int env = [] {
        const char* e = mv_dev_env("MVSDF_CHAIN_X3");
        const char* f = mv_dev_env("MVSDF_FUSE");                 // (the dev switches that pick the per-layer / split fp32 launches mean the fp32 arithmetic everywhere)
        if ((f && atoi(f) == 0) || mv_dev_env("MVSDF_SPLIT_CHAINS") || mv_dev_env("MVSDF_CHAIN_W8")) return 0;
        return e ? atoi(e) : 1;
    }();
    return env != 0;
}
int mv_chain_x3_enabled() { return mv_chain_x3_on() ? 1 : 0; }
// -> 0 and both nets when the x3 chains can run this network (needT: the transposed packs too)
static int mv_x3_nets(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, bool needT, MvNetBf* xn, MvNetBf* xnT) {
    if (!mv_chain_x3_on() || mv_make_net_x3(d, xn)) return 1;
    if (needT) { if (!dT || mv_make_net_x3(dT, xnT)) return 1; }
    else *xnT = *xn;
    xnT->skip_mask = xn->skip_mask; xnT->multires = xn->multires;
    const int S16 = xn->S > xnT->S ? xn->S : xnT->S;
    xn->S = xnT->S = S16;
    return 0;
}
// row tiles per workgroup of the x3 chains: two tiles share the weight stream that bounds a phase (tile_engine_bf16s.h: 56 vs 41 us per evaluation), so
// they pay as soon as they save a round of the 256 CUs
static int mv_chain_mt_x3(int tiles16) {
    static const int env = [] { const char* e = mv_dev_env("MVSDF_CHAIN_MT"); return e ? atoi(e) : 0; }();
    if (env == 1 || env == 2) return env;
    const int rounds1 = (tiles16 + 255) / 256, rounds2 = (tiles16 + 511) / 512;
    return 1.45 * rounds2 < rounds1 ? 2 : 1;
}

// Carried-ring depth (k-blocks of the next phase's weights requested early, chain_x3.h) of the x3 chains.  One row tile per workgroup (<= 256 tiles: c2): 4 --
// k_chain_fwd_x3 111 -> 97 us, the c2 step 1.503 -> 1.485 ms.  Two row tiles: 0 (the rolling fetch): with 2 the c5-share step went 1.51-1.55 -> 1.57-1.61 ms and c3
// 4.04 -> 4.20 ms -- at those sizes the sample rows' chain runs BESIDE the tracer (mv_chain_split_pays), and a chain that keeps the L2 busy through its epilogues
// takes that bandwidth from the tracer's own weight stream (k_ray_samples 0.446 -> 0.478 ms, k_sphere_trace 1.17 -> 1.27 ms at c3).
#ifndef MV_X3_PD1
#define MV_X3_PD1 4
#endif
#ifndef MV_X3_PD2
#define MV_X3_PD2 0
#endif

#define MV_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mv_check(e_, #expr); } while (0)
// dynamic-LDS limit of one kernel instance, raised only when a launch needs more than any launch before it (`hw`: one static high-water mark per site;
// the runtime call costs a few microseconds of host time, a training step would make four of them)
template <class K>
static hipError_t mv_lds_limit(K kern, size_t bytes, size_t& hw) {
    if (bytes <= hw) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) hw = bytes;
    return e;
}

extern "C" {

size_t mvsdf_sdf_ctx_floats(const MvsdfNetDesc* d, int M, int Mg) {
    MvNet net;
    if (mv_make_net(d, &net)) return 0;
    return sdf_ctx_layout(net, M, Mg).total;
}
size_t mvsdf_sdf_bwd_ws_floats(const MvsdfNetDesc* d, int Mb) {
    MvNet net;
    if (mv_make_net(d, &net)) return 0;
    return sdf_bwd_layout(net, Mb).total;
}

int mvsdf_sdf_forward(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, const float* x, int M, int Mg, float* y, float* nrm, float* ctx,
                      void* stream) {
    return mv_sdf_forward_gather(d, dT, x, nullptr, M, Mg, 0, M, y, nrm, ctx, stream);
}

}  // extern "C"

/* mvsdf_sdf_forward whose rows are gathered inside the fused chain kernel (g != NULL: the rows [eikonal | on-surface | jittered | pts[perm]], also
 * written to g->x_out) -- the training step's x_eval without a gather launch.  -> 1 when g was given but the per-layer route had to run: nothing was
 * launched, the caller gathers itself and calls again with x.
 * [r_begin, r_end) is the row range THIS launch evaluates (buffers and M / Mg always describe all the rows): the step evaluates its sample rows, which
 * do not depend on the tracer, beside the tracer and the rows of the rays after it.  Rows are independent, so the split changes no bit; only the
 * fused chain accepts a proper sub-range (-> 1 otherwise, nothing launched). */
int mv_sdf_forward_gather(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, const float* x, const void* gather, int M, int Mg, int r_begin, int r_end,
                          float* y, float* nrm, float* ctx, void* stream) {
    const FwdGather* g = (const FwdGather*)gather;
    MvNet net, netT;
    int rc = mv_make_net(d, &net);
    if (rc) return rc;
    if ((!x && !g) || !y || !ctx || M <= 0 || Mg < 0 || Mg > M || r_begin < 0 || r_end > M || r_begin >= r_end)
        return mv_fail(-1, "mvsdf_sdf_forward: bad arguments");
    const bool sub = r_begin > 0 || r_end < M;
    if (Mg > 0) {
        rc = mv_make_net_mode(dT, &netT, 2);
        if (rc) return rc;
        if (!nrm || !d->w[d->n_layers - 1]) return mv_fail(-1, "mvsdf_sdf_forward: normals requested but nrm / row-major last-layer weights missing");
    }
    hipStream_t s = (hipStream_t)stream;
    const SdfLayout lo = sdf_ctx_layout(net, M, Mg);
    const int nl = lo.nl, S = Mg > 0 ? stride_for(net, netT) : net.S;
    float* H0 = ctx + lo.H0;
    static int fuse_fwd = -1;
    if (fuse_fwd < 0) { const char* e = mv_dev_env("MVSDF_FUSE"); fuse_fwd = e ? atoi(e) : 1; }
    const int ntw_f = mv_chain_ntw(net);
    MvNetBf xn, xnT;
    if (fuse_fwd && ntw_f && nl >= 2 && net.L[nl - 1].NT <= 8 * ntw_f * 4 && !mv_x3_nets(d, dT, Mg > 0, &xn, &xnT)) {   // ... in the three-term bf16 arithmetic
        FwdArgsX3 f;
        memset(&f, 0, sizeof(f));
        f.net = xn; f.netT = xnT;
        const int Mr = r_end - r_begin;
        f.S = xn.S; f.M = r_end; f.Mg = Mg < r_end ? Mg : r_end; f.row_base = r_begin; f.ld0 = lo.ld0; f.x = x; f.H0 = H0;
        for (int l = 1; l < nl; ++l) f.A[l] = ctx + lo.A[l];
        for (int l = 0; l < nl - 1; ++l) { f.Z[l] = ctx + lo.Z[l]; f.Sg[l] = ctx + lo.Sg[l]; }
        for (int l = 1; l < nl - 1; ++l) f.U[l] = ctx + lo.U[l];
        f.G0 = ctx + lo.G0; f.y = y; f.ldy = net.L[nl - 1].N; f.w_last_row0 = d->w[nl - 1]; f.nrm = nrm;
        if (g) f.g = *g;
        // (hidden width 257 .. 512, two row tiles: 8 waves x 4 column tiles -- a 16-row tile streams 1.57 MB of weight terms per phase, more than its matrix
        // instructions take; at 37 000 rows of the 8x512 network 4208 (fp32 chain) / 3842 (one tile) / 2761 us (two tiles))
        const int mt = mv_chain_mt_x3((Mr + 15) / 16);
        const size_t lds = (size_t)3 * 16 * mt * xn.S * 2 + ((size_t)2 * ((16 * mt * lo.d0 + 3) & ~3) + 16 * mt * 4) * sizeof(float);
        const dim3 grid((Mr + 16 * mt - 1) / (16 * mt));
        if (ntw_f == 4 && mt == 2) {
            { static size_t hw = 0; MV_TRY(mv_lds_limit(k_chain_fwd_x3<2, 4, 8>, lds, hw)); }
            hipLaunchKernelGGL((k_chain_fwd_x3<2, 4, 8>), grid, dim3(512), lds, s, f);
            return mv_check(hipGetLastError(), "mvsdf_sdf_forward (x3 chain)");
        }
        // hidden width <= 256: 16 waves x 1 column tile; up to 512: 16 waves x 2 tiles
        if (mt == 2) {
            { static size_t hw = 0; MV_TRY(mv_lds_limit(k_chain_fwd_x3<2, 1, 16, MV_X3_PD2>, lds, hw)); }
            hipLaunchKernelGGL((k_chain_fwd_x3<2, 1, 16, MV_X3_PD2>), grid, dim3(1024), lds, s, f);
        }
        else if (ntw_f == 2) hipLaunchKernelGGL((k_chain_fwd_x3<1, 1, 16, MV_X3_PD1>), grid, dim3(1024), lds, s, f);
        else hipLaunchKernelGGL((k_chain_fwd_x3<1, 2, 16>), grid, dim3(1024), lds, s, f);
        return mv_check(hipGetLastError(), "mvsdf_sdf_forward (x3 chain)");
    }
    if (fuse_fwd && ntw_f && net.L[nl - 1].NT <= 8 * ntw_f * 4) {                  // value + normal of a row tile in one launch
        FwdArgs f;
        memset(&f, 0, sizeof(f));
        f.net = net; if (Mg > 0) f.netT = netT;
        const int Mr = r_end - r_begin;
        f.S = S; f.M = r_end; f.Mg = Mg < r_end ? Mg : r_end; f.row_base = r_begin; f.ld0 = lo.ld0; f.x = x; f.H0 = H0;
        for (int l = 1; l < nl; ++l) f.A[l] = ctx + lo.A[l];
        for (int l = 0; l < nl - 1; ++l) { f.Z[l] = ctx + lo.Z[l]; f.Sg[l] = ctx + lo.Sg[l]; }
        for (int l = 1; l < nl - 1; ++l) f.U[l] = ctx + lo.U[l];
        f.G0 = ctx + lo.G0; f.y = y; f.ldy = net.L[nl - 1].N; f.w_last_row0 = d->w[nl - 1]; f.nrm = nrm;
        if (g) f.g = *g;
        constexpr int MTC = 1, NWC = 8;
        const size_t lds = ((size_t)16 * MTC * S + 2 * ((16 * MTC * lo.d0 + 3) & ~3) + 16 * MTC * 4) * sizeof(float);
        // 16 waves per workgroup (one or two column tiles each): these launches are single waves of one-tile workgroups, i.e. chains of
        // dependent layer phases; twice the waves halve every wave's share of the global loads / stores and of the epilogue between
        // two GEMMs (measured 148 -> 127 us here, 173 -> 143 us for the backward pass).  MVSDF_CHAIN_W8=1: 8 waves (dev A/B).
        const bool w8 = mv_chain_w8();
        const dim3 grid((Mr + 16 * MTC - 1) / (16 * MTC));
        const int mt = (ntw_f == 2 && !w8) ? mv_chain_mt((Mr + 15) / 16) : 1;
        if (mt > 1) {
            const size_t ldsm = ((size_t)16 * mt * S + 2 * ((16 * mt * lo.d0 + 3) & ~3) + 16 * mt * 4) * sizeof(float);
            const dim3 gridm((Mr + 16 * mt - 1) / (16 * mt));
            { static size_t hw = 0; MV_TRY(mv_lds_limit(k_chain_fwd<2, 1, 16>, ldsm, hw)); }
            hipLaunchKernelGGL((k_chain_fwd<2, 1, 16>), gridm, dim3(1024), ldsm, s, f);
        }
        else if (ntw_f == 2 && !w8) hipLaunchKernelGGL((k_chain_fwd<MTC, 1, 16>), grid, dim3(1024), lds, s, f);
        else if (ntw_f == 2) hipLaunchKernelGGL((k_chain_fwd<MTC, 2, NWC>), grid, dim3(64 * NWC), lds, s, f);
        else if (!w8) hipLaunchKernelGGL((k_chain_fwd<MTC, 2, 16>), grid, dim3(1024), lds, s, f);
        else hipLaunchKernelGGL((k_chain_fwd<MTC, 4, NWC>), grid, dim3(64 * NWC), lds, s, f);
        return mv_check(hipGetLastError(), "mvsdf_sdf_forward");
    }
    if (g || sub) return 1;                                      // per-layer route: rows must be materialised by the caller, all of them at once
    hipLaunchKernelGGL(k_pe_global, dim3((M * (3 * net.multires + 1) + 255) / 256), dim3(256), 0, s, x, M, net.multires, H0, lo.ld0);
    for (int l = 0; l < nl - 1; ++l) {                                            // hidden layers (idr.py:82-92)
        LayerArgs a = base_args(net.L[l], S, M);
        a.A = l == 0 ? H0 : ctx + lo.A[l]; a.lda = l == 0 ? lo.ld0 : net.L[l].K;
        a.out1 = ctx + lo.Z[l]; a.ld1 = net.L[l].N;
        a.out0 = ctx + lo.A[l + 1]; a.ld0 = net.L[l + 1].K;
        a.skip_next = mv_skip_at(net.skip_mask, l + 1); a.d0 = lo.d0; a.pe = H0; a.ldpe = lo.ld0;
        MV_TRY((launch_layer<PRO_PLAIN, EPI_SOFTPLUS>(a, s)));
    }
    {
        LayerArgs a = base_args(net.L[nl - 1], S, M);
        a.A = ctx + lo.A[nl - 1]; a.lda = net.L[nl - 1].K; a.out0 = y; a.ld0 = net.L[nl - 1].N;
        MV_TRY((launch_layer<PRO_PLAIN, EPI_BIAS>(a, s)));
    }
    if (Mg > 0) {                                                                 // normal = VJP of output 0 (idr.py:96-107)
        const int sk1 = mv_single_skip(net);
        if (sk1 == -2) return mv_fail(-4, "mvsdf_sdf_forward: several skip connections need the fused chain kernels (MVSDF_FUSE unset)");
        const float* w8 = d->w[nl - 1];                                           // row 0 of the last layer = u_{nl-1}
        if (sk1 == nl - 1)                                                        // skip into the last Linear: its PE columns start the PE adjoint
            hipLaunchKernelGGL(k_pe_adj_top, dim3((Mg * lo.d0 + 255) / 256), dim3(256), 0, s, w8, net.L[nl - 1].K - lo.d0, lo.d0, Mg, ctx + lo.E, lo.ld0);
        for (int l = nl - 2; l >= 0; --l) {
            LayerArgs a = base_args(netT.L[l], S, Mg);
            a.Z = ctx + lo.Z[l]; a.ldz = net.L[l].N;
            const bool top = (l == nl - 2);
            if (top) { a.bcast = w8; a.bcast_sqrt2 = (sk1 == nl - 1); } else { a.U = ctx + lo.U[l + 1]; a.ldu = net.L[l].N; }
            a.out2 = ctx + lo.Sg[l]; a.ld2 = net.L[l].N;                            // keep s_l for the weight gradient
            if (l == sk1) {
                a.csplit = net.L[l].K - lo.d0; a.scale_sqrt2 = 1;
                a.out0 = ctx + lo.U[l]; a.ld0 = net.L[l - 1].N;
                a.out1 = ctx + lo.E; a.ld1 = lo.ld0;
            } else if (l == 0) {
                a.csplit = net.L[0].K;
                if (sk1 > 0) { a.add = ctx + lo.E; a.ldadd = lo.ld0; }
                a.out0 = ctx + lo.G0; a.ld0 = lo.ld0;
            } else {
                a.csplit = net.L[l].K;
                a.out0 = ctx + lo.U[l]; a.ld0 = net.L[l - 1].N;
            }
            if (top) MV_TRY((launch_layer<PRO_SIG_BCAST, EPI_SPLIT>(a, s)));
            else MV_TRY((launch_layer<PRO_SIG_MUL, EPI_SPLIT>(a, s)));
        }
        hipLaunchKernelGGL(k_pe_normal, dim3((Mg * 3 + 255) / 256), dim3(256), 0, s, H0, lo.ld0, ctx + lo.G0, lo.ld0, Mg, net.multires, nrm);
    }
    return mv_check(hipGetLastError(), "mvsdf_sdf_forward");
}

extern "C" {

/* Backward over rows [row0, row0 + Mb) of a forward context made with (M, Mg).  dy[Mb][Nout] (required), dn[Mb][3] or NULL
 * (rows must lie inside [0, Mg) when dn is given).  Outputs: dW_cat / db_cat (all layers concatenated, row-major [N][K]; both NULL
 * = input adjoint only), dx[Mb][3] or NULL. */
int mvsdf_sdf_backward(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, const float* x, int M, int Mg, int row0, int Mb, const float* dy,
                       const float* dn, const float* ctx, float* dW_cat, float* db_cat, float* dx, float* ws, void* stream) {
    MvNet net, netT;
    int rc = mv_make_net(d, &net);
    if (rc) return rc;
    rc = mv_make_net_mode(dT, &netT, 2);
    if (rc) return rc;
    if (!x || !dy || !ctx || !ws || Mb <= 0 || row0 < 0 || row0 + Mb > M || (dn && row0 + Mb > Mg) || ((dW_cat == nullptr) != (db_cat == nullptr)))
        return mv_fail(-1, "mvsdf_sdf_backward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const SdfLayout lo = sdf_ctx_layout(net, M, Mg);
    const SdfBwdLayout bl = sdf_bwd_layout(net, Mb);
    const int nl = lo.nl, S = stride_for(net, netT), sk = mv_single_skip(net);     // per-layer fallback launches below: one skip layer at most
    const size_t r0 = (size_t)row0;
    const float* H0 = ctx + lo.H0 + r0 * lo.ld0;                 // every context tensor is [rows][width]: a row offset is a pointer offset
    const float* w8 = d->w[nl - 1];
    if (dn && !w8) return mv_fail(-1, "mvsdf_sdf_backward: row-major last-layer weights missing");
    auto Aof = [&](int l) { return l == 0 ? H0 : ctx + lo.A[l] + r0 * net.L[l].K; };
    auto ldA = [&](int l) { return l == 0 ? lo.ld0 : net.L[l].K; };
    auto Zof = [&](int l) { return ctx + lo.Z[l] + r0 * net.L[l].N; };
    auto Uof = [&](int l) { return ctx + lo.U[l] + r0 * net.L[l - 1].N; };       // u_l has width out_{l-1}
    auto Sof = [&](int l) { return ctx + lo.Sg[l] + r0 * net.L[l].N; };
    const float* G0 = ctx + lo.G0 + r0 * lo.ld0;
    // ---- the whole pass in one launch per row tile (gbar_0, E.1, E.2, input adjoint) when the chain kernels fit the network ----
    static int fuse_all = -1;
    if (fuse_all < 0) { const char* e = mv_dev_env("MVSDF_FUSE"); fuse_all = e ? atoi(e) : 1; if (mv_dev_env("MVSDF_SPLIT_CHAINS")) fuse_all = 0; }
    const int ntw_b = mv_chain_ntw(net);
    bool chains_done = false;
    MvNetBf xn, xnT;
    if (fuse_all >= 1 && ntw_b && nl >= 2 && !mv_x3_nets(d, dT, true, &xn, &xnT)) {     // the whole pass in the three-term bf16 arithmetic (chain_x3.h)
        ChainArgsX3 c;
        memset(&c, 0, sizeof(c));
        c.net = xn; c.netT = xnT; c.S = xn.S; c.M = Mb; c.row_ld0 = lo.ld0;
        c.dy = dy; c.ld_dy = net.L[nl - 1].N; c.w_last_row0 = w8;
        for (int l = 0; l < nl - 1; ++l) {
            c.Z[l] = Zof(l); c.ZB[l] = ws + bl.ZB[l];
            c.ZB2[l] = dn ? ws + bl.ZB2[l] : nullptr; c.ZB2o[l] = ws + bl.ZB2[l];
            if (l + 1 < nl - 1) c.U[l + 1] = Uof(l + 1);
            c.VB[l + 1] = ws + bl.VB[l + 1];
        }
        c.H0B = ws + bl.H0B; c.H0 = H0; c.G0 = G0; c.dn_in = dn; c.VB0w = ws + bl.VB[0]; c.dx = dx;
        const size_t lds = (size_t)3 * 16 * xn.S * 2 + (size_t)16 * lo.d0 * sizeof(float);
        const dim3 grid((Mb + 15) / 16);
        if (ntw_b == 2) hipLaunchKernelGGL((k_chain_bwd_x3<1, 1, 16, MV_X3_PD1>), grid, dim3(1024), lds, s, c);
        else hipLaunchKernelGGL((k_chain_bwd_x3<1, 2, 16>), grid, dim3(1024), lds, s, c);
        MV_TRY(hipGetLastError());
        chains_done = true;
    }
    else if (fuse_all >= 1 && ntw_b) {                          // MVSDF_SPLIT_CHAINS=1: the separate E.1 / E.2 launches (dev A/B)
        ChainArgs c;
        memset(&c, 0, sizeof(c));
        c.net = net; c.netT = netT; c.S = S; c.M = Mb; c.row_ld0 = lo.ld0;
        c.dy = dy; c.ld_dy = net.L[nl - 1].N; c.w_last_row0 = w8;
        for (int l = 0; l < nl - 1; ++l) {
            c.Z[l] = Zof(l); c.ZB[l] = ws + bl.ZB[l];
            c.ZB2[l] = dn ? ws + bl.ZB2[l] : nullptr; c.ZB2o[l] = ws + bl.ZB2[l];
            if (l + 1 < nl - 1) c.U[l + 1] = Uof(l + 1);
            c.VB[l + 1] = ws + bl.VB[l + 1];
        }
        c.H0B = ws + bl.H0B; c.H0 = H0; c.G0 = G0; c.dn_in = dn; c.VB0w = ws + bl.VB[0]; c.dx = dx;
        constexpr int MTC = 1, NWC = 8;
        const size_t lds = (size_t)16 * MTC * (S + lo.d0) * sizeof(float);
        const bool w8 = mv_chain_w8();
        const dim3 grid((Mb + 16 * MTC - 1) / (16 * MTC));
        if (ntw_b == 2 && !w8) hipLaunchKernelGGL((k_chain_bwd<MTC, 1, 16>), grid, dim3(1024), lds, s, c);
        else if (ntw_b == 2) hipLaunchKernelGGL((k_chain_bwd<MTC, 2, NWC>), grid, dim3(64 * NWC), lds, s, c);
        else if (!w8) hipLaunchKernelGGL((k_chain_bwd<MTC, 2, 16>), grid, dim3(1024), lds, s, c);
        else hipLaunchKernelGGL((k_chain_bwd<MTC, 4, NWC>), grid, dim3(64 * NWC), lds, s, c);
        MV_TRY(hipGetLastError());
        chains_done = true;
    }
    if (!chains_done && sk == -2) return mv_fail(-4, "mvsdf_sdf_backward: several skip connections need the fused chain kernels (MVSDF_FUSE / MVSDF_SPLIT_CHAINS unset)");
    // ---- E.1: adjoint of the normal chain (ascending) ----
    if (dn && !chains_done) {
        hipLaunchKernelGGL(k_pe_normal_bwd, dim3((Mb * (3 * net.multires + 1) + 255) / 256), dim3(256), 0, s, H0, lo.ld0, dn, Mb,
                           net.multires, ws + bl.VB[0], lo.ld0, sk > 0 ? ws + bl.VB[sk] : nullptr, sk > 0 ? net.L[sk].K : 0,
                           sk > 0 ? net.L[sk].K - lo.d0 : 0);
        static int fuse1_env = -1;
        if (fuse1_env < 0) { const char* e = mv_dev_env("MVSDF_FUSE"); fuse1_env = e ? atoi(e) : 1; }
        const int ntw_1 = mv_chain_ntw(net);
        if (fuse1_env && ntw_1) {                                // the whole ascending chain in one launch
            ChainArgs c;
            memset(&c, 0, sizeof(c));
            c.net = net; c.netT = netT; c.S = S; c.M = Mb; c.row_ld0 = lo.ld0;
            c.VB0 = ws + bl.VB[0]; c.w_last_row0 = w8;
            for (int l = 0; l < nl - 1; ++l) {
                c.Z[l] = Zof(l);
                if (l + 1 < nl - 1) c.U[l + 1] = Uof(l + 1);
                c.VB[l + 1] = ws + bl.VB[l + 1];
                c.ZB2o[l] = ws + bl.ZB2[l];
            }
            constexpr int MTC = 1, NWC = 8;
            const size_t lds = (size_t)16 * MTC * (S + lo.d0) * sizeof(float);
            if (ntw_1 == 2) hipLaunchKernelGGL((k_chain_e1<MTC, 2, NWC>), dim3((Mb + 16 * MTC - 1) / (16 * MTC)), dim3(64 * NWC), lds, s, c);
            else hipLaunchKernelGGL((k_chain_e1<MTC, 4, NWC>), dim3((Mb + 16 * MTC - 1) / (16 * MTC)), dim3(64 * NWC), lds, s, c);
            MV_TRY(hipGetLastError());
        } else
        for (int l = 0; l < nl - 1; ++l) {
            LayerArgs a = base_args(net.L[l], S, Mb);
            a.A = ws + bl.VB[l]; a.lda = ldA(l);
            a.Z = Zof(l); a.ldz = net.L[l].N;
            if (l == nl - 2) { a.bcast = w8; a.bcast_sqrt2 = (sk == nl - 1); } else { a.U = Uof(l + 1); a.ldu = net.L[l].N; }
            a.out0 = ws + bl.VB[l + 1]; a.ld0 = net.L[l + 1].K;
            a.out1 = ws + bl.ZB2[l]; a.ld1 = net.L[l].N;
            a.skip_next = (l + 1 == sk);
            MV_TRY((launch_layer<PRO_PLAIN, EPI_SBAR>(a, s)));
        }
    }
    // ---- E.2: adjoint of the value chain (descending) ----
    if (!chains_done) {
    static int fuse_env = -1;
    if (fuse_env < 0) { const char* e = mv_dev_env("MVSDF_FUSE"); fuse_env = e ? atoi(e) : 1; }
    const int ntw_2 = mv_chain_ntw(net);
    if (fuse_env && ntw_2) {                                    // all layers in one launch, the running adjoint stays in LDS
        ChainArgs c;
        memset(&c, 0, sizeof(c));
        c.net = net; c.netT = netT; c.S = S; c.M = Mb; c.row_ld0 = lo.ld0;
        c.dy = dy; c.ld_dy = net.L[nl - 1].N;
        for (int l = 0; l < nl - 1; ++l) { c.Z[l] = Zof(l); c.ZB2[l] = dn ? ws + bl.ZB2[l] : nullptr; c.ZB[l] = ws + bl.ZB[l]; }
        c.H0B = ws + bl.H0B;
        constexpr int MTC = 1, NWC = 8;
        const size_t lds = (size_t)16 * MTC * (S + lo.d0) * sizeof(float);
        if (ntw_2 == 2) hipLaunchKernelGGL((k_chain_e2<MTC, 2, NWC>), dim3((Mb + 16 * MTC - 1) / (16 * MTC)), dim3(64 * NWC), lds, s, c);
        else hipLaunchKernelGGL((k_chain_e2<MTC, 4, NWC>), dim3((Mb + 16 * MTC - 1) / (16 * MTC)), dim3(64 * NWC), lds, s, c);
        MV_TRY(hipGetLastError());
    } else {
    int cur = 0;
    {
        LayerArgs a = base_args(netT.L[nl - 1], S, Mb);
        a.A = dy; a.lda = net.L[nl - 1].N;
        a.csplit = net.L[nl - 1].K; a.out0 = ws + bl.HB[cur]; a.ld0 = net.L[nl - 1].K;
        if (sk == nl - 1) {                                                        // skip into the last Linear: hidden part / PE part, both / sqrt(2)
            a.csplit = net.L[nl - 1].K - lo.d0; a.scale_sqrt2 = 1; a.ld0 = net.L[nl - 2].N;
            a.out1 = ws + bl.H0B; a.ld1 = lo.ld0;
        }
        MV_TRY((launch_layer<PRO_PLAIN, EPI_SPLIT>(a, s)));
    }
    if (sk <= 0) MV_TRY(hipMemsetAsync(ws + bl.H0B, 0, (size_t)Mb * lo.ld0 * sizeof(float), s));
    for (int l = nl - 2; l >= 0; --l) {
        LayerArgs a = base_args(netT.L[l], S, Mb);
        a.Z = Zof(l); a.ldz = net.L[l].N;
        a.U = ws + bl.HB[cur]; a.ldu = net.L[l].N;
        if (dn) { a.A = ws + bl.ZB2[l]; a.lda = net.L[l].N; a.Mg = Mb; }
        a.out2 = ws + bl.ZB[l]; a.ld2 = net.L[l].N;
        if (l == sk) {
            a.csplit = net.L[l].K - lo.d0; a.scale_sqrt2 = 1;
            a.out0 = ws + bl.HB[cur ^ 1]; a.ld0 = net.L[l - 1].N;
            a.out1 = ws + bl.H0B; a.ld1 = lo.ld0;
        } else if (l == 0) {
            a.csplit = net.L[0].K;
            a.add = ws + bl.H0B; a.ldadd = lo.ld0;
            a.out0 = ws + bl.H0B; a.ld0 = lo.ld0;
        } else {
            a.csplit = net.L[l].K;
            a.out0 = ws + bl.HB[cur ^ 1]; a.ld0 = net.L[l - 1].N;
        }
        MV_TRY((launch_layer<PRO_ZBAR, EPI_SPLIT>(a, s)));
        cur ^= 1;
    }
    }
    }   // !chains_done
    // ---- weight / bias gradients: W_l = zbar_l^T a_l (+ s_l^T vbar_l), every layer in one launch + one reduction ----
    if (dW_cat) {                                                                  // dW_cat == NULL: input adjoint only
        WgradNetArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.n_layers = nl; wa.chunk = bl.chunk;
        for (int l = 0; l < nl; ++l) {
            WgradLayer& L = wa.L[l];
            const bool last = (l == nl - 1);
            L.No = net.L[l].N; L.Ki = net.L[l].K;
            L.P1 = last ? dy : ws + bl.ZB[l]; L.ldp1 = L.No;
            L.Q1 = Aof(l); L.ldq1 = ldA(l);
            if (dn && !last) { L.P2 = Sof(l); L.ldp2 = L.No; L.Q2 = ws + bl.VB[l]; L.ldq2 = ldA(l); }
        }
        wgrad_net_layers(wa, 0, nl, Mb, bl.nchunks, ws + bl.slabA, ws + bl.bslab, dW_cat, db_cat);
        if (dn) {                                                                  // W_last[0, :] += sum_rows ubar_last   (E.1 end)
            const int Ki = net.L[nl - 1].K;
            wgrad_colsum(wa, ws + bl.VB[nl - 1], Ki, Mb, Ki, nl - 1, bl.nchunks, ws + bl.slabB);
        }
        MV_TRY(launch_wgrad_net(wa, s));
    }
    // ---- E.3: input adjoint ----
    if (dx && !chains_done)
        hipLaunchKernelGGL(k_pe_input_bwd, dim3((Mb * 3 + 255) / 256), dim3(256), 0, s, H0, lo.ld0, ws + bl.H0B, lo.ld0, G0, lo.ld0,
                           dn, Mb, net.multires, dx);
    return mv_check(hipGetLastError(), "mvsdf_sdf_backward");
}

}  // extern "C"

// ---- the training step's SDF backward as two launches of chain passes instead of three sequential ones (functional._IdrStep.backward) ----
// fill the arguments of one fused chain pass over rows [row0, row0 + Mb) of a forward context
template <class NETX>
static void fill_chain_args(ChainArgsT<NETX>& c, const MvNet& net, const NETX& xnet, const NETX& xnetT, int S, const SdfLayout& lo, const SdfBwdLayout& bl, const float* ctx,
                            int row0, int Mb, const float* dy, const float* dn, float* ws, float* dx, const float* w8) {
    memset(&c, 0, sizeof(c));
    const int nl = lo.nl;
    const size_t r0 = (size_t)row0;
    c.net = xnet; c.netT = xnetT; c.S = S; c.M = Mb; c.row_ld0 = lo.ld0;
    c.dy = dy; c.ld_dy = net.L[nl - 1].N; c.w_last_row0 = w8;
    for (int l = 0; l < nl - 1; ++l) {
        c.Z[l] = ctx + lo.Z[l] + r0 * net.L[l].N; c.ZB[l] = ws + bl.ZB[l];
        c.ZB2[l] = dn ? ws + bl.ZB2[l] : nullptr; c.ZB2o[l] = ws + bl.ZB2[l];
        if (l + 1 < nl - 1) c.U[l + 1] = ctx + lo.U[l + 1] + r0 * net.L[l].N;
        c.VB[l + 1] = ws + bl.VB[l + 1];
    }
    c.H0B = ws + bl.H0B; c.H0 = ctx + lo.H0 + r0 * lo.ld0; c.G0 = ctx + lo.G0 + r0 * lo.ld0; c.dn_in = dn; c.VB0w = ws + bl.VB[0]; c.dx = dx;
}

extern "C" {

/* Pass A: full first/second-order backward over rows [0, MbA) with upstream (dyA, dnA); keeps every per-layer adjoint in wsA
 * (mvsdf_sdf_bwd_ws_floats(net, MbA) floats) for mvsdf_sdf_backward_finish -- no weight gradients yet.
 * Pass X: input adjoint only over rows [row0X, row0X + MbX) with upstream (dyX, dnX) -> dx[MbX][3]; scratch wsX (mvsdf_sdf_bwd_ws_floats(net, MbX)).
 * The two passes are independent and run as ONE grid.  Returns -3 if the fused chain kernels do not cover this network (caller falls back to
 * mvsdf_sdf_backward). */
int mvsdf_sdf_backward_pair(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int M, int Mg, int MbA, const float* dyA, const float* dnA, float* wsA,
                            int row0X, int MbX, const float* dyX, const float* dnX, float* wsX, float* dx, const float* ctx, void* stream) {
    return mv_sdf_backward_pair_cnt(d, dT, M, Mg, MbA, dyA, dnA, wsA, row0X, MbX, dyX, dnX, wsX, dx, ctx, nullptr, 0, stream);
}

}  // extern "C"

// ... with device-side counts (the deferred step, step_internal.h): cnt != NULL -> pass A covers rows [0, row0X + cnt[0]) and pass X the cnt[0] rows from row0X;
// MbA / MbX are then the UPPER BOUNDS (row0X + MbX == MbA) the workspaces, their layouts and the grid are sized for, and n_hint (a recent cnt[0], or MbX) only
// picks the kernel form.  Same kernels, same arithmetic per row.
int mv_sdf_backward_pair_cnt(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int M, int Mg, int MbA, const float* dyA, const float* dnA, float* wsA,
                             int row0X, int MbX, const float* dyX, const float* dnX, float* wsX, float* dx, const float* ctx, const long long* cnt, int n_hint,
                             void* stream) {
    MvNet net, netT;
    int rc = mv_make_net(d, &net);
    if (rc) return rc;
    rc = mv_make_net_mode(dT, &netT, 2);
    if (rc) return rc;
    if (!dyA || !dnA || !wsA || !dyX || !wsX || !dx || !ctx || MbA <= 0 || MbX <= 0 || MbA > Mg || row0X < 0 || row0X + MbX > Mg || Mg > M)
        return mv_fail(-1, "mvsdf_sdf_backward_pair: bad arguments");
    if (cnt && row0X + MbX != MbA) return mv_fail(-1, "mv_sdf_backward_pair_cnt: the bounds of the two passes must end on the same row");
    if (n_hint < 0 || n_hint > MbX) n_hint = MbX;
    const int tiles_hint = cnt ? (row0X + n_hint + 15) / 16 + (n_hint + 15) / 16 : (MbA + 15) / 16 + (MbX + 15) / 16;
    const int ntw_b = mv_chain_ntw(net);
    if (!ntw_b) return mv_fail(-3, "mvsdf_sdf_backward_pair: network too wide for the fused chain kernels");
    const float* w8 = d->w[net.n_layers - 1];
    if (!w8) return mv_fail(-1, "mvsdf_sdf_backward_pair: row-major last-layer weights missing");
    hipStream_t s = (hipStream_t)stream;
    const SdfLayout lo = sdf_ctx_layout(net, M, Mg);
    const int S = stride_for(net, netT);
    MvNetBf xn, xnT;
    if (net.n_layers >= 2 && !mv_x3_nets(d, dT, true, &xn, &xnT)) {                 // both passes in the three-term bf16 arithmetic (chain_x3.h)
        ChainArgsX3 a, b;
        fill_chain_args(a, net, xn, xnT, xn.S, lo, sdf_bwd_layout(net, MbA), ctx, 0, MbA, dyA, dnA, wsA, nullptr, w8);
        fill_chain_args(b, net, xn, xnT, xn.S, lo, sdf_bwd_layout(net, MbX), ctx, row0X, MbX, dyX, dnX, wsX, dx, w8);
        a.cnt = b.cnt = cnt; a.cnt_base = row0X; b.cnt_base = 0;
        const int mt = mv_chain_mt_x3(tiles_hint);
        const size_t lds = (size_t)3 * 16 * mt * xn.S * 2 + (size_t)16 * mt * lo.d0 * sizeof(float);
        const int na = (MbA + 16 * mt - 1) / (16 * mt), nb = (MbX + 16 * mt - 1) / (16 * mt);
        const dim3 grid(na + nb);
        if (mt == 2 && ntw_b == 4) {
            { static size_t hw = 0; MV_TRY(mv_lds_limit(k_chain_bwd2_x3<2, 4, 8>, lds, hw)); }
            hipLaunchKernelGGL((k_chain_bwd2_x3<2, 4, 8>), grid, dim3(512), lds, s, a, b, na);
        }
        else if (mt == 2) {
            { static size_t hw = 0; MV_TRY(mv_lds_limit(k_chain_bwd2_x3<2, 1, 16, MV_X3_PD2>, lds, hw)); }
            hipLaunchKernelGGL((k_chain_bwd2_x3<2, 1, 16, MV_X3_PD2>), grid, dim3(1024), lds, s, a, b, na);
        }
        else if (ntw_b == 2) hipLaunchKernelGGL((k_chain_bwd2_x3<1, 1, 16, MV_X3_PD1>), grid, dim3(1024), lds, s, a, b, na);
        else hipLaunchKernelGGL((k_chain_bwd2_x3<1, 2, 16>), grid, dim3(1024), lds, s, a, b, na);
        return mv_check(hipGetLastError(), "mvsdf_sdf_backward_pair (x3 chains)");
    }
    ChainArgs a, b;
    fill_chain_args(a, net, net, netT, S, lo, sdf_bwd_layout(net, MbA), ctx, 0, MbA, dyA, dnA, wsA, nullptr, w8);
    fill_chain_args(b, net, net, netT, S, lo, sdf_bwd_layout(net, MbX), ctx, row0X, MbX, dyX, dnX, wsX, dx, w8);
    a.cnt = b.cnt = cnt; a.cnt_base = row0X; b.cnt_base = 0;
    const bool w8w = mv_chain_w8();
    const int mt = (ntw_b == 2 && !w8w) ? mv_chain_mt(tiles_hint) : 1;
    const size_t lds = (size_t)16 * mt * (S + lo.d0) * sizeof(float);
    const int na = (MbA + 16 * mt - 1) / (16 * mt), nb = (MbX + 16 * mt - 1) / (16 * mt);
    const dim3 grid(na + nb);
    if (mt == 2) {
        { static size_t hw = 0; MV_TRY(mv_lds_limit(k_chain_bwd2<2, 1, 16>, lds, hw)); }
        hipLaunchKernelGGL((k_chain_bwd2<2, 1, 16>), grid, dim3(1024), lds, s, a, b, na);
    }
    else if (ntw_b == 2 && !w8w) hipLaunchKernelGGL((k_chain_bwd2<1, 1, 16>), grid, dim3(1024), lds, s, a, b, na);
    else if (ntw_b == 2) hipLaunchKernelGGL((k_chain_bwd2<1, 2, 8>), grid, dim3(512), lds, s, a, b, na);
    else if (!w8w) hipLaunchKernelGGL((k_chain_bwd2<1, 2, 16>), grid, dim3(1024), lds, s, a, b, na);
    else hipLaunchKernelGGL((k_chain_bwd2<1, 4, 8>), grid, dim3(512), lds, s, a, b, na);
    return mv_check(hipGetLastError(), "mvsdf_sdf_backward_pair");
}

// delta pass of the training step's SDF backward: extra upstream fbar[MbD] on output column 0 of rows [row0D, row0D + MbD); its first-order zbar_l
// are ADDED to the adjoints pass A stored in `ws` (linearity)
// The delta pass without a chain.  Its upstream is ONE scalar per row on output column 0, so by linearity every adjoint it produces is that scalar
// times the adjoint for the upstream e_0 -- and zbar_l for e_0 is exactly s_l = sigma'_l . u_{l+1}, the tensor the forward's normal chain already
// computed and saved for the weight gradients (k_chain_fwd: Sg[l]).  Hence zbar_l += fbar . s_l: one elementwise launch (~10 us) instead of a
// 9-phase dependent chain of GEMMs over the hit rows (92 us at c2, 100-136 us at the c5 share, on the critical path of the backward).
struct DeltaArgs {
    int nl1, rows;                           // layers 0 .. L-2; hit rows
    int N[MV_MAXL];
    const float* Sg[MV_MAXL]; float* ZB[MV_MAXL];     // already offset to the first hit row
    const float* fbar;
    // FB (the step driver): fbar is computed here instead of by k_step_bwd_fbar (same operations in the same order, sample_network.py:10-20
    // backward: fbar_i = -(xbar_i . v_i) / (n_i . v_i), xbar = d_diff + dp + dx); the blocks of layer 0 write fbar_out[i] and add it to dy[i][0]
    const float* d_diff; const float* din; const float* dx; const float* view_sorted; const float* n_hit;   // n_hit: normals of the hit rows
    int din_ld, use_geo, Nout;
    float* dy_hit; float* fbar_out;          // dy already offset to the first hit row
    const long long* cnt;                    // deferred step: the hit rows are cnt[0] (the grid covers the upper bound `rows`)
};
template <bool FB>
__global__ __launch_bounds__(256) void k_delta_apply(DeltaArgs a) {
    const int row = blockIdx.x;
    if (a.cnt && row >= (int)a.cnt[0]) return;
    if (FB) {                                                     // one workgroup per row, all layers: fbar once, then 8 independent streams per thread
        float dot = 0.f, num = 0.f;
        for (int c = 0; c < 3; ++c) {
            float xb = (a.d_diff ? a.d_diff[3 * (size_t)row + c] : 0.f);
            if (a.din && a.use_geo) xb += a.din[(size_t)row * a.din_ld + c];
            if (a.dx) xb += a.dx[3 * (size_t)row + c];
            const float v = -a.view_sorted[3 * (size_t)row + c];
            num += xb * v; dot += a.n_hit[3 * (size_t)row + c] * v;
        }
        const float f = -num / dot;
        if (threadIdx.x == 0) { a.fbar_out[row] = f; a.dy_hit[(size_t)row * a.Nout] += f; }
        for (int l = 0; l < a.nl1; ++l) {
            const int N = a.N[l];
            const float* sg = a.Sg[l] + (size_t)row * N;
            float* zb = a.ZB[l] + (size_t)row * N;
            for (int c = threadIdx.x; c < N; c += 256) zb[c] = zb[c] + f * sg[c];
        }
        return;
    }
    const int l = blockIdx.y, N = a.N[l];
    const float f = a.fbar[row];
    const float* sg = a.Sg[l] + (size_t)row * N;
    float* zb = a.ZB[l] + (size_t)row * N;
    for (int c = threadIdx.x; c < N; c += 256) zb[c] = zb[c] + f * sg[c];
}
static int mv_delta_chain() {
    static int v = -1;
    if (v < 0) { const char* e = mv_dev_env("MVSDF_DELTA_CHAIN"); v = e ? atoi(e) : 0; }
    return v;
}
int mv_delta_is_chain() { return mv_delta_chain(); }

static int sdf_delta_pass(const MvNet& net, const MvNet& netT, const SdfLayout& lo, const SdfBwdLayout& bl, const float* ctx, float* ws, int row0D,
                          int MbD, const float* fbar, hipStream_t s) {
    if (!mv_delta_chain()) {
        DeltaArgs d;
        memset(&d, 0, sizeof(d));
        d.nl1 = lo.nl - 1; d.rows = MbD; d.fbar = fbar;
        for (int l = 0; l < lo.nl - 1; ++l) {
            d.N[l] = net.L[l].N;
            d.Sg[l] = ctx + lo.Sg[l] + (size_t)row0D * net.L[l].N;
            d.ZB[l] = ws + bl.ZB[l] + (size_t)row0D * net.L[l].N;
        }
        hipLaunchKernelGGL(k_delta_apply<false>, dim3(MbD, lo.nl - 1), dim3(256), 0, s, d);
        return mv_check(hipGetLastError(), "sdf_delta_pass");
    }
    // MVSDF_DELTA_CHAIN=1: the first-order chain (the form this replaced; kept as the cross-check of tests/test_gpu_diff.py)
    const int ntw_b = mv_chain_ntw(net);
    const int nl = lo.nl, S = stride_for(net, netT);
    ChainArgs c;
    memset(&c, 0, sizeof(c));
    const size_t r0 = (size_t)row0D;
    c.net = net; c.netT = netT; c.S = S; c.M = MbD; c.row_ld0 = lo.ld0;
    c.ld_dy = net.L[nl - 1].N; c.dy_col0 = fbar; c.accum = 1;
    for (int l = 0; l < nl - 1; ++l) { c.Z[l] = ctx + lo.Z[l] + r0 * net.L[l].N; c.ZB[l] = ws + bl.ZB[l] + r0 * net.L[l].N; }
    const bool w8w = mv_chain_w8();
    const int mt = (ntw_b == 2 && !w8w) ? mv_chain_mt((MbD + 15) / 16) : 1;
    const size_t lds = (size_t)16 * mt * (S + lo.d0) * sizeof(float);
    const dim3 grid((MbD + 16 * mt - 1) / (16 * mt));
    if (mt == 2) {
        { static size_t hw = 0; MV_TRY(mv_lds_limit(k_chain_bwd<2, 1, 16>, lds, hw)); }
        hipLaunchKernelGGL((k_chain_bwd<2, 1, 16>), grid, dim3(1024), lds, s, c);
    }
    else if (ntw_b == 2 && !w8w) hipLaunchKernelGGL((k_chain_bwd<1, 1, 16>), grid, dim3(1024), lds, s, c);
    else if (ntw_b == 2) hipLaunchKernelGGL((k_chain_bwd<1, 2, 8>), grid, dim3(512), lds, s, c);
    else if (!w8w) hipLaunchKernelGGL((k_chain_bwd<1, 2, 16>), grid, dim3(1024), lds, s, c);
    else hipLaunchKernelGGL((k_chain_bwd<1, 4, 8>), grid, dim3(512), lds, s, c);
    return mv_check(hipGetLastError(), "sdf_delta_pass");
}

// operands of the SDF net's weight gradients (layers a.L[l0 ...]) from the adjoints pass A (+ delta) left in `ws`
static void sdf_wgrad_operands(WgradNetArgs& wa, int l0, const MvNet& net, const SdfLayout& lo, const SdfBwdLayout& bl, const float* dy, const float* ctx,
                               float* ws) {
    const int nl = lo.nl;
    for (int l = 0; l < nl; ++l) {
        WgradLayer& L = wa.L[l0 + l];
        const bool last = (l == nl - 1);
        L.No = net.L[l].N; L.Ki = net.L[l].K;
        L.P1 = last ? dy : ws + bl.ZB[l]; L.ldp1 = L.No;
        L.Q1 = l == 0 ? ctx + lo.H0 : ctx + lo.A[l]; L.ldq1 = l == 0 ? lo.ld0 : net.L[l].K;
        if (!last) { L.P2 = ctx + lo.Sg[l]; L.ldp2 = L.No; L.Q2 = ws + bl.VB[l]; L.ldq2 = L.ldq1; }
    }
}

extern "C" {

/* Completes pass A of mvsdf_sdf_backward_pair: (1) delta pass over rows [row0D, row0D + MbD) whose extra upstream is ONE scalar per row on
 * output column 0 (fbar[MbD]: SampleNetwork's term, known only after pass X): by linearity its zbar_l are ADDED to the stored ones (first-order
 * descending chain only); dy[(row0D + i) * Nout] must already include fbar[i] (it feeds the last layer's weight gradient).  (2) weight / bias
 * gradients of every layer from the stored adjoints -> dW_cat, db_cat.  MbD = 0 skips (1). */
int mvsdf_sdf_backward_finish(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int M, int Mg, int Mb, const float* dy, const float* ctx, float* ws,
                              int row0D, int MbD, const float* fbar, float* dW_cat, float* db_cat, void* stream) {
    MvNet net, netT;
    int rc = mv_make_net(d, &net);
    if (rc) return rc;
    rc = mv_make_net_mode(dT, &netT, 2);
    if (rc) return rc;
    if (!dy || !ctx || !ws || !dW_cat || !db_cat || Mb <= 0 || Mb > Mg || Mg > M || MbD < 0 || row0D < 0 || row0D + MbD > Mb || (MbD > 0 && !fbar))
        return mv_fail(-1, "mvsdf_sdf_backward_finish: bad arguments");
    if (!mv_chain_ntw(net)) return mv_fail(-3, "mvsdf_sdf_backward_finish: network too wide for the fused chain kernels");
    hipStream_t s = (hipStream_t)stream;
    const SdfLayout lo = sdf_ctx_layout(net, M, Mg);
    const SdfBwdLayout bl = sdf_bwd_layout(net, Mb);
    const int nl = lo.nl;
    if (MbD > 0) {
        rc = sdf_delta_pass(net, netT, lo, bl, ctx, ws, row0D, MbD, fbar, s);
        if (rc) return rc;
    }
    WgradNetArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.n_layers = nl; wa.chunk = bl.chunk;
    sdf_wgrad_operands(wa, 0, net, lo, bl, dy, ctx, ws);
    wgrad_net_layers(wa, 0, nl, Mb, bl.nchunks, ws + bl.slabA, ws + bl.bslab, dW_cat, db_cat);
    const int Ki = net.L[nl - 1].K;                                                // W_last[0, :] += sum_rows ubar_last   (E.1 end)
    wgrad_colsum(wa, ws + bl.VB[nl - 1], Ki, Mb, Ki, nl - 1, bl.nchunks, ws + bl.slabB);
    MV_TRY(launch_wgrad_net(wa, s));
    return mv_check(hipGetLastError(), "mvsdf_sdf_backward_finish");
}

}  // extern "C"

// ================================================================================================ rendering network
struct RenderLayout { size_t A[MV_MAXL], rgb, total; };
static RenderLayout render_layout(const MvNet& net, int N) {
    RenderLayout o;
    memset(&o, 0, sizeof(o));
    size_t p = 0;
    for (int l = 0; l < net.n_layers; ++l) { o.A[l] = p; p += (size_t)N * net.L[l].K; }
    o.rgb = p; p += (size_t)N * net.L[net.n_layers - 1].N;
    o.total = p;
    return o;
}
struct RenderBwdLayout { size_t ZB[MV_MAXL], slab, bslab, total; int chunk, nchunks; };
static RenderBwdLayout render_bwd_layout(const MvNet& net, int N) {
    RenderBwdLayout o;
    memset(&o, 0, sizeof(o));
    size_t p = 0, maxnk = 0;
    int maxw = 0;
    for (int l = 0; l < net.n_layers; ++l) {
        o.ZB[l] = p; p += (size_t)N * net.L[l].N;
        const size_t nk = (size_t)net.L[l].N * net.L[l].K;
        maxnk = nk > maxnk ? nk : maxnk;
        maxw = net.L[l].N > maxw ? net.L[l].N : maxw;
    }
    o.chunk = MV_WG_CHUNK; o.nchunks = (N + MV_WG_CHUNK - 1) / MV_WG_CHUNK; if (o.nchunks < 1) o.nchunks = 1;
    o.slab = p; p += wgrad_net_slab_floats(net, o.nchunks);
    o.bslab = p; p += wgrad_net_bslab_floats(net, o.nchunks);
    o.total = p;
    return o;
}

// A0[row] = cat[points(3), view(3), sin/cos(2^m view) m<mv, normals(3), feat(F)]   (idr.py:146-154; view / normal parts per the mode bits of mv)
__global__ void k_render_input(const float* __restrict__ points, const float* __restrict__ view, const float* __restrict__ normals,
                               const float* __restrict__ feat, int ldfeat, int N, int mv, int K0, float* __restrict__ A0) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * K0) return;
    const int row = (int)(idx / K0), k = (int)(idx - (size_t)row * K0);
    const float v = mv_render_input_raw(points, view, normals, feat, ldfeat, mv, row, k);
    A0[idx] = v;
}

extern "C" {

size_t mvsdf_render_ctx_floats(const MvsdfNetDesc* d, int N) {
    MvNet net;
    if (mv_make_net_mode(d, &net, 1)) return 0;
    return render_layout(net, N).total;
}
size_t mvsdf_render_bwd_ws_floats(const MvsdfNetDesc* d, int N) {
    MvNet net;
    if (mv_make_net_mode(d, &net, 1)) return 0;
    return render_bwd_layout(net, N).total;
}

/* RenderingNetwork.forward, mode 'idr' (idr.py:145-167): rgb = tanh(MLP(cat[points, PE(view), normals, feat])). */
int mvsdf_render_forward(const MvsdfNetDesc* d, const float* points, const float* view, const float* normals, const float* feat,
                         int ldfeat, int N, int multires_view, float* rgb, float* ctx, void* stream) {
    MvNet net;
    int rc = mv_make_net_mode(d, &net, 1);
    if (rc) return rc;
    if (!points || !view || !normals || !feat || !rgb || !ctx || N <= 0) return mv_fail(-1, "mvsdf_render_forward: bad arguments");
    const int nl = net.n_layers, K0 = net.L[0].K;
    if ((multires_view & ~0x3ff) || (multires_view & 0xff) > 16 || (multires_view & 0x300) == 0x300)
        return mv_fail(-1, "mvsdf_render_forward: bad multires_view / mode bits");
    if (K0 <= 3 + mv_render_dv(multires_view) + mv_render_dn(multires_view))
        return mv_fail(-1, "mvsdf_render_forward: first layer too narrow for cat[points, PE(view), normals, feat]");
    hipStream_t s = (hipStream_t)stream;
    const RenderLayout lo = render_layout(net, N);
    static int fuse_r = -1;
    if (fuse_r < 0) { const char* e = mv_dev_env("MVSDF_FUSE"); fuse_r = e ? atoi(e) : 1; }
    const int ntw_r = mv_chain_ntw(net);
    if (fuse_r && ntw_r && net.L[nl - 1].NT <= 2) {                                // the whole network in one launch per row tile
        RenderChainArgs c;
        memset(&c, 0, sizeof(c));
        c.net = net; c.S = net.S; c.N = N; c.mv = multires_view; c.K0 = K0;
        c.points = points; c.view = view; c.normals = normals; c.feat = feat; c.ldfeat = ldfeat;
        for (int l = 0; l < nl; ++l) c.A[l] = ctx + lo.A[l];
        c.rgb_ctx = ctx + lo.rgb; c.rgb = rgb;
        constexpr int MTC = 1, NWC = 8;
        const bool w8 = mv_chain_w8();
        const dim3 grid((N + 16 * MTC - 1) / (16 * MTC));
        const size_t ldsr = (size_t)16 * MTC * net.S * sizeof(float);
        if (ntw_r == 2 && !w8) hipLaunchKernelGGL((k_render_chain_fwd<MTC, 1, 16>), grid, dim3(1024), ldsr, s, c);
        else if (ntw_r == 2) hipLaunchKernelGGL((k_render_chain_fwd<MTC, 2, NWC>), grid, dim3(64 * NWC), ldsr, s, c);
        else if (!w8) hipLaunchKernelGGL((k_render_chain_fwd<MTC, 2, 16>), grid, dim3(1024), ldsr, s, c);
        else hipLaunchKernelGGL((k_render_chain_fwd<MTC, 4, NWC>), grid, dim3(64 * NWC), ldsr, s, c);
        return mv_check(hipGetLastError(), "mvsdf_render_forward");
    }
    const size_t tot = (size_t)N * K0;
    hipLaunchKernelGGL(k_render_input, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, points, view, normals, feat, ldfeat, N,
                       multires_view, K0, ctx + lo.A[0]);
    for (int l = 0; l < nl; ++l) {
        LayerArgs a = base_args(net.L[l], net.S, N);
        const bool last = (l == nl - 1);
        a.A = ctx + lo.A[l]; a.lda = net.L[l].K;
        a.out0 = last ? ctx + lo.rgb : ctx + lo.A[l + 1]; a.ld0 = net.L[l].N;
        if (last) MV_TRY((launch_layer<PRO_PLAIN, EPI_TANH>(a, s))); else MV_TRY((launch_layer<PRO_PLAIN, EPI_RELU>(a, s)));
    }
    MV_TRY(hipMemcpyAsync(rgb, ctx + lo.rgb, (size_t)N * net.L[nl - 1].N * sizeof(float), hipMemcpyDeviceToDevice, s));
    return mv_check(hipGetLastError(), "mvsdf_render_forward");
}

}  // extern "C"

// the descending chain of the rendering net's backward: drgb[N][3] -> per-layer adjoints in `ws` + din[N][K0]
static int render_backward_chain(const MvNet& net, const MvNet& netT, int N, int Nctx, const float* drgb, const float* ctx, float* din, float* ws,
                                 hipStream_t s, const long long* drgb_rows = nullptr, const long long* cnt = nullptr) {
    const int nl = net.n_layers, S = stride_for(net, netT);
    const RenderLayout lo = render_layout(net, Nctx);        // the forward context holds Nctx rows; the backward covers the first N
    const RenderBwdLayout bl = render_bwd_layout(net, N);
    static int fuse_rb = -1;
    if (fuse_rb < 0) { const char* e = mv_dev_env("MVSDF_FUSE"); fuse_rb = e ? atoi(e) : 1; }
    const int ntw_rb = mv_chain_ntw(net);
    bool fused_bwd = fuse_rb && ntw_rb;
    for (int l = 1; l < nl; ++l) fused_bwd = fused_bwd && netT.L[l].NT <= 8 * ntw_rb;   // one column-tile group per wave above the first layer
    if (fused_bwd) {
        RenderChainArgs c;
        memset(&c, 0, sizeof(c));
        c.net = net; c.netT = netT; c.S = S; c.N = N; c.K0 = net.L[0].K;
        c.drgb = drgb; c.rgbc = ctx + lo.rgb; c.din = din; c.drgb_rows = drgb_rows; c.cnt = cnt;
        for (int l = 0; l < nl; ++l) { c.Ac[l] = ctx + lo.A[l]; c.ZB[l] = ws + bl.ZB[l]; }
        constexpr int MTC = 1, NWC = 8;
        const bool w8 = mv_chain_w8();
        const dim3 grid((N + 16 * MTC - 1) / (16 * MTC));
        const size_t ldsr = (size_t)16 * MTC * S * sizeof(float);
        if (ntw_rb == 2 && !w8) hipLaunchKernelGGL((k_render_chain_bwd<MTC, 1, 16>), grid, dim3(1024), ldsr, s, c);
        else if (ntw_rb == 2) hipLaunchKernelGGL((k_render_chain_bwd<MTC, 2, NWC>), grid, dim3(64 * NWC), ldsr, s, c);
        else if (!w8) hipLaunchKernelGGL((k_render_chain_bwd<MTC, 2, 16>), grid, dim3(1024), ldsr, s, c);
        else hipLaunchKernelGGL((k_render_chain_bwd<MTC, 4, NWC>), grid, dim3(64 * NWC), ldsr, s, c);
        MV_TRY(hipGetLastError());
    } else {
    if (drgb_rows || cnt) return mv_fail(-3, "render_backward_chain: row indirection / device-side counts need the fused chain kernel");
    for (int l = nl - 1; l >= 0; --l) {                      // abar_l = zbar_l W_l ; zbar_{l-1} = abar_l . relu'(z_{l-1})
        LayerArgs a = base_args(netT.L[l], S, N);
        const bool last = (l == nl - 1);
        if (last) { a.A = drgb; a.lda = net.L[l].N; a.U = ctx + lo.rgb; a.ldu = net.L[l].N; a.out2 = ws + bl.ZB[l]; a.ld2 = net.L[l].N; }
        else { a.A = ws + bl.ZB[l]; a.lda = net.L[l].N; }
        if (l > 0) {
            a.add = ctx + lo.A[l]; a.ldadd = net.L[l].K;     // relu mask: stored post-activation > 0
            a.out0 = ws + bl.ZB[l - 1]; a.ld0 = net.L[l - 1].N;
            if (last) MV_TRY((launch_layer<PRO_TANH_BWD, EPI_RELU_MASK>(a, s))); else MV_TRY((launch_layer<PRO_PLAIN, EPI_RELU_MASK>(a, s)));
        } else {
            a.csplit = net.L[0].K; a.out0 = din; a.ld0 = net.L[0].K;
            if (last) MV_TRY((launch_layer<PRO_TANH_BWD, EPI_SPLIT>(a, s))); else MV_TRY((launch_layer<PRO_PLAIN, EPI_SPLIT>(a, s)));
        }
    }
    }
    return 0;
}
static void render_wgrad_operands(WgradNetArgs& wa, int l0, const MvNet& net, const RenderLayout& lo, const RenderBwdLayout& bl, const float* ctx, float* ws) {
    for (int l = 0; l < net.n_layers; ++l) {
        WgradLayer& L = wa.L[l0 + l];
        L.No = net.L[l].N; L.Ki = net.L[l].K;
        L.P1 = ws + bl.ZB[l]; L.ldp1 = L.No;
        L.Q1 = ctx + lo.A[l]; L.ldq1 = L.Ki;
    }
}

extern "C" {

/* Backward: drgb[N][3] -> dW_cat, db_cat, din[N][K0] (adjoint of the concatenated input; the caller slices
 * points = [:, 0:3], normals = [:, 3+dv : 6+dv], feat = [:, 6+dv :]). */
int mvsdf_render_backward(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int N, int Nctx, const float* drgb, const float* ctx, float* dW_cat,
                          float* db_cat, float* din, float* ws, void* stream) {
    MvNet net, netT;
    int rc = mv_make_net_mode(d, &net, 1);
    if (rc) return rc;
    rc = mv_make_net_mode(dT, &netT, 2);
    if (rc) return rc;
    if (!drgb || !ctx || !dW_cat || !db_cat || !din || !ws || N <= 0 || Nctx < N) return mv_fail(-1, "mvsdf_render_backward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    rc = render_backward_chain(net, netT, N, Nctx, drgb, ctx, din, ws, s);
    if (rc) return rc;
    const RenderLayout lo = render_layout(net, Nctx);
    const RenderBwdLayout bl = render_bwd_layout(net, N);
    WgradNetArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.n_layers = net.n_layers; wa.chunk = bl.chunk;
    render_wgrad_operands(wa, 0, net, lo, bl, ctx, ws);
    wgrad_net_layers(wa, 0, net.n_layers, N, bl.nchunks, ws + bl.slab, ws + bl.bslab, dW_cat, db_cat);
    MV_TRY(launch_wgrad_net(wa, s));
    return mv_check(hipGetLastError(), "mvsdf_render_backward");
}

}  // extern "C"

// ================================================================================================ step driver pieces (step_internal.h)
// The backward of a training step in pieces: rendering-net chain, delta, and the weight gradients of BOTH networks in one k_wgrad_net / k_reduce_net pair.
int mv_render_backward_chain(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int N, int Nctx, const float* drgb, const long long* drgb_rows, const float* ctx,
                             float* din, float* ws, const long long* cnt, void* stream) {
    MvNet net, netT;
    int rc = mv_make_net_mode(d, &net, 1);
    if (rc) return rc;
    rc = mv_make_net_mode(dT, &netT, 2);
    if (rc) return rc;
    if (!drgb || !ctx || !din || !ws || N <= 0 || Nctx < N) return mv_fail(-1, "mv_render_backward_chain: bad arguments");
    rc = render_backward_chain(net, netT, N, Nctx, drgb, ctx, din, ws, (hipStream_t)stream, drgb_rows, cnt);
    return rc ? rc : mv_check(hipGetLastError(), "mv_render_backward_chain");
}

int mv_sdf_backward_delta(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int M, int Mg, int Mb, const float* ctx, float* ws, int row0D, int MbD,
                          const float* fbar, void* stream) {
    MvNet net, netT;
    int rc = mv_make_net(d, &net);
    if (rc) return rc;
    rc = mv_make_net_mode(dT, &netT, 2);
    if (rc) return rc;
    if (!ctx || !ws || !fbar || Mb <= 0 || Mb > Mg || Mg > M || MbD <= 0 || row0D < 0 || row0D + MbD > Mb) return mv_fail(-1, "mv_sdf_backward_delta: bad arguments");
    if (!mv_chain_ntw(net)) return mv_fail(-3, "mv_sdf_backward_delta: network too wide for the fused chain kernels");
    return sdf_delta_pass(net, netT, sdf_ctx_layout(net, M, Mg), sdf_bwd_layout(net, Mb), ctx, ws, row0D, MbD, fbar, (hipStream_t)stream);
}

// mvsdf_step_backward_fbar + mv_sdf_backward_delta in one launch (the step driver's route; identical results): the hit rows are rows
// [row0D, row0D + MbD) of the evaluation, dy / n_eval are the evaluation's [M][Nout] / [M][3]
int mv_sdf_backward_delta_fbar(const MvsdfNetDesc* d, int M, int Mg, int Mb, const float* ctx, float* ws, int row0D, int MbD, int Nout, const float* din,
                               int din_ld, int use_geo, const float* d_diff, const float* dx, const float* view_sorted, const float* n_eval, float* dy,
                               float* fbar, const long long* cnt, void* stream) {
    MvNet net;
    int rc = mv_make_net(d, &net);
    if (rc) return rc;
    if (!ctx || !ws || !fbar || !view_sorted || !n_eval || !dy || Mb <= 0 || Mb > Mg || Mg > M || MbD <= 0 || row0D < 0 || row0D + MbD > Mb)
        return mv_fail(-1, "mv_sdf_backward_delta_fbar: bad arguments");
    const SdfLayout lo = sdf_ctx_layout(net, M, Mg);
    const SdfBwdLayout bl = sdf_bwd_layout(net, Mb);
    DeltaArgs a;
    memset(&a, 0, sizeof(a));
    a.nl1 = lo.nl - 1; a.rows = MbD;
    for (int l = 0; l < lo.nl - 1; ++l) {
        a.N[l] = net.L[l].N;
        a.Sg[l] = ctx + lo.Sg[l] + (size_t)row0D * net.L[l].N;
        a.ZB[l] = ws + bl.ZB[l] + (size_t)row0D * net.L[l].N;
    }
    a.d_diff = d_diff; a.din = din; a.dx = dx; a.view_sorted = view_sorted; a.n_hit = n_eval + 3 * (size_t)row0D;
    a.din_ld = din_ld; a.use_geo = use_geo; a.Nout = Nout; a.dy_hit = dy + (size_t)row0D * Nout; a.fbar_out = fbar; a.cnt = cnt;
    hipLaunchKernelGGL(k_delta_apply<true>, dim3(MbD), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mv_sdf_backward_delta_fbar");
}

/* Weight / bias gradients of BOTH networks of a training step: one k_wgrad_net launch over every layer and chunk (+ the column sums of ubar_last) and
 * ONE reduction of all slabs -> dW_s / db_s / dW_r / db_r.  The rendering net takes part when N > 0 and rctx / rws are given; otherwise the caller
 * zero-fills its targets. */
int mv_step_wgrad(const MvsdfNetDesc* sd, const MvsdfNetDesc* rd, int M, int Mg, int Mb, const float* dy, const float* ctx, float* wsA, int N, int Nctx,
                  const float* rctx, float* rws, float* dW_s, float* db_s, float* dW_r, float* db_r, const long long* cnt, int cnt_base, void* stream) {
    MvNet net, rnet;
    int rc = mv_make_net(sd, &net);
    if (rc) return rc;
    const bool with_r = N > 0 && rctx && rws;
    if (with_r) { rc = mv_make_net_mode(rd, &rnet, 1); if (rc) return rc; }
    if (!dy || !ctx || !wsA || !dW_s || !db_s || Mb <= 0 || Mb > Mg || Mg > M || (with_r && (!dW_r || !db_r || Nctx < N)))
        return mv_fail(-1, "mv_step_wgrad: bad arguments");
    const SdfLayout lo = sdf_ctx_layout(net, M, Mg);
    const SdfBwdLayout bl = sdf_bwd_layout(net, Mb);
    const int nl = lo.nl, nr = with_r ? rnet.n_layers : 0;
    if (nl + nr > MV_WG_MAXL) return mv_fail(-1, "mv_step_wgrad: too many layers");
    WgradNetArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.n_layers = nl + nr; wa.chunk = bl.chunk;
    sdf_wgrad_operands(wa, 0, net, lo, bl, dy, ctx, wsA);
    wgrad_net_layers(wa, 0, nl, Mb, bl.nchunks, wsA + bl.slabA, wsA + bl.bslab, dW_s, db_s);
    if (with_r) {
        const RenderLayout rlo = render_layout(rnet, Nctx);
        const RenderBwdLayout rbl = render_bwd_layout(rnet, N);
        if (rbl.chunk != bl.chunk) return mv_fail(-1, "mv_step_wgrad: chunk sizes differ");
        render_wgrad_operands(wa, nl, rnet, rlo, rbl, rctx, rws);
        wgrad_net_layers(wa, nl, nr, N, rbl.nchunks, rws + rbl.slab, rws + rbl.bslab, dW_r, db_r);
    }
    const int Ki = net.L[nl - 1].K;
    wgrad_colsum(wa, wsA + bl.VB[nl - 1], Ki, Mb, Ki, nl - 1, bl.nchunks, wsA + bl.slabB);
    if (cnt) {                                                  // deferred step: SDF layers cover cnt_base + cnt[0] rows, rendering layers cnt[0]
        wa.cnt = cnt; wa.col_mbase = cnt_base;
        for (int l = 0; l < nl; ++l) wa.L[l].mbase = cnt_base;
        for (int l = nl; l < nl + nr; ++l) wa.L[l].mbase = 0;
    }
    MV_TRY(launch_wgrad_net(wa, (hipStream_t)stream));
    return mv_check(hipGetLastError(), "mv_step_wgrad");
}

// Can mvsdf_step_backward run these networks with device-side counts?  Every launch of that route must be one of the fused forms that take them: the SDF
// chains (any arithmetic), the elementwise delta, the rendering net's fused descending chain, k_wgrad_net.  (The per-layer fall-backs size their grids from host
// numbers: a step on such a network waits for the counts as before.)
int mv_step_can_defer(const MvsdfNetDesc* sdf, const MvsdfNetDesc* sdfT, const MvsdfNetDesc* rnd, const MvsdfNetDesc* rndT) {
    MvNet net, netT, rnet, rnetT;
    if (mv_make_net(sdf, &net) || mv_make_net_mode(sdfT, &netT, 2) || mv_make_net_mode(rnd, &rnet, 1) || mv_make_net_mode(rndT, &rnetT, 2)) return 0;
    if (!mv_chain_ntw(net) || mv_delta_chain()) return 0;
    const char* e = mv_dev_env("MVSDF_FUSE");
    if (e && atoi(e) == 0) return 0;
    const int ntw_rb = mv_chain_ntw(rnet);
    if (!ntw_rb) return 0;
    for (int l = 1; l < rnet.n_layers; ++l) if (rnetT.L[l].NT > 8 * ntw_rb) return 0;
    return net.n_layers + rnet.n_layers <= MV_WG_MAXL ? 1 : 0;
}

#ifdef MV_CHAIN_PROBE
// dev probe (tools/chain_probe.py): per-wave clock ticks (100 MHz) between the marks of workgroup 0 of k_chain_fwd, summed over the launches since the last reset
extern "C" int mv_chain_probe_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ph), sizeof(unsigned long long) * 256) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[256]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_chain_ph), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

