// basic.hip -- weight-norm fold + MFMA packing, camera rays, the stand-alone tracing-MLP kernel and the
// device self-test of det_math.  C ABI entry points for these live at the bottom (see include/mvsdf_hip.h).
#include <stdlib.h>
#include <string.h>
#include "tile_engine.h"
#include "trace_params.h"
#include "capi_util.h"
#include "det_math_pk.h"

// ---- weight norm (idr.py:70-71): one wave per output row.  The row is loaded coalesced; lane 0 then walks it through
// v_readlane in ascending k -- the SAME k-ascending fmaf chain as the CPU restatement (bit-exact), at ~1 us per row. ----
__global__ __launch_bounds__(256) void k_fold(const float* __restrict__ v, const float* __restrict__ g, int N, int K, float* __restrict__ w) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const float* vr = v + (size_t)j * K;
    float ss = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const float mine = (k0 + lane < K) ? vr[k0 + lane] : 0.0f;
        const int n = min(64, K - k0);
        for (int i = 0; i < n; ++i) {
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), i));   // uniform i -> v_readlane
            ss = fmaf(x, x, ss);
        }
    }
    const float a = g[j] / sqrtf(ss);
    for (int k = lane; k < K; k += 64) w[(size_t)j * K + k] = vr[k] * a;
}

// wp[ct][kb][lane][s] = W[ct*16 + (lane&15)][kb*16 + 4s + (lane>>4)];  transposed=1 packs W^T ([K][N] seen as out=K, in=N)
__global__ void k_pack(const float* __restrict__ w, int N, int K, int transposed, float* __restrict__ wp) {
    const int No = transposed ? K : N, Ko = transposed ? N : K;
    const int KB = mv_kpad(Ko) / 16;
    const size_t total = (size_t)mv_ceil16(No) * mv_kpad(Ko);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int s = idx & 3, lane = (idx >> 2) & 63;
        const size_t blk = idx >> 8;
        const int kb = (int)(blk % KB), ct = (int)(blk / KB);
        const int o = ct * 16 + (lane & 15), i = kb * 16 + 4 * s + (lane >> 4);
        float val = 0.0f;
        if (o < No && i < Ko) val = transposed ? w[(size_t)i * K + o] : w[(size_t)o * K + i];
        wp[idx] = val;
    }
}

// ---- whole-network variants: one launch folds / packs every layer (blockIdx.y selects the layer); the layer list may span several
// networks (SDF + rendering net of a step: 9 + 5 layers) ----
#define MV_FOLD_MAXL 24
struct FoldNetArgs {
    int n_layers;
    const float* v[MV_FOLD_MAXL]; const float* g[MV_FOLD_MAXL]; const float* dW[MV_FOLD_MAXL];
    float* w[MV_FOLD_MAXL]; float* wp[MV_FOLD_MAXL]; float* wpT[MV_FOLD_MAXL]; float* dv[MV_FOLD_MAXL]; float* dg[MV_FOLD_MAXL];
    const float* db[MV_FOLD_MAXL]; float* dbias[MV_FOLD_MAXL];      // backward only: bias gradients routed to their sink (optional)
    int N[MV_FOLD_MAXL], K[MV_FOLD_MAXL];
    int accumulate;                                       // backward only: add into dv / dg / dbias instead of overwriting
};

__global__ __launch_bounds__(256) void k_fold_net(FoldNetArgs a) {
    const int l = blockIdx.y, N = a.N[l], K = a.K[l];
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const float* vr = a.v[l] + (size_t)j * K;
    if (!a.g[l]) {                                              // no weight norm for this layer (weight_norm=False): w = v
        float* wr0 = a.w[l] + (size_t)j * K;
        for (int k = lane; k < K; k += 64) wr0[k] = vr[k];
        return;
    }
    float ss = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const float mine = (k0 + lane < K) ? vr[k0 + lane] : 0.0f;
        const int n = min(64, K - k0);
        for (int i = 0; i < n; ++i) {
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), i));
            ss = fmaf(x, x, ss);
        }
    }
    const float s = a.g[l][j] / sqrtf(ss);
    float* wr = a.w[l] + (size_t)j * K;
    for (int k = lane; k < K; k += 64) wr[k] = vr[k] * s;
}

__global__ void k_pack_net(FoldNetArgs a) {
    const int l = blockIdx.y >> 1, transposed = blockIdx.y & 1;
    float* wp = transposed ? a.wpT[l] : a.wp[l];
    if (!wp) return;
    const float* w = a.w[l];
    const int N = a.N[l], K = a.K[l];
    const int No = transposed ? K : N, Ko = transposed ? N : K;
    const int KB = mv_kpad(Ko) / 16;
    const size_t total = (size_t)mv_ceil16(No) * mv_kpad(Ko);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int s = idx & 3, lane = (idx >> 2) & 63;
        const size_t blk = idx >> 8;
        const int kb = (int)(blk % KB), ct = (int)(blk / KB);
        const int o = ct * 16 + (lane & 15), i = kb * 16 + 4 * s + (lane >> 4);
        float val = 0.0f;
        if (o < No && i < Ko) val = transposed ? w[(size_t)i * K + o] : w[(size_t)o * K + i];
        wp[idx] = val;
    }
}

__global__ void k_fold_bwd_net(FoldNetArgs a) {
    const int l = blockIdx.y, N = a.N[l], K = a.K[l];
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const float* vr = a.v[l] + (size_t)j * K;
    const float* dr = a.dW[l] + (size_t)j * K;
    if (!a.g[l]) {                                              // w = v: dv = dW, no dg
        float* dv0 = a.dv[l] + (size_t)j * K;
        for (int k = lane; k < K; k += 64) dv0[k] = a.accumulate ? dv0[k] + dr[k] : dr[k];
        if (lane == 0 && a.db[l]) a.dbias[l][j] = a.accumulate ? a.dbias[l][j] + a.db[l][j] : a.db[l][j];
        return;
    }
    float* dvr = a.dv[l] + (size_t)j * K;
    const bool acc = a.accumulate != 0;
    if (K <= 64 * 9) {                                          // the row's three operands once, all requested up front (clamped addresses), kept in registers
        float vv[9], dd[9], old[9];
        const float gj = a.g[l][j];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int k = 64 * c + lane, kc = k < K ? k : K - 1;
            const float x = vr[kc], y = dr[kc], z = acc ? dvr[kc] : 0.0f;
            vv[c] = k < K ? x : 0.0f; dd[c] = k < K ? y : 0.0f; old[c] = z;
        }
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int c = 0; c < 9; ++c) if (64 * c < K) { ss = fmaf(vv[c], vv[c], ss); dot = fmaf(dd[c], vv[c], dot); }
        for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
        const float inv = 1.0f / sqrtf(ss);
        const float dgj = dot * inv, s = gj * inv;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int k = 64 * c + lane;
            if (k < K) { const float t = s * (dd[c] - dgj * vv[c] * inv); dvr[k] = acc ? old[c] + t : t; }
        }
        if (lane == 0) {
            a.dg[l][j] = acc ? a.dg[l][j] + dgj : dgj;
            if (a.db[l]) a.dbias[l][j] = acc ? a.dbias[l][j] + a.db[l][j] : a.db[l][j];
        }
        return;
    }
    float ss = 0.f, dot = 0.f;
    for (int k = lane; k < K; k += 64) { ss = fmaf(vr[k], vr[k], ss); dot = fmaf(dr[k], vr[k], dot); }
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
    const float inv = 1.0f / sqrtf(ss);
    const float dgj = dot * inv, s = a.g[l][j] * inv;
    for (int k = lane; k < K; k += 64) {
        const float t = s * (dr[k] - dgj * vr[k] * inv);
        dvr[k] = acc ? dvr[k] + t : t;
    }
    if (lane == 0) {
        a.dg[l][j] = acc ? a.dg[l][j] + dgj : dgj;
        if (a.db[l]) a.dbias[l][j] = acc ? a.dbias[l][j] + a.db[l][j] : a.db[l][j];
    }
}

// backward of the fold: one wave per row.
__global__ void k_fold_bwd(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ dW, int N, int K,
                           float* __restrict__ dv, float* __restrict__ dg) {
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const float* vr = v + (size_t)j * K;
    const float* dr = dW + (size_t)j * K;
    float ss = 0.f, dot = 0.f;
    for (int k = lane; k < K; k += 64) { ss = fmaf(vr[k], vr[k], ss); dot = fmaf(dr[k], vr[k], dot); }
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
    const float nrm = sqrtf(ss), inv = 1.0f / nrm;
    const float dgj = dot * inv;                         // dW . v_hat
    const float a = g[j] * inv;
    for (int k = lane; k < K; k += 64) dv[(size_t)j * K + k] = a * (dr[k] - dgj * vr[k] * inv);
    if (lane == 0) dg[j] = dgj;
}

// ---- rend_util.get_camera_params + lift (rend_util.py:48-75, 87-100) ----
__global__ void k_camera_rays(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ Kin, int B, int P,
                              float* __restrict__ dirs, float* __restrict__ cam_loc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * P) return;
    const int b = idx / P;
    const float* p = pose + 16 * b;
    const float* k = Kin + 16 * b;
    const float fx = k[0], fy = k[5], cx = k[2], cy = k[6], sk = k[1];
    const float cl[3] = {p[3], p[7], p[11]};
    if (idx == b * P) { cam_loc[3 * b] = cl[0]; cam_loc[3 * b + 1] = cl[1]; cam_loc[3 * b + 2] = cl[2]; }
    const float x = uv[2 * (size_t)idx] + 0.5f, y = uv[2 * (size_t)idx + 1] + 0.5f, z = 1.0f;
    const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z;
    const float yl = (y - cy) / fy * z;
    const float h[4] = {xl, yl, z, 1.0f};
    float wv[3];
    for (int r = 0; r < 3; ++r) {
        float acc = 0.0f;
        for (int c = 0; c < 4; ++c) acc = fmaf(p[4 * r + c], h[c], acc);
        wv[r] = acc - cl[r];
    }
    float nn = sqrtf(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]);
    if (nn < 1e-12f) nn = 1e-12f;
    dirs[3 * (size_t)idx] = wv[0] / nn;
    dirs[3 * (size_t)idx + 1] = wv[1] / nn;
    dirs[3 * (size_t)idx + 2] = wv[2] / nn;
}

// ---- rend_util.get_sphere_intersection (rend_util.py:141-162), same op order as the tracer's fused copy ----
__global__ void k_sphere_intersection(const float* __restrict__ cam_loc, const float* __restrict__ dirs, int B, int P, float r,
                                      float* __restrict__ t, uint8_t* __restrict__ mask) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * P) return;
    const float* c = cam_loc + 3 * (idx / P);
    const float* d = dirs + 3 * (size_t)idx;
    const float dot = fmaf(d[2], c[2], fmaf(d[1], c[1], d[0] * c[0]));
    const float nrm = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float under = dot * dot - (nrm * nrm - r * r);
    const bool hit = under > 0.0f;
    float a = 0.0f, b = 0.0f;
    if (hit) { const float s = sqrtf(under); a = s * -1.0f - dot; b = s * 1.0f - dot; }
    t[2 * (size_t)idx] = a < 0.0f ? 0.0f : a;
    t[2 * (size_t)idx + 1] = b < 0.0f ? 0.0f : b;
    mask[idx] = hit ? 1 : 0;
}

// ---- the tracing MLP alone: y[i] = ImplicitNetwork(x[i])[0] ----
template <int MT, int NTW, int NW, bool XR = false>
__global__ __launch_bounds__(64 * NW) void k_sdf_col0(MvNet net, const float* __restrict__ x, int n, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x, d0 = 3 + 6 * net.multires;
    float* act = smem;
    float* pe = act + ROWS * net.S;
    float* pts = pe + ((ROWS * d0 + 3) & ~3);
    float* out = pts + ROWS * 4;
    const int row0 = blockIdx.x * ROWS;
    for (int i = tid; i < ROWS * 3; i += 64 * NW) {
        const int row = row0 + i / 3;
        pts[i] = row < n ? x[3 * (size_t)row0 + i] : 0.0f;
    }
    __syncthreads();
    mv_sdf_eval_col0<MT, NTW, NW, XR>(net, act, pe, pts, out, tid);
    if (tid < ROWS && row0 + tid < n) y[row0 + tid] = out[tid];
}

// bf16 packs (tile_engine_bf16.h): one thread per packed element
struct PackBfArgs { const float* w[MV_MAXL]; uint16_t* wp[MV_MAXL]; int N[MV_MAXL], K[MV_MAXL], nsplit[MV_MAXL]; int tr; };   // tr (k_pack_bf16x3_net): pack W^T (N, K stay W's dims)
__global__ void k_pack_bf16_net(PackBfArgs a) {
    const int l = blockIdx.y;
    const int N = a.N[l], K = a.K[l], ns = a.nsplit[l], KB = mv_bf_kb(K, ns);
    const size_t total = mv_packed_bf16_elems(N, K, ns);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = idx & 7, lane = (idx >> 3) & 63;
        const size_t blk = idx >> 9;
        const int kb = (int)(blk % KB), ct = (int)(blk / KB);
        const int o = ct * 16 + (lane & 15), kp = kb * 32 + 8 * (lane >> 4) + i;
        const int col = kp < K ? kp : (kp < K + ns ? kp - ns : -1);          // the lo copy of a split column shares the hi column's weight
        a.wp[l][idx] = (o < N && col >= 0) ? mv_f2bf(a.w[l][(size_t)o * K + col]) : (uint16_t)0;
    }
}

// trace_dtype = 5: the packs of tile_engine_bf16s.h's weight-term engine: w = t0 + t1 + t2 exactly, wp[((ct * KB + kb) * 3 + term) * 64 + lane][8]
__global__ void k_pack_bf16x3_net(PackBfArgs a) {
    const int l = blockIdx.y;
    const int N = a.tr ? a.K[l] : a.N[l], K = a.tr ? a.N[l] : a.K[l], KB = mv_bf_kb(K, 0);      // dims of the matrix being packed (W or W^T)
    const size_t total = 3 * mv_packed_bf16_elems(N, K, 0);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = idx & 7, lane = (idx >> 3) & 63;
        const size_t blk = idx >> 9;
        const int term = (int)(blk % 3), kb = (int)((blk / 3) % KB), ct = (int)(blk / 3 / KB);
        const int o = ct * 16 + (lane & 15), kp = kb * 32 + 8 * (lane >> 4) + i;
        uint16_t v = 0;
        if (o < N && kp < K) {
            float r = a.tr ? a.w[l][(size_t)kp * N + o] : a.w[l][(size_t)o * K + kp];
            if (fabsf(r) < 9.094947017729282e-13f) r = 0.0f;                            // |w| < 2^-40 -> 0 (tile_engine_bf16s.h, MV_X3_FLUSH: the oracle does the same)
            v = mv_f2bf(r);
            for (int t = 0; t < term; ++t) { r = r - mv_bf2f(v); v = mv_f2bf(r); }       // every subtraction is exact
        }
        a.wp[l][idx] = v;
    }
}

// trace_dtype = 2: fp32 packs of the bf16-rounded weights (layout of k_pack, non-transposed)
__global__ void k_pack_round_net(PackBfArgs a) {
    const int l = blockIdx.y;
    const int N = a.N[l], K = a.K[l], KB = mv_kpad(K) / 16;
    float* wp = (float*)a.wp[l];
    const size_t total = mv_packed_floats(N, K);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int s = idx & 3, lane = (idx >> 2) & 63;
        const size_t blk = idx >> 8;
        const int kb = (int)(blk % KB), ct = (int)(blk / KB);
        const int o = ct * 16 + (lane & 15), i = kb * 16 + 4 * s + (lane >> 4);
        wp[idx] = (o < N && i < K) ? mv_bf2f(mv_f2bf(a.w[l][(size_t)o * K + i])) : 0.0f;
    }
}

template <int MT, int NTW, bool CARRY, class NET>
__global__ __launch_bounds__(512) void k_sdf_col0_bf(NET net, const float* __restrict__ x, int n, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x, d0 = 3 + 6 * net.multires;
    float* act = smem;
    float* pe = act + ROWS * net.S;
    float* pts = pe + ((ROWS * d0 + 3) & ~3);
    float* out = pts + ROWS * 4;
    const int row0 = blockIdx.x * ROWS;
    for (int i = tid; i < ROWS * 3; i += 512) {
        const int row = row0 + i / 3;
        pts[i] = row < n ? x[3 * (size_t)row0 + i] : 0.0f;
    }
    __syncthreads();
    mv_sdf_eval_col0<MT, NTW, 8, CARRY>(net, act, pe, pts, out, tid);
    if (tid < ROWS && row0 + tid < n) y[row0 + tid] = out[tid];
}

// NET = MvNetBf (bf16 weights + activations) or MvNetBs<NS> (bf16 weights, activations as NS bf16 terms: tile_engine_bf16s.h)
template <int MT, int NTW, class NET>
static int launch_col0_bf(const NET& net, const float* x, int n, float* y, hipStream_t s) {
    const int rows = 16 * MT, d0 = 3 + 6 * net.multires;
    const size_t lds = ((size_t)rows * net.S + ((rows * d0 + 3) & ~3) + rows * 4 + rows) * 4;
    // MVSDF_BF_CARRY=1 (dev / tests): the weight-fetch scheme of k_sphere_trace (tile_engine_bf16.h, CARRIED) instead of the row-sample kernels' (ROLLING);
    // same arithmetic, bit-identical results (tests/test_gpu_bf16.py)
    static int carry = -1;
    if (carry < 0) { const char* ev = mv_dev_env("MVSDF_BF_CARRY"); carry = ev ? atoi(ev) : 0; }
    hipError_t e = carry ? hipFuncSetAttribute((const void*)k_sdf_col0_bf<MT, NTW, true, NET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                         : hipFuncSetAttribute((const void*)k_sdf_col0_bf<MT, NTW, false, NET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mv_check(e, "mvsdf_sdf_col0: LDS attribute");
    if (carry) hipLaunchKernelGGL((k_sdf_col0_bf<MT, NTW, true, NET>), dim3((n + rows - 1) / rows), dim3(512), lds, s, net, x, n, y);
    else hipLaunchKernelGGL((k_sdf_col0_bf<MT, NTW, false, NET>), dim3((n + rows - 1) / rows), dim3(512), lds, s, net, x, n, y);
    return mv_check(hipGetLastError(), "mvsdf_sdf_col0 (bf16)");
}

template <class NET>
static int dispatch_col0_bf(const NET& nb, const float* x, int n, float* y, int mt, hipStream_t s) {
    int mx = 0;
    for (int l = 0; l < nb.n_layers - 1; ++l) mx = nb.L[l].NT > mx ? nb.L[l].NT : mx;
    if (mx > 32) return mv_fail(-1, "mvsdf_sdf_col0: network too wide");
    if (mx > 16) return mt >= 2 ? launch_col0_bf<2, 4>(nb, x, n, y, s) : launch_col0_bf<1, 4>(nb, x, n, y, s);
    if (mt >= 4) return launch_col0_bf<4, 2>(nb, x, n, y, s);
    if (mt >= 2) return launch_col0_bf<2, 2>(nb, x, n, y, s);
    return launch_col0_bf<1, 2>(nb, x, n, y, s);
}

__global__ void k_det_math(int op, const float* __restrict__ x, int n, float* __restrict__ y0, float* __restrict__ y1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float a = 0.f, b = 0.f;
    switch (op) {
        case 0: a = dm_softplus100(v); b = dm_sigmoid100(v); break;
        case 1: a = dm_expneg(v); break;
        case 2: a = dm_log1p01(v); break;
        case 3: dm_sincos(v, &a, &b); break;
        case 4: a = dm_div100(v); b = dm_div_sqrt2(v); break;
        case 6: { const dm_f2 h = dm2_softplus100(dm_f2{v, -v}); a = h.x; b = h.y; break; }   // two-wide softplus: y0 = f(x), y1 = f(-x)
        case 7: a = dm_softplus100_lean(v); b = dm_softplus100_lean(-v); break;                  // the f32x3 engine's activation, scalar form
        case 8: { const dm_f2 h = dm2_softplus100_lean(dm_f2{v, -v}); a = h.x; b = h.y; break; }   // ... and the two-wide form the epilogue runs
        default: a = sqrtf(fabsf(v)); b = 1.0f / v; break;
    }
    y0[i] = a;
    if (y1) y1[i] = b;
}

template <int MT, int NTW, int NW, bool XR = false>
static int launch_col0(const MvNet& net, const float* x, int n, float* y, hipStream_t s) {
    const int rows = 16 * MT, d0 = 3 + 6 * net.multires;
    const size_t lds = ((size_t)rows * net.S + ((rows * d0 + 3) & ~3) + rows * 4 + rows) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)k_sdf_col0<MT, NTW, NW, XR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mv_check(e, "mvsdf_sdf_col0: LDS attribute");
    hipLaunchKernelGGL((k_sdf_col0<MT, NTW, NW, XR>), dim3((n + rows - 1) / rows), dim3(64 * NW), lds, s, net, x, n, y);
    return mv_check(hipGetLastError(), "mvsdf_sdf_col0");
}

// =============================================================================================================
extern "C" {

int mvsdf_version(void) { return 100; }

size_t mvsdf_packed_floats(int N, int K) { return mv_packed_floats(N, K); }

int mvsdf_fold_pack(const float* v, const float* g, int N, int K, float* w, float* wp, float* wpT, void* stream) {
    if (!v || !g || N <= 0 || K <= 0 || !w) return mv_fail(-1, "mvsdf_fold_pack: bad arguments (w must be given)");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fold, dim3((N + 3) / 4), dim3(256), 0, s, v, g, N, K, w);
    const size_t total = mv_packed_floats(N, K);
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    if (wp) hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, s, w, N, K, 0, wp);
    if (wpT) hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, s, w, N, K, 1, wpT);
    return mv_check(hipGetLastError(), "mvsdf_fold_pack");
}

static int fill_fold_args(FoldNetArgs& a, int n_layers, const int* N, const int* K, int* maxN, size_t* maxTot) {
    if (n_layers < 1 || n_layers > MV_FOLD_MAXL || !N || !K) return mv_fail(-1, "fold (net): bad layer count / dims");
    memset(&a, 0, sizeof(a));
    a.n_layers = n_layers;
    *maxN = 0; *maxTot = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (N[l] <= 0 || K[l] <= 0) return mv_fail(-1, "fold (net): bad dims");
        a.N[l] = N[l]; a.K[l] = K[l];
        if (N[l] > *maxN) *maxN = N[l];
        const size_t t = mv_packed_floats(N[l], K[l]) > mv_packed_floats(K[l], N[l]) ? mv_packed_floats(N[l], K[l]) : mv_packed_floats(K[l], N[l]);
        if (t > *maxTot) *maxTot = t;
    }
    return 0;
}

/* every layer of a network in one call: fold (one launch) + pack W and W^T (one launch).  Pointer arrays are HOST arrays of
 * device pointers; wp / wpT entries may be NULL. */
int mvsdf_fold_pack_net(int n_layers, const float* const* v, const float* const* g, const int* N, const int* K, float* const* w,
                        float* const* wp, float* const* wpT, void* stream) {
    FoldNetArgs a;
    int maxN; size_t maxTot;
    int rc = fill_fold_args(a, n_layers, N, K, &maxN, &maxTot);
    if (rc) return rc;
    if (!v || !g || !w || !wp || !wpT) return mv_fail(-1, "mvsdf_fold_pack_net: null argument");
    for (int l = 0; l < n_layers; ++l) {
        if (!v[l] || !w[l]) return mv_fail(-1, "mvsdf_fold_pack_net: null layer pointer");    // g[l] NULL: layer without weight norm (w = v)
        a.v[l] = v[l]; a.g[l] = g[l]; a.w[l] = w[l]; a.wp[l] = wp[l]; a.wpT[l] = wpT[l];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fold_net, dim3((maxN + 3) / 4, n_layers), dim3(256), 0, s, a);
    const int blocks = (int)((maxTot + 255) / 256 < 256 ? (maxTot + 255) / 256 : 256);
    hipLaunchKernelGGL(k_pack_net, dim3(blocks, 2 * n_layers), dim3(256), 0, s, a);
    return mv_check(hipGetLastError(), "mvsdf_fold_pack_net");
}

int mvsdf_fold_backward_net(int n_layers, const float* const* v, const float* const* g, const float* const* dW, const float* const* db,
                            const int* N, const int* K, float* const* dv, float* const* dg, float* const* dbias, int accumulate,
                            void* stream) {
    FoldNetArgs a;
    int maxN; size_t maxTot;
    int rc = fill_fold_args(a, n_layers, N, K, &maxN, &maxTot);
    if (rc) return rc;
    if (!v || !g || !dW || !dv || !dg || ((db == nullptr) != (dbias == nullptr))) return mv_fail(-1, "mvsdf_fold_backward_net: null argument");
    a.accumulate = accumulate;
    for (int l = 0; l < n_layers; ++l) {
        if (!v[l] || !dW[l] || !dv[l] || (g[l] && !dg[l])) return mv_fail(-1, "mvsdf_fold_backward_net: null layer pointer");
        a.v[l] = v[l]; a.g[l] = g[l]; a.dW[l] = dW[l]; a.dv[l] = dv[l]; a.dg[l] = dg[l];
        if (db) {
            if ((db[l] == nullptr) != (dbias[l] == nullptr)) return mv_fail(-1, "mvsdf_fold_backward_net: db / dbias must pair up");
            a.db[l] = db[l]; a.dbias[l] = dbias[l];
        }
    }
    hipLaunchKernelGGL(k_fold_bwd_net, dim3((maxN + 3) / 4, n_layers), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_fold_backward_net");
}

int mvsdf_fold_backward(const float* v, const float* g, const float* dW, int N, int K, float* dv, float* dg, void* stream) {
    if (!v || !g || !dW || !dv || !dg || N <= 0 || K <= 0) return mv_fail(-1, "mvsdf_fold_backward: bad arguments");
    hipLaunchKernelGGL(k_fold_bwd, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, g, dW, N, K, dv, dg);
    return mv_check(hipGetLastError(), "mvsdf_fold_backward");
}

int mvsdf_camera_rays(const float* uv, const float* pose, const float* intrinsics, int B, int P, float* ray_dirs, float* cam_loc,
                      void* stream) {
    if (!uv || !pose || !intrinsics || !ray_dirs || !cam_loc || B <= 0 || P <= 0) return mv_fail(-1, "mvsdf_camera_rays: bad arguments");
    hipLaunchKernelGGL(k_camera_rays, dim3((B * P + 255) / 256), dim3(256), 0, (hipStream_t)stream, uv, pose, intrinsics, B, P, ray_dirs,
                       cam_loc);
    return mv_check(hipGetLastError(), "mvsdf_camera_rays");
}

int mvsdf_sphere_intersection(const float* cam_loc, const float* ray_dirs, int B, int P, float r, float* t, uint8_t* mask, void* stream) {
    if (!cam_loc || !ray_dirs || !t || !mask || B <= 0 || P <= 0) return mv_fail(-1, "mvsdf_sphere_intersection: bad arguments");
    hipLaunchKernelGGL(k_sphere_intersection, dim3((B * P + 255) / 256), dim3(256), 0, (hipStream_t)stream, cam_loc, ray_dirs, B, P, r, t, mask);
    return mv_check(hipGetLastError(), "mvsdf_sphere_intersection");
}

size_t mvsdf_packed_bf16_bytes(int N, int K, int nsplit) { return mv_packed_bf16_elems(N, K, nsplit) * 2; }

/* trace_dtype = 3 / 4: the bf16 packs without duplicated columns (every activation, the positional encoding included, is split into bf16
 * terms in LDS: tile_engine_bf16s.h) */
int mvsdf_pack_bf16s_net(int n_layers, const float* const* w, const int* N, const int* K, void* const* wp16, void* stream) {
    if (n_layers < 1 || n_layers > MV_MAXL || !w || !N || !K || !wp16) return mv_fail(-1, "mvsdf_pack_bf16s_net: bad arguments");
    PackBfArgs a;
    memset(&a, 0, sizeof(a));
    size_t maxTot = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!w[l] || !wp16[l] || N[l] <= 0 || K[l] <= 0) return mv_fail(-1, "mvsdf_pack_bf16s_net: null layer pointer / bad dims");
        a.w[l] = w[l]; a.wp[l] = (uint16_t*)wp16[l]; a.N[l] = N[l]; a.K[l] = K[l]; a.nsplit[l] = 0;
        const size_t t = mv_packed_bf16_elems(N[l], K[l], 0);
        if (t > maxTot) maxTot = t;
    }
    const int blocks = (int)((maxTot + 255) / 256 < 256 ? (maxTot + 255) / 256 : 256);
    hipLaunchKernelGGL(k_pack_bf16_net, dim3(blocks, n_layers), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_pack_bf16s_net");
}

/* trace_dtype = 5: three-term bf16 packs of the fp32 weights (3 x mvsdf_packed_bf16_bytes(N, K, 0) bytes per layer) */
static int pack_x3_net(int n_layers, const float* const* w, const int* N, const int* K, void* const* out, int tr, void* stream) {
    if (n_layers < 1 || n_layers > MV_MAXL || !w || !N || !K || !out) return mv_fail(-1, "mvsdf_pack_bf16x3_net / mvsdf_pack_bf16x3t_net: bad arguments");
    PackBfArgs a;
    memset(&a, 0, sizeof(a));
    a.tr = tr;
    size_t maxTot = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!w[l] || !out[l] || N[l] <= 0 || K[l] <= 0) return mv_fail(-1, "mvsdf_pack_bf16x3_net / mvsdf_pack_bf16x3t_net: null layer pointer / bad dims");
        a.w[l] = w[l]; a.wp[l] = (uint16_t*)out[l]; a.N[l] = N[l]; a.K[l] = K[l]; a.nsplit[l] = 0;
        const size_t t = 3 * (tr ? mv_packed_bf16_elems(K[l], N[l], 0) : mv_packed_bf16_elems(N[l], K[l], 0));
        if (t > maxTot) maxTot = t;
    }
    const int blocks = (int)((maxTot + 255) / 256 < 512 ? (maxTot + 255) / 256 : 512);
    hipLaunchKernelGGL(k_pack_bf16x3_net, dim3(blocks, n_layers), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_pack_bf16x3_net");
}
int mvsdf_pack_bf16x3_net(int n_layers, const float* const* w, const int* N, const int* K, void* const* wp16, void* stream) {
    return pack_x3_net(n_layers, w, N, K, wp16, 0, stream);
}
/* the same pack of W^T (N, K: W's dims; 3 x mvsdf_packed_bf16_bytes(K, N, 0) bytes per layer): MvsdfNetDesc.wx3 of the transposed descriptor */
int mvsdf_pack_bf16x3t_net(int n_layers, const float* const* w, const int* N, const int* K, void* const* wx3t, void* stream) {
    return pack_x3_net(n_layers, w, N, K, wx3t, 1, stream);
}

int mvsdf_pack_bf16w_net(int n_layers, const float* const* w, const int* N, const int* K, float* const* wp_rounded, void* stream) {
    if (n_layers < 1 || n_layers > MV_MAXL || !w || !N || !K || !wp_rounded) return mv_fail(-1, "mvsdf_pack_bf16w_net: bad arguments");
    PackBfArgs a;
    memset(&a, 0, sizeof(a));
    size_t maxTot = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!w[l] || !wp_rounded[l] || N[l] <= 0 || K[l] <= 0) return mv_fail(-1, "mvsdf_pack_bf16w_net: null layer pointer / bad dims");
        a.w[l] = w[l]; a.wp[l] = (uint16_t*)wp_rounded[l]; a.N[l] = N[l]; a.K[l] = K[l];
        const size_t t = mv_packed_floats(N[l], K[l]);
        if (t > maxTot) maxTot = t;
    }
    const int blocks = (int)((maxTot + 255) / 256 < 256 ? (maxTot + 255) / 256 : 256);
    hipLaunchKernelGGL(k_pack_round_net, dim3(blocks, n_layers), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_pack_bf16w_net");
}

int mvsdf_sdf_col0(const MvsdfNetDesc* desc, const float* x, int n, float* y, int mt, void* stream) {
    if (!x || !y || n <= 0) return mv_fail(-1, "mvsdf_sdf_col0: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (desc && desc->trace_dtype == 1) return mv_fail(-2, "mvsdf_sdf_col0: trace_dtype 1 (bf16 weights AND 8-bit activations) was removed in round 5: use 3 (bf16x2: same speed, parity-checked)");
    if (desc && (desc->trace_dtype == 3 || desc->trace_dtype == 4)) {   // bf16 weights x activations carried as 2 / 3 bf16 terms
        MvNetBs<2> n2;
        MvNetBs<3> n3;
        int rcb = desc->trace_dtype == 3 ? mv_make_net_bs(desc, &n2, 2) : mv_make_net_bs(desc, &n3, 3);
        if (rcb) return rcb;
        return desc->trace_dtype == 3 ? dispatch_col0_bf(n2, x, n, y, mt, s) : dispatch_col0_bf(n3, x, n, y, mt, s);
    }
    if (desc && desc->trace_dtype == 5) {                       // fp32 weights and activations as three bf16 terms each, six products
        MvNetBs<3, 3> n33;
        int rcb = mv_make_net_bs(desc, &n33, 3);
        if (rcb) return rcb;
        return dispatch_col0_bf(n33, x, n, y, mt, s);
    }
    MvNet net;
    int rc = mv_make_net_trace(desc, &net);
    if (rc) return rc;
    if (mt != 1 && mt != 2 && mt != 4 && mt != 49) return mv_fail(-1, "mvsdf_sdf_col0: mt must be 1, 2, 4 (row tiles per workgroup) or 49 (the sphere tracer's carried-ring engine)");
    if (mt == 49) {                                                        // the sphere tracer's engine: weight ring carried across layers (two column tiles per wave: width <= 256)
        if (mv_wide(net)) return mv_fail(-1, "mvsdf_sdf_col0: mt = 49 needs a hidden width <= 256");
        return launch_col0<1, 2, 8, true>(net, x, n, y, s);
    }
    int maxnt = 0;
    for (int l = 0; l < net.n_layers - 1; ++l) maxnt = net.L[l].NT > maxnt ? net.L[l].NT : maxnt;
    const bool eight = maxnt >= 16;                              // 8 waves x 2 column tiles from width 256 on; narrow nets: 4 waves x 4 tiles
    if (eight) {
        if (mv_wide(net)) return mt >= 2 ? launch_col0<2, 4, 8>(net, x, n, y, s) : launch_col0<1, 4, 8>(net, x, n, y, s);
        if (mt >= 4) return launch_col0<4, 2, 8>(net, x, n, y, s);
        if (mt >= 2) return launch_col0<2, 2, 8>(net, x, n, y, s);
        return launch_col0<1, 2, 8>(net, x, n, y, s);
    }
    if (mv_wide(net)) return mt >= 2 ? launch_col0<2, 8, 4>(net, x, n, y, s) : launch_col0<1, 8, 4>(net, x, n, y, s);
    if (mt >= 4) return launch_col0<4, 4, 4>(net, x, n, y, s);
    if (mt >= 2) return launch_col0<2, 4, 4>(net, x, n, y, s);
    return launch_col0<1, 4, 4>(net, x, n, y, s);
}

int mvsdf_det_math(int op, const float* x, int n, float* y0, float* y1, void* stream) {
    if (!x || !y0 || n <= 0) return mv_fail(-1, "mvsdf_det_math: bad arguments");
    hipLaunchKernelGGL(k_det_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, x, n, y0, y1);
    return mv_check(hipGetLastError(), "mvsdf_det_math");
}

}  // extern "C"

// ================================================================================================ step prologue (step_internal.h)
// Everything a training forward needs before the tracer, in ONE launch: the weight-norm fold of every layer of both networks (idr.py:70-71),
// the MFMA packs of W and W^T, the bf16 packs of the tracing MLP (trace_dtype = 1) and the camera rays (rend_util.py:48-75,87-100) -- the work
// of k_fold_net, k_pack_net, k_pack_bf16_net and k_camera_rays.  A workgroup owns 16 rows of one layer: its sixteen waves fold one row each
// (the same serialised fmaf chain over k as k_fold_net: bit-identical), the 16 folded rows stay in LDS and the workgroup writes its tile row of the
// W pack, its k-block column of the W^T pack and its tile row of the bf16 pack from there.  The last workgroups compute the rays.
// 16 waves per workgroup, one row each: a row's norm is a serial fmaf chain of ~K steps (~10 us), so the rows must all run side by side.
struct ProloArgs {
    FoldNetArgs f;
    int blk0[MV_FOLD_MAXL + 1];          // first workgroup of each layer; blk0[n_layers] = first ray workgroup
 uint16_t* wp16[MV_FOLD_MAXL]; int nsplit[MV_FOLD_MAXL];
    uint16_t* wx3[MV_FOLD_MAXL]; uint16_t* wx3T[MV_FOLD_MAXL];   // optional three-term bf16 packs of W_l / W_l^T (the differentiable chains of chain_x3.h)
    int wp16_mode;                       // 0: bf16 pack; 1: wp16[l] receives the fp32 pack of the bf16-rounded weights (trace_dtype = 2); 2: the three-term bf16 pack (trace_dtype = 5)
    const float* uv; const float* pose; const float* Kin; int B, P; float* dirs; float* cam_loc;
    uint8_t* ones;                       // optional [B * P]: filled with 1 (the all-ones object mask of the output dict, idr.py:187)
    unsigned long long* counters;        // optional [16]: zeroed (the tracer's device counters: saves its memset node)
    const float* stage_src; float* stage_a; float* stage_b; int stage_na, stage_nb;   // optional: pinned host memory [na | nb] -> two device buffers
    int ld;                              // LDS row stride (floats) >= max K
};
__global__ __launch_bounds__(1024) void k_step_prologue(ProloArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];              // [16][ld]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nl = a.f.n_layers;
    if ((int)blockIdx.x >= a.blk0[nl]) {                                       // ---- camera rays
        const int idx = ((int)blockIdx.x - a.blk0[nl]) * 1024 + tid;
        if (a.counters && idx < 16) a.counters[idx] = 0ull;
        if (a.stage_src) {                                                      // the step's CPU-generator draws, read straight from pinned host memory
            const int nthr = ((int)gridDim.x - a.blk0[nl]) * 1024, ntot = a.stage_na + a.stage_nb;
            for (int i = idx; i < ntot; i += nthr) {
                const float x = a.stage_src[i];
                if (i < a.stage_na) a.stage_a[i] = x; else a.stage_b[i - a.stage_na] = x;
            }
        }
        if (idx >= a.B * a.P) return;
        if (a.ones) a.ones[idx] = 1;
        const int b = idx / a.P;
        const float* p = a.pose + 16 * b;
        const float* k = a.Kin + 16 * b;
        const float fx = k[0], fy = k[5], cx = k[2], cy = k[6], sk = k[1];
        const float cl[3] = {p[3], p[7], p[11]};
        if (idx == b * a.P) { a.cam_loc[3 * b] = cl[0]; a.cam_loc[3 * b + 1] = cl[1]; a.cam_loc[3 * b + 2] = cl[2]; }
        const float x = a.uv[2 * (size_t)idx] + 0.5f, y = a.uv[2 * (size_t)idx + 1] + 0.5f, z = 1.0f;
        const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z;
        const float yl = (y - cy) / fy * z;
        const float h[4] = {xl, yl, z, 1.0f};
        float wvv[3];
        for (int r = 0; r < 3; ++r) {
            float acc = 0.0f;
            for (int c = 0; c < 4; ++c) acc = fmaf(p[4 * r + c], h[c], acc);
            wvv[r] = acc - cl[r];
        }
        float nn = sqrtf(wvv[0] * wvv[0] + wvv[1] * wvv[1] + wvv[2] * wvv[2]);
        if (nn < 1e-12f) nn = 1e-12f;
        a.dirs[3 * (size_t)idx] = wvv[0] / nn;
        a.dirs[3 * (size_t)idx + 1] = wvv[1] / nn;
        a.dirs[3 * (size_t)idx + 2] = wvv[2] / nn;
        return;
    }
    int l = 0;
    while (l + 1 < nl && (int)blockIdx.x >= a.blk0[l + 1]) ++l;
    const int rg = (int)blockIdx.x - a.blk0[l];                                // 16-row group of layer l (groups beyond the rows: pack padding only)
    const int N = a.f.N[l], K = a.f.K[l], ld = a.ld;
    // ---- fold: wave wv takes row 16 rg + wv
    {
        const int jl = wv, j = 16 * rg + jl;
        float* tr = tile + jl * ld;
        if (j >= N) { for (int k = lane; k < K; k += 64) tr[k] = 0.0f; }
        else {
            const float* vr = a.f.v[l] + (size_t)j * K;
            float* wr = a.f.w[l] + (size_t)j * K;
            if (!a.f.g[l]) {                                                   // weight_norm=False: w = v
                for (int k = lane; k < K; k += 64) { const float x = vr[k]; wr[k] = x; tr[k] = x; }
            } else if (K <= 64 * 9) {
                // the whole row is requested before the k-ordered chain starts (chunk by chunk inside the chain, every chunk's load round trip sat
                // between two pieces of it: 4 x ~2 us of the launch's 20 at K = 256) and is reused for the scaling pass
                float mine[9];
                const float gj = a.f.g[l][j];
#pragma unroll
                for (int c = 0; c < 9; ++c) { const int k = 64 * c + lane; const float x = vr[k < K ? k : K - 1]; mine[c] = k < K ? x : 0.0f; }
                // the k-ascending fmaf chain of orc_fold, bit for bit.  Full 64-blocks are unrolled with constant lane numbers: the v_readlane of the
                // elements ahead are independent of the chain and get issued early, so a step costs the fma's latency and not the
                // readlane -> SGPR -> VALU round trip of a rolled loop (which made this chain ~10 us of the launch's 18)
                float ss = 0.0f;
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    if (64 * (c + 1) <= K) {
#pragma unroll
                        for (int i = 0; i < 64; ++i) {
                            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine[c]), i));
                            ss = fmaf(x, x, ss);
                        }
                    } else if (64 * c < K) {
                        const int n = K - 64 * c;
                        for (int i = 0; i < n; ++i) {
                            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine[c]), i));
                            ss = fmaf(x, x, ss);
                        }
                    }
                }
                const float sc = gj / sqrtf(ss);
#pragma unroll
                for (int c = 0; c < 9; ++c) { const int k = 64 * c + lane; if (k < K) { const float x = mine[c] * sc; wr[k] = x; tr[k] = x; } }
            } else {
                float ss = 0.0f;
                for (int k0 = 0; k0 < K; k0 += 64) {
                    const float mine = (k0 + lane < K) ? vr[k0 + lane] : 0.0f;
                    const int n = min(64, K - k0);
                    for (int i = 0; i < n; ++i) {
                        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), i));
                        ss = fmaf(x, x, ss);
                    }
                }
                const float sc = a.f.g[l][j] / sqrtf(ss);
                for (int k = lane; k < K; k += 64) { const float x = vr[k] * sc; wr[k] = x; tr[k] = x; }
            }
        }
    }
    __syncthreads();
    // ---- pack of W: tile row ct = rg, every k-block
    const int NT = mv_ceil16(N) / 16, KB = mv_kpad(K) / 16;
    if (a.f.wp[l] && rg < NT) {
        float* dst = a.f.wp[l] + (size_t)rg * KB * 256;
        for (int idx = tid; idx < KB * 256; idx += 1024) {
            const int s = idx & 3, ln = (idx >> 2) & 63, kb = idx >> 8;
            const int ol = ln & 15, i = kb * 16 + 4 * s + (ln >> 4);
            dst[idx] = (i < K) ? tile[ol * ld + i] : 0.0f;                      // rows >= N are zero in the tile
        }
    }
    // ---- pack of W^T: k-block kb' = rg of every tile row ct' (columns of W)
    if (a.f.wpT[l]) {
        const int NTt = mv_ceil16(K) / 16, KBt = mv_kpad(N) / 16;
        for (int idx = tid; idx < NTt * 256; idx += 1024) {
            const int e = idx & 255, ct = idx >> 8;
            const int s = e & 3, ln = (e >> 2) & 63;
            const int o = ct * 16 + (ln & 15), il = 4 * s + (ln >> 4);
            a.f.wpT[l][((size_t)ct * KBt + rg) * 256 + e] = (o < K) ? tile[il * ld + o] : 0.0f;
        }
    }
    // ---- bf16 pack of the tracing MLP: tile row ct = rg
    if (a.wp16[l] && rg < NT && a.wp16_mode == 1) {                                 // trace_dtype = 2: the W pack again, values rounded to bf16
        float* dst = (float*)a.wp16[l] + (size_t)rg * KB * 256;
        for (int idx = tid; idx < KB * 256; idx += 1024) {
            const int s = idx & 3, ln = (idx >> 2) & 63, kb = idx >> 8;
            const int ol = ln & 15, i = kb * 16 + 4 * s + (ln >> 4);
            dst[idx] = (i < K) ? mv_bf2f(mv_f2bf(tile[ol * ld + i])) : 0.0f;
        }
    } else if (a.wp16[l] && rg < NT && a.wp16_mode == 2) {                     // trace_dtype = 5: three bf16 terms of every weight (k_pack_bf16x3_net's layout)
        const int KB32 = mv_bf_kb(K, 0);
        uint16_t* dst = a.wp16[l] + (size_t)rg * KB32 * 3 * 512;
        for (int idx = tid; idx < KB32 * 3 * 512; idx += 1024) {
            const int i = idx & 7, ln = (idx >> 3) & 63, blk = idx >> 9, term = blk % 3, kb = blk / 3;
            const int ol = ln & 15, kp = kb * 32 + 8 * (ln >> 4) + i;
            uint16_t v = 0;
            if (16 * rg + ol < N && kp < K) {
                float r = tile[ol * ld + kp];
                if (fabsf(r) < 9.094947017729282e-13f) r = 0.0f;
                v = mv_f2bf(r);
                for (int t = 0; t < term; ++t) { r = r - mv_bf2f(v); v = mv_f2bf(r); }
            }
            dst[idx] = v;
        }
    }
    if (a.wx3[l] && rg < NT) {                                                  // three bf16 terms of every weight (k_pack_bf16x3_net's layout) for the differentiable chains
        const int KB32 = mv_bf_kb(K, 0);
        uint16_t* dst = a.wx3[l] + (size_t)rg * KB32 * 3 * 512;
        for (int idx = tid; idx < KB32 * 3 * 512; idx += 1024) {
            const int i = idx & 7, ln = (idx >> 3) & 63, blk = idx >> 9, term = blk % 3, kb = blk / 3;
            const int ol = ln & 15, kp = kb * 32 + 8 * (ln >> 4) + i;
            uint16_t v = 0;
            if (16 * rg + ol < N && kp < K) {
                float r = tile[ol * ld + kp];
                if (fabsf(r) < 9.094947017729282e-13f) r = 0.0f;
                v = mv_f2bf(r);
                for (int t = 0; t < term; ++t) { r = r - mv_bf2f(v); v = mv_f2bf(r); }
            }
            dst[idx] = v;
        }
    }
    if (a.wx3T[l]) {                                                            // ... and of W^T: this workgroup's 16 rows of W are half a k-block of every column tile
        const int NTt = mv_ceil16(K) / 16, KBt = mv_bf_kb(N, 0);
        const int kbt = rg >> 1, qh = 2 * (rg & 1);                             // k-block of W^T, lane quarter (lane >> 4) of the first 8 of the 16 rows
        for (int g = tid; g < NTt * 3 * 32; g += 1024) {                        // one 16-byte group (8 consecutive k of one lane) per iteration
            const int jh = g & 1, rl = (g >> 1) & 15, term = (g >> 5) % 3, ct = (g >> 5) / 3;
            const int o = ct * 16 + rl;                                         // column of W = row of W^T
            uint16_t v8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint16_t v = 0;
                if (o < K) {                                                    // (rows >= N are zero in the tile)
                    float r = tile[(8 * jh + i) * ld + o];
                    if (fabsf(r) < 9.094947017729282e-13f) r = 0.0f;
                    v = mv_f2bf(r);
                    for (int t = 0; t < term; ++t) { r = r - mv_bf2f(v); v = mv_f2bf(r); }
                }
                v8[i] = v;
            }
            const int ln = 16 * (qh + jh) + rl;
            uint4 pk;
            pk.x = v8[0] | ((uint32_t)v8[1] << 16); pk.y = v8[2] | ((uint32_t)v8[3] << 16); pk.z = v8[4] | ((uint32_t)v8[5] << 16); pk.w = v8[6] | ((uint32_t)v8[7] << 16);
            if (kbt < KBt) *(uint4*)(a.wx3T[l] + ((((size_t)ct * KBt + kbt) * 3 + term) * 64 + ln) * 8) = pk;
        }
    }
    if (a.wp16[l] && rg < NT && a.wp16_mode != 1 && a.wp16_mode != 2) {
        const int ns = a.nsplit[l], KB32 = mv_bf_kb(K, ns);
        uint16_t* dst = a.wp16[l] + (size_t)rg * KB32 * 512;
        for (int idx = tid; idx < KB32 * 512; idx += 1024) {
            const int i = idx & 7, ln = (idx >> 3) & 63, kb = idx >> 9;
            const int ol = ln & 15, kp = kb * 32 + 8 * (ln >> 4) + i;
            const int col = kp < K ? kp : (kp < K + ns ? kp - ns : -1);        // the lo copy of a split column shares the hi column's weight
            dst[idx] = (16 * rg + ol < N && col >= 0) ? mv_f2bf(tile[ol * ld + col]) : (uint16_t)0;
        }
    }
}

int mv_step_prologue(int n_layers, const float* const* v, const float* const* g, const int* N, const int* K, float* const* w, float* const* wp,
                     float* const* wpT, void* const* wp16, const int* nsplit, int wp16_mode, void* const* wx3, void* const* wx3T, const float* uv, const float* pose,
                     const float* intrinsics, int B, int P,
                     float* ray_dirs, float* cam_loc, uint8_t* ones, unsigned long long* counters, const float* stage_src, float* stage_a, int stage_na,
                     float* stage_b, int stage_nb, void* stream) {
    ProloArgs a;
    int maxN; size_t maxTot;
    int rc = fill_fold_args(a.f, n_layers, N, K, &maxN, &maxTot);
    if (rc) return rc;
    if (!v || !g || !w || !wp || !wpT || !uv || !pose || !intrinsics || !ray_dirs || !cam_loc || B <= 0 || P <= 0) return mv_fail(-1, "mv_step_prologue: null argument");
    int blk = 0, maxK = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!v[l] || !w[l] || !wp[l] || !wpT[l]) return mv_fail(-1, "mv_step_prologue: null layer pointer");
        a.f.v[l] = v[l]; a.f.g[l] = g[l]; a.f.w[l] = w[l]; a.f.wp[l] = wp[l]; a.f.wpT[l] = wpT[l];
        a.wp16[l] = wp16 ? (uint16_t*)wp16[l] : nullptr; a.nsplit[l] = nsplit ? nsplit[l] : 0;
        a.wx3[l] = wx3 ? (uint16_t*)wx3[l] : nullptr; a.wx3T[l] = wx3T ? (uint16_t*)wx3T[l] : nullptr;
        a.blk0[l] = blk; blk += mv_kpad(N[l]) / 16;                              // row groups incl. the W^T pack's zero padding (N padded to 32)
        if (K[l] > maxK) maxK = K[l];
    }
    for (int l = n_layers; l < MV_FOLD_MAXL; ++l) { a.wp16[l] = nullptr; a.wx3[l] = a.wx3T[l] = nullptr; a.nsplit[l] = 0; a.blk0[l] = blk; }
    a.blk0[n_layers] = blk;
    a.uv = uv; a.pose = pose; a.Kin = intrinsics; a.B = B; a.P = P; a.dirs = ray_dirs; a.cam_loc = cam_loc;
    a.wp16_mode = wp16_mode;
    a.ones = ones; a.counters = counters;
    a.stage_src = stage_src; a.stage_a = stage_a; a.stage_b = stage_b; a.stage_na = stage_src ? stage_na : 0; a.stage_nb = stage_src ? stage_nb : 0;
    if (stage_src && (!stage_a || !stage_b || stage_na < 0 || stage_nb < 0)) return mv_fail(-1, "mv_step_prologue: staged inputs without targets");
    a.ld = ((maxK + 3) & ~3) + 4;
    const size_t lds = (size_t)16 * a.ld * sizeof(float);
    blk += (B * P + 1023) / 1024;
    hipLaunchKernelGGL(k_step_prologue, dim3(blk), dim3(1024), lds, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mv_step_prologue");
}
