// step_kernels.hip -- the bookkeeping of one training step between the big kernels, as a handful of small launches instead of
// dozens of framework ops: row partition (hit rays first), output gathers of IDRNetwork.forward (idr.py:253-304) and the assembly
// of the upstream gradients of the fused SDF backward.  Index / copy work: bit-exact by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "capi_util.h"
#include "det_math.h"

// ---------------------------------------------------------------------------------------------------------------
// Stable partition of the rays by "surface" = net_mask & object_mask: perm = [surface rays in ray order | the others in ray order],
// inv = inverse permutation, true_rows = ranks (positions among the surface rays) of the surface rays inside true_mask, in order;
// counts = {#surface, #surface & true}.  view_sorted[r] = -ray_dirs[perm[r]] (the rendering net's view directions, idr.py:300).
// One workgroup walks the rays in chunks of 1024 (R is a few thousand per step).
__global__ __launch_bounds__(1024) void k_partition_rays(const uint8_t* __restrict__ net_mask, const uint8_t* __restrict__ object_mask,
                                                         const uint8_t* __restrict__ true_mask, const float* __restrict__ ray_dirs, int R,
                                                         long long* __restrict__ perm, long long* __restrict__ inv,
                                                         long long* __restrict__ true_rows, long long* __restrict__ counts,
                                                         float* __restrict__ view_sorted) {
    __shared__ int wsum[3][16];
    __shared__ int base[3];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // pass 1: number of surface rays (offset of the second group)
    int local = 0;
    for (int i = tid; i < R; i += 1024) local += (net_mask[i] && (!object_mask || object_mask[i])) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if (lane == 0) wsum[0][w] = local;
    __syncthreads();
    int n_hit = 0;
    for (int k = 0; k < 16; ++k) n_hit += wsum[0][k];
    if (tid == 0) { base[0] = 0; base[1] = n_hit; base[2] = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < R; i0 += 1024) {
        const int i = i0 + tid;
        const bool in = i < R;
        const bool hit = in && net_mask[i] && (!object_mask || object_mask[i]);
        const bool rest = in && !hit;
        const bool tr = hit && (!true_mask || true_mask[i]);
        const unsigned long long bh = __ballot(hit), br = __ballot(rest), bt = __ballot(tr);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { wsum[0][w] = __popcll(bh); wsum[1][w] = __popcll(br); wsum[2][w] = __popcll(bt); }
        __syncthreads();
        int oh = base[0], orr = base[1], ot = base[2];
        for (int k = 0; k < w; ++k) { oh += wsum[0][k]; orr += wsum[1][k]; ot += wsum[2][k]; }
        if (in) {
            const int pos = hit ? oh + __popcll(bh & below) : orr + __popcll(br & below);
            perm[pos] = i; inv[i] = pos;
            if (view_sorted) for (int c = 0; c < 3; ++c) view_sorted[3 * (size_t)pos + c] = -ray_dirs[3 * (size_t)i + c];
            if (tr) true_rows[ot + __popcll(bt & below)] = pos;
        }
        __syncthreads();
        if (tid == 0) {
            int a = 0, b = 0, c = 0;
            for (int k = 0; k < 16; ++k) { a += wsum[0][k]; b += wsum[1][k]; c += wsum[2][k]; }
            base[0] += a; base[1] += b; base[2] += c;
        }
        __syncthreads();
    }
    if (tid == 0) { counts[0] = base[0]; counts[1] = base[2]; }
}

// ---------------------------------------------------------------------------------------------------------------
// Point groups of a training step in the reference's order (idr.py:253-257): 0 = hit rays (N, evaluation rows E..E+N),
// 1 = eikonal samples (rows 0..n_eik), 2 = on-surface samples, 3 = jittered samples (n_ds rows each).  A term selects groups by mask.
struct StepGroups { int n_eik, n_ds, E, N, mask; };
__device__ __forceinline__ int mv_group_total(const StepGroups& g) {
    return ((g.mask & 1) ? g.N : 0) + ((g.mask & 2) ? g.n_eik : 0) + ((g.mask & 4) ? g.n_ds : 0) + ((g.mask & 8) ? g.n_ds : 0);
}
// i-th row of the concatenation of the selected groups -> evaluation row
__device__ __forceinline__ int mv_group_row(const StepGroups& g, int i) {
    if (g.mask & 1) { if (i < g.N) return g.E + i; i -= g.N; }
    if (g.mask & 2) { if (i < g.n_eik) return i; i -= g.n_eik; }
    if (g.mask & 4) { if (i < g.n_ds) return g.n_eik + i; i -= g.n_ds; }
    return g.n_eik + g.n_ds + i;
}

// Outputs of the training forward gathered from the fused evaluation (rows [samples (E) | rays sorted, hit first]).  N / n_true come
// from DEVICE memory, so the launch does not wait for the host to learn them.
struct StepOutArgs {
    int R, E, n_eik, n_ds, Nout, d_mask, e_mask;
    const long long* counts;                                            // {N, n_true}
    const float* x_eval; const float* y_eval; const float* n_eval;      // [E+R][3], [E+R][Nout], [E+R][3]
    const long long* inv; const long long* true_rows;
    const float* rgb_sorted;                                            // [R][3]: rendering-net output of every sorted ray row
    float* rgb_values;      // [R][3]
    float* sdf_output;      // [R]
    float* diff_pts;        // [R][3], first N valid
    float* eik_out;         // first (N if selected) + samples valid
    float* points_hom;      // [.][4]
    float* grad_theta;      // [.][3]
    float* surf;            // first n_true + n_eik valid
};
__global__ void k_step_outputs(StepOutArgs a) {
    const int N = (int)a.counts[0], n_true = (int)a.counts[1];
    const StepGroups gd = {a.n_eik, a.n_ds, a.E, N, a.d_mask}, ge = {a.n_eik, a.n_ds, a.E, N, a.e_mask};
    const int nd = mv_group_total(gd), ne = mv_group_total(ge);
    const int seg0 = a.R, seg1 = seg0 + N, seg2 = seg1 + nd, seg3 = seg2 + ne, seg4 = seg3 + n_true + a.n_eik;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < seg4; i += gridDim.x * blockDim.x) {
        if (i < seg0) {                                                     // per ray: rgb (1 where not hit, idr.py:302-304), sdf_output
            const int pos = (int)a.inv[i];
            const bool hit = pos < N;
            for (int c = 0; c < 3; ++c) a.rgb_values[3 * (size_t)i + c] = hit ? a.rgb_sorted[3 * (size_t)pos + c] : 1.0f;
            a.sdf_output[i] = a.y_eval[(size_t)(a.E + pos) * a.Nout];
        } else if (i < seg1) {
            const int k = i - seg0;
            for (int c = 0; c < 3; ++c) a.diff_pts[3 * (size_t)k + c] = a.x_eval[3 * (size_t)(a.E + k) + c];
        } else if (i < seg2) {
            const int k = i - seg1, row = mv_group_row(gd, k);
            a.eik_out[k] = a.y_eval[(size_t)row * a.Nout];
            for (int c = 0; c < 3; ++c) a.points_hom[4 * (size_t)k + c] = a.x_eval[3 * (size_t)row + c];
            a.points_hom[4 * (size_t)k + 3] = 1.0f;
        } else if (i < seg3) {
            const int k = i - seg2, row = mv_group_row(ge, k);
            for (int c = 0; c < 3; ++c) a.grad_theta[3 * (size_t)k + c] = a.n_eval[3 * (size_t)row + c];
        } else {
            const int k = i - seg3;
            const int row = k < n_true ? a.E + (int)a.true_rows[k] : (k - n_true);
            a.surf[k] = a.y_eval[(size_t)row * a.Nout + 1];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Upstream gradients of the fused SDF backward (rows [0, Mb) = [samples | hit rays]):
//   stage 0 (before the input-adjoint pass): dy = 0 except dy[E+i][2:] = dfeat_i; dn = 0 except dn[E+i] = dnrm_i (rendering-net adjoints)
//   stage 1 (after it): SampleNetwork's scalar fbar_i = -(xbar_i . v_i) / (n_i . v_i) with xbar = d_diff + dp + dx (sample_network.py:10-20)
//            added to dy[E+i][0]; d(eikonal_output) -> column 0, d(surf_indicator_output) -> column 1, d(grad_theta) -> dn.
struct StepBwdArgs {
    int E, N, Nout, Mb, n_true, n_eik, n_ds, d_mask, e_mask, din_ld, din_feat0, din_nrm0, use_geo;
    const float* din;                     // [N][din_ld] adjoint of the rendering net's input (null: none)
    const float* d_diff; const float* dx; // [N][3] (either may be null)
    const float* view_sorted;             // [R][3] = -ray direction of sorted row
    const float* n_eval;                  // [E+R][3]
    const long long* true_rows;
    const float* d_eo; const float* d_gth; const float* d_si;          // upstream of eikonal_output / grad_theta / surf (null: none)
    float* dy; float* dn;                 // [Mb][Nout], [Mb][3]
    int no_fbar;                          // stage 1 without SampleNetwork's term (added later by k_step_bwd_fbar)
    float* fbar;                          // k_step_bwd_fbar: [N]
};
__global__ void k_step_bwd_stage0(StepBwdArgs a) {
    const size_t total = (size_t)a.Mb * (a.Nout + 3);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (a.Nout + 3)), c = (int)(i - (size_t)row * (a.Nout + 3));
        const int k = row - a.E;
        if (c < a.Nout) {
            float v = 0.0f;
            if (k >= 0 && c >= 2 && a.din) v = a.din[(size_t)k * a.din_ld + a.din_feat0 + (c - 2)];
            a.dy[(size_t)row * a.Nout + c] = v;
        } else {
            float v = 0.0f;
            if (k >= 0 && a.din && a.use_geo && a.din_nrm0 >= 0) v = a.din[(size_t)k * a.din_ld + a.din_nrm0 + (c - a.Nout)];   // din_nrm0 < 0: mode 'no_normal'
            a.dn[(size_t)row * 3 + (c - a.Nout)] = v;
        }
    }
}
__global__ void k_step_bwd_stage1(StepBwdArgs a) {
    const StepGroups gd = {a.n_eik, a.n_ds, a.E, a.N, a.d_mask}, ge = {a.n_eik, a.n_ds, a.E, a.N, a.e_mask};
    const int nd = mv_group_total(gd), ne = mv_group_total(ge);
    const int seg0 = a.N, seg1 = seg0 + (a.d_eo ? nd : 0), seg2 = seg1 + (a.d_gth ? ne : 0), seg3 = seg2 + (a.d_si ? a.n_true + a.n_eik : 0);
    // the groups touch disjoint (row, column) cells except fbar / d_eo on column 0 of the hit rows: the hit-row d_eo cells are folded
    // into the fbar thread.
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < seg3; i += gridDim.x * blockDim.x) {
        if (i < seg0) {
            const int k = i, row = a.E + k;
            float xb[3], v[3], dot = 0.f, num = 0.f;
            for (int c = 0; c < 3; ++c) {
                xb[c] = (a.d_diff ? a.d_diff[3 * (size_t)k + c] : 0.f);
                if (a.din && a.use_geo) xb[c] += a.din[(size_t)k * a.din_ld + c];
                if (a.dx) xb[c] += a.dx[3 * (size_t)k + c];
                v[c] = -a.view_sorted[3 * (size_t)k + c];
            }
            for (int c = 0; c < 3; ++c) { num += xb[c] * v[c]; dot += a.n_eval[3 * (size_t)row + c] * v[c]; }
            float add = a.no_fbar ? 0.0f : -num / dot;
            if (a.d_eo && (a.d_mask & 1)) add += a.d_eo[k];                  // the hit group leads eikonal_output when selected
            a.dy[(size_t)row * a.Nout] += add;
        } else if (i < seg1) {
            const int k = i - seg0, row = mv_group_row(gd, k);
            if (row < a.E) a.dy[(size_t)row * a.Nout] += a.d_eo[k];         // hit rows were handled above
        } else if (i < seg2) {
            const int k = i - seg1, row = mv_group_row(ge, k);
            for (int c = 0; c < 3; ++c) a.dn[3 * (size_t)row + c] += a.d_gth[3 * (size_t)k + c];
        } else {
            const int k = i - seg2;
            const int row = k < a.n_true ? a.E + (int)a.true_rows[k] : (k - a.n_true);
            a.dy[(size_t)row * a.Nout + 1] += a.d_si[k];
        }
    }
}

// SampleNetwork's scalar alone (sample_network.py:10-20 backward): fbar_i = -(xbar_i . v_i) / (n_i . v_i), xbar = d_diff + dp + dx; written to
// fbar[i] and added to dy[E + i][0]
__global__ void k_step_bwd_fbar(StepBwdArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    const int row = a.E + k;
    float dot = 0.f, num = 0.f;
    for (int c = 0; c < 3; ++c) {
        float xb = (a.d_diff ? a.d_diff[3 * (size_t)k + c] : 0.f);
        if (a.din && a.use_geo) xb += a.din[(size_t)k * a.din_ld + c];
        if (a.dx) xb += a.dx[3 * (size_t)k + c];
        const float v = -a.view_sorted[3 * (size_t)k + c];
        num += xb * v; dot += a.n_eval[3 * (size_t)row + c] * v;
    }
    const float f = -num / dot;
    a.fbar[k] = f;
    a.dy[(size_t)row * a.Nout] += f;
}

extern "C" {

int mvsdf_partition_rays(const uint8_t* net_mask, const uint8_t* object_mask, const uint8_t* true_mask, const float* ray_dirs, int R,
                         long long* perm, long long* inv, long long* true_rows, long long* counts, float* view_sorted, void* stream) {
    if (!net_mask || !perm || !inv || !true_rows || !counts || R <= 0 || (view_sorted && !ray_dirs))
        return mv_fail(-1, "mvsdf_partition_rays: bad arguments");
    hipLaunchKernelGGL(k_partition_rays, dim3(1), dim3(1024), 0, (hipStream_t)stream, net_mask, object_mask, true_mask, ray_dirs, R, perm, inv,
                       true_rows, counts, view_sorted);
    return mv_check(hipGetLastError(), "mvsdf_partition_rays");
}

int mvsdf_step_outputs(int R, int n_eik, int n_ds, int Nout, const long long* counts, const float* x_eval, const float* y_eval,
                       const float* n_eval, const long long* inv, const long long* true_rows, const float* rgb_sorted, int d_mask,
                       int e_mask, float* rgb_values, float* sdf_output, float* diff_pts, float* eik_out, float* points_hom, float* grad_theta,
                       float* surf, void* stream) {
    if (R <= 0 || n_eik < 0 || n_ds < 0 || !counts || !x_eval || !y_eval || !n_eval || !inv || !true_rows || !rgb_sorted || !rgb_values ||
        !sdf_output || !diff_pts || !eik_out || !points_hom || !grad_theta || !surf || (d_mask & ~15) || (e_mask & ~15))
        return mv_fail(-1, "mvsdf_step_outputs: bad arguments");
    StepOutArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R; a.E = n_eik + 2 * n_ds; a.n_eik = n_eik; a.n_ds = n_ds; a.Nout = Nout; a.d_mask = d_mask; a.e_mask = e_mask; a.counts = counts;
    a.x_eval = x_eval; a.y_eval = y_eval; a.n_eval = n_eval; a.inv = inv; a.true_rows = true_rows; a.rgb_sorted = rgb_sorted;
    a.rgb_values = rgb_values; a.sdf_output = sdf_output; a.diff_pts = diff_pts; a.eik_out = eik_out; a.points_hom = points_hom;
    a.grad_theta = grad_theta; a.surf = surf;
    const long long worst = 5ll * R + 3ll * a.E + n_eik;                  // rays + diff_pts + two group lists + surf at N = R
    hipLaunchKernelGGL(k_step_outputs, dim3((unsigned)((worst + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_step_outputs");
}

int mvsdf_step_backward_inputs(int stage, int n_eik, int n_ds, int N, int Nout, int n_true, const float* din, int din_ld, int din_feat0,
                               int din_nrm0, int use_geo, const float* d_diff, const float* dx, const float* view_sorted, const float* n_eval,
                               const long long* true_rows, const float* d_eo, const float* d_gth, const float* d_si, int d_mask, int e_mask,
                               float* dy, float* dn, void* stream) {
    const int E = n_eik + 2 * n_ds;
    if (n_eik < 0 || n_ds < 0 || N < 0 || E + N <= 0 || !dy || !dn || (stage < 0 || stage > 2) || (stage == 1 && N > 0 && (!view_sorted || !n_eval)) ||
        (d_mask & ~15) || (e_mask & ~15))
        return mv_fail(-1, "mvsdf_step_backward_inputs: bad arguments");
    StepBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.E = E; a.N = N; a.Nout = Nout; a.Mb = E + N; a.n_true = n_true; a.n_eik = n_eik; a.n_ds = n_ds; a.d_mask = d_mask; a.e_mask = e_mask;
    a.din = din; a.din_ld = din_ld; a.din_feat0 = din_feat0; a.din_nrm0 = din_nrm0; a.use_geo = use_geo;
    a.d_diff = d_diff; a.dx = dx; a.view_sorted = view_sorted; a.n_eval = n_eval; a.true_rows = true_rows;
    a.d_eo = d_eo; a.d_gth = d_gth; a.d_si = d_si; a.dy = dy; a.dn = dn;
    a.no_fbar = stage == 2;                                                    // stage 2 = stage 1 minus SampleNetwork's term (mvsdf_step_backward_fbar adds it)
    hipStream_t s = (hipStream_t)stream;
    if (stage == 0) {
        const size_t total = (size_t)a.Mb * (Nout + 3);
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(k_step_bwd_stage0, dim3(blocks), dim3(256), 0, s, a);
    } else {
        const int total = 3 * N + 2 * (n_eik + 2 * n_ds) + n_true + n_eik;    // upper bound of the work items
        if (total > 0) hipLaunchKernelGGL(k_step_bwd_stage1, dim3((total + 255) / 256), dim3(256), 0, s, a);
    }
    return mv_check(hipGetLastError(), "mvsdf_step_backward_inputs");
}

int mvsdf_step_backward_fbar(int n_eik, int n_ds, int N, int Nout, const float* din, int din_ld, int use_geo, const float* d_diff, const float* dx,
                             const float* view_sorted, const float* n_eval, float* dy, float* fbar, void* stream) {
    if (n_eik < 0 || n_ds < 0 || N <= 0 || !view_sorted || !n_eval || !dy || !fbar) return mv_fail(-1, "mvsdf_step_backward_fbar: bad arguments");
    StepBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.E = n_eik + 2 * n_ds; a.N = N; a.Nout = Nout; a.din = din; a.din_ld = din_ld; a.use_geo = use_geo; a.d_diff = d_diff; a.dx = dx;
    a.view_sorted = view_sorted; a.n_eval = n_eval; a.dy = dy; a.fbar = fbar;
    hipLaunchKernelGGL(k_step_bwd_fbar, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_step_backward_fbar");
}

}  // extern "C"
