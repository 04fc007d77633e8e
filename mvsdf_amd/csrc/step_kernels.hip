// step_kernels.hip -- the bookkeeping of one training step between the big kernels, as a handful of small launches instead of
// dozens of framework ops: row partition (hit rays first), output gathers of IDRNetwork.forward (idr.py:253-304) and the assembly
// of the upstream gradients of the fused SDF backward.  Index / copy work: bit-exact by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "capi_util.h"
#include "det_math.h"
#include "step_internal.h"

// ---------------------------------------------------------------------------------------------------------------
// Stable partition of the rays by "surface" = net_mask & object_mask: perm = [surface rays in ray order | the others in ray order],
// inv = inverse permutation, true_rows = ranks (positions among the surface rays) of the surface rays inside true_mask, in order;
// counts = {#surface, #surface & true}.  view_sorted[r] = -ray_dirs[perm[r]] (the rendering net's view directions, idr.py:300).
// One workgroup walks the rays in chunks of 1024 (R is a few thousand per step).
__global__ __launch_bounds__(1024) void k_partition_rays(const uint8_t* __restrict__ net_mask, const uint8_t* __restrict__ object_mask,
                                                         const uint8_t* __restrict__ true_mask, const float* __restrict__ ray_dirs, int R,
                                                         long long* __restrict__ perm, long long* __restrict__ inv,
                                                         long long* __restrict__ true_rows, long long* __restrict__ counts,
                                                         float* __restrict__ view_sorted, int* __restrict__ true_rank,
                                                         const long long* __restrict__ extra_counts, long long* __restrict__ counts_host,
                                                         long long counts_seq, float* __restrict__ term_rows, int n_eik, int n_ds, int d_mask, int e_mask) {
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
            if (true_rank && hit) true_rank[pos] = tr ? ot + __popcll(bt & below) : -1;
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
    // rows of the three count-normalised loss terms of this batch as floats (eikonal: grad_theta rows, depth: eikonal_output entries, surface: its logits,
    // loss.py:34,61,173): what a data-parallel step all-reduces to normalise by the global counts (IDRLoss.exact_data_parallel) without a host round trip
    if (term_rows && tid == 0) {
        const int N = base[0];
        auto rows = [&](int mask) { return ((mask & 1) ? N : 0) + ((mask & 2) ? n_eik : 0) + ((mask & 4) ? n_ds : 0) + ((mask & 8) ? n_ds : 0); };
        term_rows[0] = (float)rows(e_mask); term_rows[1] = (float)rows(d_mask); term_rows[2] = (float)(base[2] + n_eik);
    }
    if (true_rank && tid < 2) counts[2 + tid] = extra_counts ? extra_counts[tid] : 0;      // (the step driver's 4-entry count record)
    // the same record straight into host-mapped pinned memory (the step driver: no copy node -- a D2H copy between two kernels costs its 4 us plus a
    // ~10 us bubble before the next kernel starts -- and no event either: an event record behind this kernel is a ~6 us bubble of its own).  The host
    // polls entry 4: the forward's sequence number, stored with system-scope release AFTER the four counts.
    if (counts_host && tid == 0) {
        counts_host[0] = base[0]; counts_host[1] = base[2];
        counts_host[2] = extra_counts ? extra_counts[0] : 0; counts_host[3] = extra_counts ? extra_counts[1] : 0;
        __hip_atomic_store(&counts_host[4], counts_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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

// ---------------------------------------------------------------------------------------------------------------
// The upstream of the training step's SDF backward in ONE gather pass (the step driver's route): what k_step_bwd_stage0, two row-block
// copies and k_step_bwd_stage1 (without SampleNetwork's term) produce in four launches --
//   dy[row][c] = (rendering-net feature adjoint on the hit rows, c >= 2) + d(eikonal_output) on column 0 + d(surf_indicator_output) on column 1
//   dn[row]    = (rendering-net normal adjoint on the hit rows, when the geometry is attached) + d(grad_theta)
//   dy_x / dn_x[k] = the rendering-net part alone on hit row k (upstream of the input-adjoint pass X, functional._IdrStep.backward)
// -- with every cell written exactly once: v0 + add, the same single addition the staged kernels perform (0 + add and v0 + 0 are exact).
// Index of an evaluation row inside the concatenation of the groups a mask selects (inverse of mv_group_row), -1 when not selected.
__device__ __forceinline__ int mv_group_index(const StepGroups& g, int row) {
    if (row >= g.E) return (g.mask & 1) ? row - g.E : -1;
    int base = (g.mask & 1) ? g.N : 0;
    if (row < g.n_eik) return (g.mask & 2) ? base + row : -1;
    base += (g.mask & 2) ? g.n_eik : 0;
    if (row < g.n_eik + g.n_ds) return (g.mask & 4) ? base + (row - g.n_eik) : -1;
    base += (g.mask & 4) ? g.n_ds : 0;
    return (g.mask & 8) ? base + (row - g.n_eik - g.n_ds) : -1;
}
struct StepAsmArgs {
    StepBwdArgs b;
    const int* true_rank;                 // [>= N] rank of sorted hit row k among the true-mask hit rows, -1 outside
    float* dy_x; float* dn_x;             // [N][Nout], [N][3]
    const long long* cnt;                 // deferred step (step_internal.h): {N, n_true} on the device; b.N / b.Mb / b.n_true are then upper bounds
};
__global__ void k_step_bwd_assemble(StepAsmArgs s) {
    const StepBwdArgs& a = s.b;
    const int N = s.cnt ? (int)s.cnt[0] : a.N, n_true = s.cnt ? (int)s.cnt[1] : a.n_true;
    const StepGroups gd = {a.n_eik, a.n_ds, a.E, N, a.d_mask}, ge = {a.n_eik, a.n_ds, a.E, N, a.e_mask};
    const int W = a.Nout + 3;
    const size_t total = (size_t)(a.E + N) * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / W), c = (int)(i - (size_t)row * W);
        const int k = row - a.E;
        if (c < a.Nout) {
            float v0 = 0.0f;
            if (k >= 0 && c >= 2 && a.din) v0 = a.din[(size_t)k * a.din_ld + a.din_feat0 + (c - 2)];
            if (k >= 0) s.dy_x[(size_t)k * a.Nout + c] = v0;
            float v = v0;
            if (c == 0 && a.d_eo) {
                const int idx = mv_group_index(gd, row);
                if (idx >= 0) v = v0 + a.d_eo[idx];
            } else if (c == 1 && a.d_si) {
                const int idx = k >= 0 ? s.true_rank[k] : (row < a.n_eik ? n_true + row : -1);
                if (idx >= 0) v = v0 + a.d_si[idx];
            }
            a.dy[(size_t)row * a.Nout + c] = v;
        } else {
            const int cc = c - a.Nout;
            float v0 = 0.0f;
            if (k >= 0 && a.din && a.use_geo && a.din_nrm0 >= 0) v0 = a.din[(size_t)k * a.din_ld + a.din_nrm0 + cc];
            if (k >= 0) s.dn_x[(size_t)k * 3 + cc] = v0;
            float v = v0;
            if (a.d_gth) {
                const int idx = mv_group_index(ge, row);
                if (idx >= 0) v = v0 + a.d_gth[3 * (size_t)idx + cc];
            }
            a.dn[(size_t)row * 3 + cc] = v;
        }
    }
}

int mv_step_backward_assemble(int n_eik, int n_ds, int N, int Nout, int n_true, const float* din, int din_ld, int din_feat0, int din_nrm0, int use_geo,
                              const int* true_rank, const float* d_eo, const float* d_gth, const float* d_si, int d_mask, int e_mask, float* dy,
                              float* dn, float* dy_x, float* dn_x, const long long* cnt, void* stream) {
    const int E = n_eik + 2 * n_ds;
    if (n_eik < 0 || n_ds < 0 || N <= 0 || !dy || !dn || !dy_x || !dn_x || !true_rank || (d_mask & ~15) || (e_mask & ~15))
        return mv_fail(-1, "mv_step_backward_assemble: bad arguments");
    StepAsmArgs s;
    memset(&s, 0, sizeof(s));
    StepBwdArgs& a = s.b;
    a.E = E; a.N = N; a.Nout = Nout; a.Mb = E + N; a.n_true = n_true; a.n_eik = n_eik; a.n_ds = n_ds; a.d_mask = d_mask; a.e_mask = e_mask;
    a.din = din; a.din_ld = din_ld; a.din_feat0 = din_feat0; a.din_nrm0 = din_nrm0; a.use_geo = use_geo;
    a.d_eo = d_eo; a.d_gth = d_gth; a.d_si = d_si; a.dy = dy; a.dn = dn;
    s.true_rank = true_rank; s.dy_x = dy_x; s.dn_x = dn_x; s.cnt = cnt;
    const size_t total = (size_t)a.Mb * (Nout + 3);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(k_step_bwd_assemble, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s);
    return mv_check(hipGetLastError(), "mv_step_backward_assemble");
}

extern "C" {

int mvsdf_partition_rays(const uint8_t* net_mask, const uint8_t* object_mask, const uint8_t* true_mask, const float* ray_dirs, int R,
                         long long* perm, long long* inv, long long* true_rows, long long* counts, float* view_sorted, void* stream) {
    if (!net_mask || !perm || !inv || !true_rows || !counts || R <= 0 || (view_sorted && !ray_dirs))
        return mv_fail(-1, "mvsdf_partition_rays: bad arguments");
    hipLaunchKernelGGL(k_partition_rays, dim3(1), dim3(1024), 0, (hipStream_t)stream, net_mask, object_mask, true_mask, ray_dirs, R, perm, inv,
                       true_rows, counts, view_sorted, (int*)nullptr, (const long long*)nullptr, (long long*)nullptr, 0ll, (float*)nullptr, 0, 0, 0, 0);
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

// the step driver's form (csrc/step_driver.hip): also true_rank[pos] for the sorted hit rows (rank among the true-mask hit rows, -1 outside
// the true mask: the inverse of true_rows, so that the backward's upstream assembly is one gather pass) and counts[2..3] = extra_counts (the
// depth-surface sample counts travelling to the host with the hit counts)
int mv_partition_rays_step(const uint8_t* net_mask, const uint8_t* object_mask, const uint8_t* true_mask, const float* ray_dirs, int R, long long* perm,
                           long long* inv, long long* true_rows, long long* counts, float* view_sorted, int* true_rank, const long long* extra_counts,
                           long long* counts_host, long long counts_seq, float* term_rows, int n_eik, int n_ds, int d_mask, int e_mask, void* stream) {
    if (!net_mask || !perm || !inv || !true_rows || !counts || !true_rank || R <= 0 || (view_sorted && !ray_dirs))
        return mv_fail(-1, "mv_partition_rays_step: bad arguments");
    hipLaunchKernelGGL(k_partition_rays, dim3(1), dim3(1024), 0, (hipStream_t)stream, net_mask, object_mask, true_mask, ray_dirs, R, perm, inv,
                       true_rows, counts, view_sorted, true_rank, extra_counts, counts_host, counts_seq, term_rows, n_eik, n_ds, d_mask, e_mask);
    return mv_check(hipGetLastError(), "mv_partition_rays_step");
}
