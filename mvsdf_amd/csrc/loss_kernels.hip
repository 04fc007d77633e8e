// loss_kernels.hip -- multi-view feature consistency (IDRLoss.get_feat_loss_corr, model/loss.py:115-165) and the
// depth-carving target of get_depth_loss (model/loss.py:37-63 -> utils/my_utils.py:269-331, carving_t2).
//
// k_feat_corr: one wave per hit point.  Each 32-lane half owns the 32 feature channels of one source view at a time:
//   project the point into the reference and the source camera (idx_world2cam / idx_cam2img, my_utils.py:98-110, with
//   their three "+1e-9" divisions), normalise for grid_sample (my_utils.py:152-156, clamp +-1.1), 4-tap bilinear gather
//   (align_corners=False, zeros padding; loss.py:145), cosine correlation, masks, and -- in the same pass -- the analytic
//   gradient of the loss w.r.t. the point (features are constants, scene_dataset.py:139-149), so that backward is a scale.
//   Feature maps are addressed through element strides: NCHW gathers 128 scattered dwords per (point, view); a
//   channels_last tensor (same shape, torch.channels_last) makes every tap one 128-byte line.
// HBM-shaped: 2 * 4 * C * 4 B = 1 KiB of gathered features per (point, source view).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include "capi_util.h"

// Device-side counts of a deferred training step (include/mvsdf_hip.h, MvsdfLossArgs.counts_dev): p = {N, n_true}; the rows of a term are the point groups its
// mask selects among [N hit | n_eik | n_ds | n_ds] (step_kernels.hip, mv_group_total).  p == NULL: the by-value counts of the argument structs are exact.
struct MvDevCounts { const long long* p; int n_eik, n_ds, d_mask, e_mask; };
__device__ __forceinline__ int mv_dc_group_rows(const MvDevCounts& c, int mask) {
    const int N = (int)c.p[0];
    return ((mask & 1) ? N : 0) + ((mask & 2) ? c.n_eik : 0) + ((mask & 4) ? c.n_ds : 0) + ((mask & 8) ? c.n_ds : 0);
}

struct Proj {
    float u, v;          // image coordinates
    float ju[3], jv[3];  // d(u,v)/d(world point)
};

// pts_img = idx_cam2img(idx_world2cam(x)), with forward-mode Jacobian w.r.t. the world point
__device__ __forceinline__ Proj mv_project(const float* __restrict__ cam /*[2][4][4]*/, const float pw[3]) {
    const float* E = cam;
    const float* K = cam + 16;
    float c[4], dc[4][3];
    for (int r = 0; r < 4; ++r) {
        c[r] = E[4 * r] * pw[0] + E[4 * r + 1] * pw[1] + E[4 * r + 2] * pw[2] + E[4 * r + 3];
        for (int j = 0; j < 3; ++j) dc[r][j] = E[4 * r + j];
    }
    const float s1 = 1.0f / (c[3] + 1e-9f);                    // idx_cam_homo / (w + 1e-9)          my_utils.py:101
    float c1[4], dc1[4][3];
    for (int r = 0; r < 4; ++r) {
        c1[r] = c[r] * s1;
        for (int j = 0; j < 3; ++j) dc1[r][j] = dc[r][j] * s1 - c[r] * s1 * s1 * dc[3][j];
    }
    const float s2 = 1.0f / (c1[3] + 1e-9f);                   // [:3] / (w + 1e-9)                   my_utils.py:107
    float c3[3], dc3[3][3];
    for (int r = 0; r < 3; ++r) {
        c3[r] = c1[r] * s2;
        for (int j = 0; j < 3; ++j) dc3[r][j] = dc1[r][j] * s2 - c1[r] * s2 * s2 * dc1[3][j];
    }
    float im[3], dim[3][3];
    for (int r = 0; r < 3; ++r) {                              // K[:3,:3] @ idx_cam                     my_utils.py:108
        im[r] = K[4 * r] * c3[0] + K[4 * r + 1] * c3[1] + K[4 * r + 2] * c3[2];
        for (int j = 0; j < 3; ++j) dim[r][j] = K[4 * r] * dc3[0][j] + K[4 * r + 1] * dc3[1][j] + K[4 * r + 2] * dc3[2][j];
    }
    const float s3 = 1.0f / (im[2] + 1e-9f);                   // / (z + 1e-9)                          my_utils.py:109
    Proj p;
    p.u = im[0] * s3; p.v = im[1] * s3;
    for (int j = 0; j < 3; ++j) {
        p.ju[j] = dim[0][j] * s3 - im[0] * s3 * s3 * dim[2][j];
        p.jv[j] = dim[1][j] * s3 - im[1] * s3 * s3 * dim[2][j];
    }
    return p;
}

struct Sample { float f, fx, fy; };   // value, d/d(gx), d/d(gy) of one channel

// F.grid_sample(bilinear, zeros, align_corners=False) of channel `c` at normalised (gx, gy)
__device__ __forceinline__ Sample mv_bilinear(const float* __restrict__ map, long long sC, long long sH, long long sW, int c, int H, int W,
                                              float gx, float gy) {
    const float ix = ((gx + 1.0f) * W - 1.0f) * 0.5f, iy = ((gy + 1.0f) * H - 1.0f) * 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float* base = map + (long long)c * sC;
    auto tap = [&](int yy, int xx) -> float {
        return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? base[(long long)yy * sH + (long long)xx * sW] : 0.0f;
    };
    const float v00 = tap(y0, x0), v01 = tap(y0, x0 + 1), v10 = tap(y0 + 1, x0), v11 = tap(y0 + 1, x0 + 1);
    Sample s;
    s.f = v00 * (1.f - fx) * (1.f - fy) + v01 * fx * (1.f - fy) + v10 * (1.f - fx) * fy + v11 * fx * fy;
    s.fx = ((v01 - v00) * (1.f - fy) + (v11 - v10) * fy) * (0.5f * W);
    s.fy = ((v10 - v00) * (1.f - fx) + (v11 - v01) * fx) * (0.5f * H);
    return s;
}

__device__ __forceinline__ float half_sum(float v) {           // sum over the 32 lanes of this half-wave
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct FeatArgs {
    const float* pts; int N;
    const int* view_start;     // [B+1] prefix sums of per-view hit counts (device)
    int B, V, C, H, W;
    const float* feat; long long fs[4];        // strides (B, C, H, W) in elements
    const float* feat_src; long long ss[5];    // strides (B, V, C, H, W)
    const float* cam;          // [B][2][4][4]
    const float* src_cams;     // [B][V][2][4][4]
    const float* size;         // [1]
    const float* center;       // [3]
    float* loss_pp;            // [N]   per-point loss, already weighted by 1 / (B * V * m_b)
    float* dpts;               // [N][3] d(total loss)/d(point)
    const long long* cnt;      // deferred step: the points are cnt[0] (N bounds the grid)
};

__global__ __launch_bounds__(256) void k_feat_corr(FeatArgs a) {
    const int lane = threadIdx.x & 63, half = lane >> 5, ch = lane & 31;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= (a.cnt ? (int)a.cnt[0] : a.N)) return;
    int b = 0;
    while (b + 1 < a.B && i >= a.view_start[b + 1]) ++b;
    const int m_b = a.view_start[b + 1] - a.view_start[b];
    const float size = a.size[0];
    float pw[3];
    for (int j = 0; j < 3; ++j) pw[j] = a.pts[3 * (size_t)i + j] / 2.0f * size + a.center[j];     // loss.py:132
    const float Wf = (float)a.W, Hf = (float)a.H;
    // reference view
    const Proj p0 = mv_project(a.cam + (size_t)b * 32, pw);
    const float gx0r = (p0.u / 2.0f) / Wf * 2.0f - 1.0f, gy0r = (p0.v / 2.0f) / Hf * 2.0f - 1.0f;  // loss.py:142, my_utils.py:152-156
    const float gx0 = fminf(fmaxf(gx0r, -1.1f), 1.1f), gy0 = fminf(fmaxf(gy0r, -1.1f), 1.1f);
    const bool in0 = gx0 <= 1.f && gx0 >= -1.f && gy0 <= 1.f && gy0 >= -1.f;
    const float kx0 = (gx0r >= -1.1f && gx0r <= 1.1f) ? 1.0f / Wf : 0.0f;                          // d gx / d u (clamp passes inside)
    const float ky0 = (gy0r >= -1.1f && gy0r <= 1.1f) ? 1.0f / Hf : 0.0f;
    float loss_acc = 0.f, g[3] = {0.f, 0.f, 0.f};
    {                                                                                              // C <= 32 (MVSDF: 32): one channel per lane of a half
        const int c = ch;
        const bool cok = c < a.C;
        Sample s0 = {0.f, 0.f, 0.f};
        if (cok) s0 = mv_bilinear(a.feat + (long long)b * a.fs[0], a.fs[1], a.fs[2], a.fs[3], c, a.H, a.W, gx0, gy0);
        const float n0sq = half_sum(s0.f * s0.f);
        for (int v0 = 0; v0 < a.V; v0 += 2) {
            const int v = v0 + half;
            const bool vok = v < a.V;
            const int vv = vok ? v : 0;
            const Proj pv = mv_project(a.src_cams + ((size_t)b * a.V + vv) * 32, pw);
            const float gxr = (pv.u / 2.0f) / Wf * 2.0f - 1.0f, gyr = (pv.v / 2.0f) / Hf * 2.0f - 1.0f;
            const float gx = fminf(fmaxf(gxr, -1.1f), 1.1f), gy = fminf(fmaxf(gyr, -1.1f), 1.1f);
            const bool inv = gx <= 1.f && gx >= -1.f && gy <= 1.f && gy >= -1.f;
            const float kx = (gxr >= -1.1f && gxr <= 1.1f) ? 1.0f / Wf : 0.0f, ky = (gyr >= -1.1f && gyr <= 1.1f) ? 1.0f / Hf : 0.0f;
            Sample sv = {0.f, 0.f, 0.f};
            if (cok && vok)
                sv = mv_bilinear(a.feat_src + (long long)b * a.ss[0] + (long long)vv * a.ss[1], a.ss[2], a.ss[3], a.ss[4], c, a.H, a.W, gx, gy);
            const float dot = half_sum(s0.f * sv.f), nvsq = half_sum(sv.f * sv.f);
            const float n0 = sqrtf(n0sq), nv = sqrtf(nvsq);
            const float n0c = fmaxf(n0, 1e-9f), nvc = fmaxf(nv, 1e-9f);
            const float corr = dot / n0c / nvc;                                                     // loss.py:149-150
            const float cl = fabsf(1.0f - corr);
            const bool on = vok && in0 && inv && (cl < 0.5f);                                       // loss.py:144,153,155
            // d corr / d f0[c], d corr / d fv[c]   (norm clamp: no gradient through a clamped norm)
            const float a0 = sv.f / (n0c * nvc) - (n0 > 1e-9f ? corr * s0.f / (n0c * n0c) : 0.0f);
            const float av = s0.f / (n0c * nvc) - (nv > 1e-9f ? corr * sv.f / (nvc * nvc) : 0.0f);
            const float d_gx0 = half_sum(a0 * s0.fx), d_gy0 = half_sum(a0 * s0.fy);
            const float d_gxv = half_sum(av * sv.fx), d_gyv = half_sum(av * sv.fy);
            if (on) {
                const float sgn = (1.0f - corr) > 0.f ? -1.0f : ((1.0f - corr) < 0.f ? 1.0f : 0.0f);  // d|1-corr|/dcorr
                loss_acc += cl;
                for (int j = 0; j < 3; ++j) {
                    const float dcorr = d_gx0 * kx0 * p0.ju[j] + d_gy0 * ky0 * p0.jv[j] + d_gxv * kx * pv.ju[j] + d_gyv * ky * pv.jv[j];
                    g[j] += sgn * dcorr;
                }
            }
        }
    }
    // combine the two halves (each handled different source views)
    loss_acc += __shfl_xor(loss_acc, 32);
    for (int j = 0; j < 3; ++j) g[j] += __shfl_xor(g[j], 32);
    if (lane == 0) {
        const float wgt = 1.0f / ((float)a.B * (float)a.V * (float)m_b);                            // mean over [V,1,m,1], then over B
        a.loss_pp[i] = loss_acc * wgt;
        for (int j = 0; j < 3; ++j) a.dpts[3 * (size_t)i + j] = g[j] * wgt * (size / 2.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct CarveArgs {
    const float* pts; int M, pts_ld;  // normalised points [M][pts_ld >= 3]
    float* pts_world;                 // optional [M][pts_ld]: receives the world-space points (may alias pts: loss.py:42 rescales in place)
    const float* depths; int B, h, w; // [B][h][w]
    const float* cams;                // [B][2][4][4]
    const float* size; const float* center;
    float out_thresh_perc, far_thresh, far_att, near_thresh, near_att;
    int use_invalid;                  // conf.use_invalid (loss.py:43-46): carving_t (an in-range view WITHOUT a depth counts as half an 'outside' vote, the in-range mask replaces the valid mask) instead of carving_t2
    float* dist_r; float* weight;     // [M]
};

// carving_t2 / carving_t (my_utils.py:269-331 / 204-266) + the weighting of get_depth_loss (loss.py:42-60) for one point per thread.
struct CarveAcc { float tot_valid, tot_inside, pos_min, neg_max, tot_in; };
// the views v0, v0 + vstep, ... of one point (carving_t2's per-view part, my_utils.py:280-312)
__device__ __forceinline__ CarveAcc mv_carve_views(const CarveArgs& a, const float* pw, int v0, int vstep) {
    const float MAXF = 1e30f / (float)a.B;
    CarveAcc r = {0.f, 0.f, INFINITY, -INFINITY, 0.f};
    for (int v = v0; v < a.B; v += vstep) {
        const float* E = a.cams + (size_t)v * 32;
        const float* K = E + 16;
        float c[4];
        for (int q = 0; q < 4; ++q) c[q] = E[4 * q] * pw[0] + E[4 * q + 1] * pw[1] + E[4 * q + 2] * pw[2] + E[4 * q + 3];
        const float s1 = c[3] + 1e-9f;
        for (int q = 0; q < 4; ++q) c[q] = c[q] / s1;
        const float pdepth = c[2];                                                                // my_utils.py:296
        const float s2 = c[3] + 1e-9f;
        const float c3[3] = {c[0] / s2, c[1] / s2, c[2] / s2};
        float im[3];
        for (int q = 0; q < 3; ++q) im[q] = K[4 * q] * c3[0] + K[4 * q + 1] * c3[1] + K[4 * q + 2] * c3[2];
        const float u = im[0] / (im[2] + 1e-9f), vv = im[1] / (im[2] + 1e-9f);
        const float gx = fminf(fmaxf(u / (float)a.w * 2.0f - 1.0f, -1.1f), 1.1f);
        const float gy = fminf(fmaxf(vv / (float)a.h * 2.0f - 1.0f, -1.1f), 1.1f);
        const bool in_range = gx <= 1.f && gx >= -1.f && gy <= 1.f && gy >= -1.f;
        const int ix = (int)nearbyintf(((gx + 1.0f) * a.w - 1.0f) * 0.5f), iy = (int)nearbyintf(((gy + 1.0f) * a.h - 1.0f) * 0.5f);
        float gd = 0.0f;                                                                          // grid_sample nearest, zeros padding
        if (ix >= 0 && ix < a.w && iy >= 0 && iy < a.h) gd = a.depths[((size_t)v * a.h + iy) * a.w + ix];
        const bool valid = (gd > 0.f) && in_range;
        const bool inside = (pdepth > gd * 0.99f) && valid;
        const bool outside = valid != inside;
        const float dist = valid ? (pdepth - gd) : 0.0f;
        r.tot_valid += valid; r.tot_inside += inside; r.tot_in += in_range;                       // counts of 0 / 1: exact in any order
        r.pos_min = fminf(r.pos_min, inside ? dist : MAXF);
        r.neg_max = fmaxf(r.neg_max, outside ? dist : -MAXF);
    }
    return r;
}
__device__ __forceinline__ void mv_carve_world(const CarveArgs& a, int i, float* pw) {
    const float size = a.size[0];
    for (int j = 0; j < 3; ++j) pw[j] = a.pts[(size_t)a.pts_ld * i + j] / 2.0f * size + a.center[j];       // loss.py:42
}
// the voting over the views + the weighting of get_depth_loss (my_utils.py:313-331, loss.py:42-60)
__device__ __forceinline__ void mv_carve_finish(const CarveArgs& a, int i, const float* pw, const CarveAcc& r) {
    const float size = a.size[0];
    if (a.pts_world) for (int j = 0; j < 3; ++j) a.pts_world[(size_t)a.pts_ld * i + j] = pw[j];
    const float MAXF = 1e30f / (float)a.B;
    const float tot_valid = r.tot_valid, tot_inside = r.tot_inside, pos_min = r.pos_min, neg_max = r.neg_max;
    auto agg = [&](float res, float sign) {                                                       // RunningTopK(k=1).aggregate, my_utils.py:190-201
        const bool validm = fabsf(res) < MAXF * .99f;
        const float num = validm ? 1.f : 0.f;
        const float ret = (validm ? res : 0.f) / (num + 1e-9f);
        return ret * (num > 0.5f ? 1.f : 0.f) + MAXF * sign * (num < 0.5f ? 1.f : 0.f);
    };
    const float dpos = agg(pos_min, 1.f), dneg = agg(neg_max, -1.f);
    // carving_t2: votes among the views that HAVE a depth at the point's pixel (my_utils.py:321-328); carving_t (conf.use_invalid): among the views that see the
    // point at all, one without a depth there counting as half an 'outside' vote (my_utils.py:256-263) -- and the in-range mask takes the valid mask's place
    const float tot_in = r.tot_in;
    const float outside_perc = a.use_invalid ? ((tot_valid - tot_inside) + (tot_in - tot_valid) * 0.5f) / (tot_in + 1e-9f) : (tot_valid - tot_inside) / (tot_valid + 1e-9f);
    const bool scene_valid = a.use_invalid ? (tot_in > 0.f) : (tot_valid > 0.f);
    const bool scene_outside = (outside_perc > a.out_thresh_perc) && scene_valid;
    const bool scene_inside = scene_valid != scene_outside;
    const float dist = dpos * (scene_inside ? 1.f : 0.f) + dneg * (scene_outside ? 1.f : 0.f);
    float dr = dist / size * 2.0f + (-1.25f) * (scene_valid ? 0.f : 1.f);                          // loss.py:47
    dr = fminf(fmaxf(dr, -1.25f), 1.25f);
    const bool far = fabsf(dr) > a.far_thresh, near = fabsf(dr) < a.near_thresh;
    const float fw = far ? a.far_att : 1.0f, nw = near ? a.near_att : 1.0f;
    a.dist_r[i] = dr;
    a.weight[i] = fw * nw * (scene_valid ? 1.f : 0.f);
}
// carving_t2 (my_utils.py:269-331) + the weighting of get_depth_loss (loss.py:42-60) for one point per thread.
__device__ __forceinline__ void mv_carve_point(const CarveArgs& a, int i) {
    float pw[3];
    mv_carve_world(a, i, pw);
    const CarveAcc r = mv_carve_views(a, pw, 0, 1);
    mv_carve_finish(a, i, pw, r);
}
__global__ void k_carve(CarveArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.M) mv_carve_point(a, i);
}

// ---------------------------------------------------------------------------------------------------------------
// The elementwise terms of IDRLoss.forward (model/loss.py:21-35, 58-61, 167-174, 206-210) in ONE single-workgroup launch:
// rgb L1, eikonal, weighted depth L1, surface BCE-with-logits, the feature-loss sum and the weighted total -- plus the
// unit gradients of each term w.r.t. its network input, so that backward is four scalings.  Reductions are LDS trees in a fixed
// order (deterministic).  ~20 k elements at 2048 rays: one workgroup is enough and keeps it to a single launch.
struct LossArgs {
    const float* rgb; const float* rgb_gt; const uint8_t* rgb_mask; int R;          // rgb_values[R][3], gt[R][3], mask[R]
    const float* grad_theta; int n_eik;                                              // [n_eik][3]
    const float* eik_out; const float* dist_r; const float* dweight; int n_depth;    // [n_depth]
    const float* surf; int n_surf; const long long* n_pos;                           // logits[n_surf], targets = (i < *n_pos)
    const float* feat_pp; int n_feat;                                                // per-point feature-loss terms (may be null)
    float w_rgb, w_eik, w_surf, w_feat, w_depth; int surf_on, feat_on;
    float smooth;                                                                    // depth term: 0 = L1 (loss.py:60), s > 0 = SmoothL1(eo / s, -dist_r / s) * s (loss.py:57-58, conf.smooth)
    const float* inv_counts;                                                         // optional [3]: 1/count of the eikonal / depth / surf means (data-parallel exact mode)
    float* out;                                                                      // [6]: loss, rgb, eikonal, depth, feat, surf
    float* d_rgb; float* d_grad; float* d_eik_out; float* d_surf;                    // unit gradients (same shapes as the inputs)
    // optional: the gradients of the TOTAL loss (unit gradient x the term's weight: what k_loss_scale produces for an upstream of exactly 1 on `loss`
    // alone -- 1 * w + 0 is w, so the values are bit-identical) and of the feature term's points; lets `loss.backward()` skip that launch
    float* s_rgb; float* s_grad; float* s_eik_out; float* s_surf; const float* dpts; float* s_diff; int n_dpts;
    // launch over MV_LOSS_SLICES workgroups (mvsdf_loss_forward): partial[MV_LOSS_SLICES][8] and a ticket counter zeroed before the launch; both null: one workgroup
    float* partial; unsigned* ticket;
    MvDevCounts dc;                                                                  // dc.p set (deferred step): n_eik / n_depth / n_surf / n_feat / n_dpts above are upper bounds
};

// sum over the 1024 threads of the workgroup, the same value in every thread; fixed order (butterfly inside a wave, then the 16 wave sums in
// index order): deterministic.  Two barriers instead of the eleven of an LDS tree -- the kernel is five of these in a row.
__device__ float block_sum_1024(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.0f;
    for (int k = 0; k < 16; ++k) r += red[k];
    __syncthreads();
    return r;
}

// The five sums are defined over a FIXED partition of the index space into MV_LOSS_SLICES interleaved slices (slice s owns i = s * 1024 + tid + k *
// MV_LOSS_SLICES * the term's wave count * 64): per slice the sum over the term's waves, then the slice sums in index order.  One workgroup walks all slices
// (mvsdf_loss_terms), or MV_LOSS_SLICES workgroups take one each and the last one to finish adds them up (mvsdf_loss_forward: 15-25 us of one CU
// chasing loads -> a few us) -- the same bits either way.
#define MV_LOSS_SLICES 16
__global__ __launch_bounds__(1024) void k_loss_terms(LossArgs a) {
    __shared__ float red[1024];
    __shared__ float s_part[MV_LOSS_SLICES][5];
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, G = gridDim.x, STR = MV_LOSS_SLICES * 1024;
    // row counts of the terms: by value, or (deferred step) derived from the device-side {N, n_true}
    const bool dev = a.dc.p != nullptr;
    const int n_eik = dev ? min(a.n_eik, mv_dc_group_rows(a.dc, a.dc.e_mask)) : a.n_eik;
    const int n_depth = dev ? min(a.n_depth, mv_dc_group_rows(a.dc, a.dc.d_mask)) : a.n_depth;
    const int n_surf = dev ? min(a.n_surf, (int)a.dc.p[1] + a.dc.n_eik) : a.n_surf;
    const int n_feat = dev ? min(a.n_feat, (int)a.dc.p[0]) : a.n_feat;
    const int n_dpts = dev ? min(a.n_dpts, 3 * (int)a.dc.p[0]) : a.n_dpts;
    const float invR = 1.0f / (float)a.R;
    const float invE = n_eik > 0 ? (a.inv_counts ? a.inv_counts[0] : 1.0f / (float)n_eik) : 0.f;
    const float invD = n_depth > 0 ? (a.inv_counts ? a.inv_counts[1] : 1.0f / (float)n_depth) : 0.f;
    const float invS = n_surf > 0 ? (a.inv_counts ? a.inv_counts[2] : 1.0f / (float)n_surf) : 0.f;
    const long long npos = a.surf_on ? *a.n_pos : 0;
    // the five terms run side by side on disjoint groups of the 16 waves (rgb 6, eikonal 4, depth 3, surface 2, feature 1): their loads are
    // in flight together instead of one term after the other behind a workgroup sum each (5 x ~3 us of load latency)
    const int w = tid >> 6, lane = tid & 63;
    for (int slice = blockIdx.x; slice < MV_LOSS_SLICES; slice += G) {
        float s = 0.f;                                            // this thread's partial of ITS term
        if (w < 6) {
            // rgb: L1Loss(reduction='sum')(rgb[mask], gt[mask]) / R                                  loss.py:21-28
            for (int i = slice * 384 + w * 64 + lane; i < a.R * 3; i += MV_LOSS_SLICES * 384) {
                const bool m = a.rgb_mask[i / 3] != 0;
                const float df = a.rgb[i] - a.rgb_gt[i];
                if (m) s += fabsf(df);
                const float dr = m ? (df > 0.f ? invR : (df < 0.f ? -invR : 0.f)) : 0.f;
                a.d_rgb[i] = dr;
                if (a.s_rgb) a.s_rgb[i] = dr * a.w_rgb;
            }
        } else if (w < 10) {
            // eikonal: mean((||g|| - 1)^2)                                                           loss.py:30-35
            for (int i = slice * 256 + (w - 6) * 64 + lane; i < n_eik; i += MV_LOSS_SLICES * 256) {
                const float gx = a.grad_theta[3 * i], gy = a.grad_theta[3 * i + 1], gz = a.grad_theta[3 * i + 2];
                const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
                const float e = nrm - 1.0f;
                s += e * e;
                const float k = nrm > 0.f ? 2.0f * e / nrm * invE : 0.f;
                a.d_grad[3 * i] = k * gx; a.d_grad[3 * i + 1] = k * gy; a.d_grad[3 * i + 2] = k * gz;
                if (a.s_grad) { a.s_grad[3 * i] = (k * gx) * a.w_eik; a.s_grad[3 * i + 1] = (k * gy) * a.w_eik; a.s_grad[3 * i + 2] = (k * gz) * a.w_eik; }
            }
        } else if (w < 13) {
            // depth: mean(|eikonal_output + dist_r| * weight); with conf.smooth = s: SmoothL1(eo / s, -dist_r / s) * s (beta = 1)      loss.py:57-61
            for (int i = slice * 192 + (w - 10) * 64 + lane; i < n_depth; i += MV_LOSS_SLICES * 192) {
                const float df = a.eik_out[i] + a.dist_r[i], wgt = a.dweight[i];
                float el = fabsf(df), gu = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);           // term and its derivative w.r.t. eikonal_output
                if (a.smooth > 0.f) {
                    const float x = a.eik_out[i] / a.smooth - (-a.dist_r[i] / a.smooth), ax = fabsf(x);
                    el = (ax < 1.f ? 0.5f * x * x : ax - 0.5f) * a.smooth;
                    gu = ax < 1.f ? x : (x > 0.f ? 1.f : -1.f);                                    // (d/d eo of s * h(eo / s) = h')
                }
                s += el * wgt;
                const float de = gu * wgt * invD;
                a.d_eik_out[i] = de;
                if (a.s_eik_out) a.s_eik_out[i] = de * a.w_depth;
            }
        } else if (w < 15) {
            // surface indicator: BCEWithLogits(mean) against [1]*n_pos + [0]*rest                    loss.py:167-174
            for (int i = slice * 128 + (w - 13) * 64 + lane; i < n_surf; i += MV_LOSS_SLICES * 128) {
                if (a.surf_on) {
                    const float x = a.surf[i], t = (long long)i < npos ? 1.0f : 0.0f;
                    s += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
                    const float ds = (1.0f / (1.0f + expf(-x)) - t) * invS;
                    a.d_surf[i] = ds;
                    if (a.s_surf) a.s_surf[i] = ds * a.w_surf;
                } else { a.d_surf[i] = 0.f; if (a.s_surf) a.s_surf[i] = 0.f; }
            }
        } else {
            // feature consistency: sum of the per-point terms of k_feat_corr
            if (a.feat_on && a.feat_pp) for (int i = slice * 64 + lane; i < n_feat; i += MV_LOSS_SLICES * 64) s += a.feat_pp[i];
        }
        if (a.s_diff && a.dpts) for (int i = slice * 1024 + tid; i < n_dpts; i += STR) a.s_diff[i] = a.dpts[i] * a.w_feat;
        // the five sums of this slice: butterfly inside every wave, then the waves of each term in index order
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[w] = s;
        __syncthreads();
        if (tid == 0) {
            const float p_rgb = ((((red[0] + red[1]) + red[2]) + red[3]) + red[4]) + red[5], p_eik = ((red[6] + red[7]) + red[8]) + red[9];
            const float p_depth = (red[10] + red[11]) + red[12], p_surf = red[13] + red[14], p_feat = red[15];
            if (G > 1) {                                          // device-coherent stores: no release fence (= a write-back of this XCD's whole L2) needed
                float* pp = a.partial + 8 * slice;
                const float pv[5] = {p_rgb, p_eik, p_depth, p_surf, p_feat};
                for (int t = 0; t < 5; ++t) __hip_atomic_store(pp + t, pv[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else { s_part[slice][0] = p_rgb; s_part[slice][1] = p_eik; s_part[slice][2] = p_depth; s_part[slice][3] = p_surf; s_part[slice][4] = p_feat; }
        }
        __syncthreads();
    }
    if (G > 1) {                                                  // the last workgroup to arrive adds the slices up
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's slice sums have reached the coherence point before its ticket
            s_last = (__hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) {                                           // cache-bypassing loads of everybody's slice sums
            for (int sl = 0; sl < MV_LOSS_SLICES; ++sl)
                for (int t = 0; t < 5; ++t) s_part[sl][t] = __hip_atomic_load(a.partial + 8 * sl + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        float t5[5];
        for (int t = 0; t < 5; ++t) { float r = 0.f; for (int sl = 0; sl < MV_LOSS_SLICES; ++sl) r += s_part[sl][t]; t5[t] = r; }
        const float rgb_loss = t5[0] * invR, eik_loss = t5[1] * invE, depth_loss = t5[2] * invD, surf_loss = a.surf_on ? t5[3] * invS : 0.f;
        const float feat_loss = (a.feat_on && a.feat_pp) ? t5[4] : 0.f;
        a.out[1] = rgb_loss; a.out[2] = eik_loss; a.out[3] = depth_loss; a.out[4] = feat_loss; a.out[5] = surf_loss;
        a.out[0] = rgb_loss * a.w_rgb + eik_loss * a.w_eik + surf_loss * a.w_surf + feat_loss * a.w_feat + depth_loss * a.w_depth;   // loss.py:206-210
    }
}

extern "C" {

/* IDRLoss.forward's elementwise terms + weighted total (loss.py:21-35, 58-61, 167-174, 206-210) and their unit gradients.
 * out[6] = {loss, rgb_loss, eikonal_loss, depth_loss, feat_loss, surf_loss}. */
int mvsdf_loss_terms(const float* rgb, const float* rgb_gt, const uint8_t* rgb_mask, int R, const float* grad_theta, int n_eik,
                     const float* eik_out, const float* dist_r, const float* dweight, int n_depth, const float* surf, int n_surf,
                     const long long* n_pos, const float* feat_pp, int n_feat, float w_rgb, float w_eik, float w_surf, float w_feat,
                     float w_depth, float smooth, int surf_on, int feat_on, const float* inv_counts, float* out, float* d_rgb, float* d_grad, float* d_eik_out,
                     float* d_surf, void* stream) {
    if (!rgb || !rgb_gt || !rgb_mask || R <= 0 || !out || !d_rgb || (n_eik > 0 && (!grad_theta || !d_grad)) ||
        (n_depth > 0 && (!eik_out || !dist_r || !dweight || !d_eik_out)) || (n_surf > 0 && (!surf || !d_surf || !n_pos)))
        return mv_fail(-1, "mvsdf_loss_terms: bad arguments");
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.inv_counts = inv_counts;
    a.rgb = rgb; a.rgb_gt = rgb_gt; a.rgb_mask = rgb_mask; a.R = R; a.grad_theta = grad_theta; a.n_eik = n_eik;
    a.eik_out = eik_out; a.dist_r = dist_r; a.dweight = dweight; a.n_depth = n_depth; a.surf = surf; a.n_surf = n_surf; a.n_pos = n_pos;
    a.feat_pp = feat_pp; a.n_feat = n_feat; a.w_rgb = w_rgb; a.w_eik = w_eik; a.w_surf = w_surf; a.w_feat = w_feat; a.w_depth = w_depth;
    a.smooth = smooth > 0.f ? smooth : 0.f;
    a.surf_on = surf_on; a.feat_on = feat_on; a.out = out; a.d_rgb = d_rgb; a.d_grad = d_grad; a.d_eik_out = d_eik_out; a.d_surf = d_surf;
    hipLaunchKernelGGL(k_loss_terms, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_loss_terms");
}

/* IDRLoss.get_feat_loss_corr (loss.py:115-165) forward + analytic d/d(points) in one launch.
 * pts[N][3]: diff_surf_pts (hit points, view-major); view_start[B+1] (device int32): prefix sums of per-view hit counts.
 * feat[B][C][H][W] / feat_src[B][V][C][H][W] addressed by element strides (NCHW or channels_last).
 * cam[B][2][4][4], src_cams[B][V][2][4][4], size[1], center[3] (device).  Outputs: loss_pp[N] (sum = the loss), dpts[N][3]. */
static int mv_feat_corr_cnt(const float* pts, int N, const int* view_start, int B, int V, int C, int H, int W, const float* feat,
                            const long long* feat_strides, const float* feat_src, const long long* src_strides, const float* cam,
                            const float* src_cams, const float* size, const float* center, float* loss_pp, float* dpts, const long long* cnt, void* stream);
int mvsdf_feat_corr(const float* pts, int N, const int* view_start, int B, int V, int C, int H, int W, const float* feat,
                    const long long* feat_strides, const float* feat_src, const long long* src_strides, const float* cam,
                    const float* src_cams, const float* size, const float* center, float* loss_pp, float* dpts, void* stream) {
    return mv_feat_corr_cnt(pts, N, view_start, B, V, C, H, W, feat, feat_strides, feat_src, src_strides, cam, src_cams, size, center, loss_pp, dpts, nullptr, stream);
}
// cnt (device, optional): the points are cnt[0]; N then bounds the grid (the deferred step)
static int mv_feat_corr_cnt(const float* pts, int N, const int* view_start, int B, int V, int C, int H, int W, const float* feat,
                            const long long* feat_strides, const float* feat_src, const long long* src_strides, const float* cam,
                            const float* src_cams, const float* size, const float* center, float* loss_pp, float* dpts, const long long* cnt, void* stream) {
    if (!pts || !view_start || !feat || !feat_src || !cam || !src_cams || !size || !center || !loss_pp || !dpts || !feat_strides || !src_strides)
        return mv_fail(-1, "mvsdf_feat_corr: null argument");
    if (N <= 0 || B <= 0 || V <= 0 || H <= 0 || W <= 0) return mv_fail(-1, "mvsdf_feat_corr: bad sizes");
    if (C <= 0 || C > 32) return mv_fail(-1, "mvsdf_feat_corr: C must be in 1..32 (MVSDF features have 32 channels)");
    FeatArgs a;
    a.pts = pts; a.N = N; a.view_start = view_start; a.B = B; a.V = V; a.C = C; a.H = H; a.W = W;
    a.feat = feat; a.feat_src = feat_src;
    for (int i = 0; i < 4; ++i) a.fs[i] = feat_strides[i];
    for (int i = 0; i < 5; ++i) a.ss[i] = src_strides[i];
    a.cam = cam; a.src_cams = src_cams; a.size = size; a.center = center; a.loss_pp = loss_pp; a.dpts = dpts; a.cnt = cnt;
    hipLaunchKernelGGL(k_feat_corr, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_feat_corr");
}

/* Depth-carving target of IDRLoss.get_depth_loss (loss.py:37-63, carving_t2): pts[M][3] normalised sample points,
 * depths[B][h][w], cams[B][2][4][4] -> dist_r[M], weight[M];  loss = mean(|eikonal_output + dist_r| * weight). */
int mvsdf_depth_carve(const float* pts, int pts_ld, int M, const float* depths, int B, int h, int w, const float* cams, const float* size,
                      const float* center, float out_thresh_perc, float far_thresh, float far_att, float near_thresh, float near_att, int use_invalid,
                      float* dist_r, float* weight, float* pts_world, void* stream) {
    if (!pts || pts_ld < 3 || !depths || !cams || !size || !center || !dist_r || !weight || M <= 0 || B <= 0 || h <= 0 || w <= 0)
        return mv_fail(-1, "mvsdf_depth_carve: bad arguments");
    CarveArgs a;
    a.pts = pts; a.M = M; a.pts_ld = pts_ld; a.pts_world = pts_world; a.depths = depths; a.B = B; a.h = h; a.w = w; a.cams = cams; a.size = size; a.center = center;
    a.out_thresh_perc = out_thresh_perc; a.far_thresh = far_thresh; a.far_att = far_att; a.near_thresh = near_thresh; a.near_att = near_att;
    a.use_invalid = use_invalid ? 1 : 0;
    a.dist_r = dist_r; a.weight = weight;
    hipLaunchKernelGGL(k_carve, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_depth_carve");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Bookkeeping of IDRLoss.forward / backward as two launches instead of ~25 framework ops.
//   k_loss_prep : hit = network_object_mask & object_mask (loss.py:21,206); per-view hit counts -> prefix sums view_start[B+1]
//                 (which rows of diff_surf_pts belong to which view, loss.py:119-127); n_pos = #(network_object_mask & object_mask_true)
//                 (the positives of the surface-indicator BCE, loss.py:167-173).
//   k_loss_scale: backward of the weighted total: every stored unit gradient times (dL/dloss * weight + dL/dterm).
__device__ __forceinline__ void mv_loss_prep_block(const uint8_t* __restrict__ net_mask, const uint8_t* __restrict__ obj_mask,
                                                   const uint8_t* __restrict__ true_mask, int R, int B, uint8_t* __restrict__ hit,
                                                   int* __restrict__ view_start, long long* __restrict__ n_pos) {
    // one wave per view (views beyond 16 take turns); the prefix sum over the B counts is done by thread 0
    __shared__ int cnt[1024];
    __shared__ int pos[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, P = R / B;
    int pos_local = 0;
    for (int b = w; b < B; b += 16) {
        int c = 0;
        for (int i = lane; i < P; i += 64) {
            const int r = b * P + i;
            const bool h = net_mask[r] && obj_mask[r];
            hit[r] = h ? 1 : 0;
            c += h ? 1 : 0;
            pos_local += (net_mask[r] && true_mask[r]) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0 && b < 1024) cnt[b] = c;
    }
    for (int o = 32; o > 0; o >>= 1) pos_local += __shfl_xor(pos_local, o);
    if (lane == 0) pos[w] = pos_local;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        view_start[0] = 0;
        for (int b = 0; b < B; ++b) { run += cnt[b]; view_start[b + 1] = run; }
        long long s = 0;
        for (int k = 0; k < 16; ++k) s += pos[k];
        *n_pos = s;
    }
}
__global__ __launch_bounds__(1024) void k_loss_prep(const uint8_t* __restrict__ net_mask, const uint8_t* __restrict__ obj_mask,
                                                    const uint8_t* __restrict__ true_mask, int R, int B, uint8_t* __restrict__ hit,
                                                    int* __restrict__ view_start, long long* __restrict__ n_pos) {
    mv_loss_prep_block(net_mask, obj_mask, true_mask, R, B, hit, view_start, n_pos);
}
// mask bookkeeping (workgroup 0) and depth carving (the other workgroups, 1024 points each) in one launch: independent work of IDRLoss.forward
struct PrepCarveArgs {
    const uint8_t* net_mask; const uint8_t* obj_mask; const uint8_t* true_mask; int R, B; uint8_t* hit; int* view_start; long long* n_pos;
    unsigned* ticket;                     // zeroed here for k_loss_terms' last-workgroup-done count
    CarveArgs c;
    MvDevCounts dc;                       // dc.p set (deferred step): c.M is an upper bound, the points are the rows of the groups dc.d_mask selects
};
// carve workgroups: 64 points each, the 16 waves split the views (wave w takes views w, w + 16, ...): the per-view counts and min / max combine
// exactly in any order, so the result equals the one-thread-per-point kernel's bit for bit at a sixteenth of its latency
__global__ __launch_bounds__(1024) void k_loss_prep_carve(PrepCarveArgs a) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0 && a.ticket) *a.ticket = 0u;
        mv_loss_prep_block(a.net_mask, a.obj_mask, a.true_mask, a.R, a.B, a.hit, a.view_start, a.n_pos);
        return;
    }
    __shared__ float part[16][5][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = (blockIdx.x - 1) * 64 + lane;
    const int Mc = a.dc.p ? min(a.c.M, mv_dc_group_rows(a.dc, a.dc.d_mask)) : a.c.M;
    if ((int)(blockIdx.x - 1) * 64 >= Mc) return;                 // (workgroup-uniform)
    const bool in = i < Mc;
    float pw[3] = {0.f, 0.f, 0.f};
    CarveAcc r = {0.f, 0.f, INFINITY, -INFINITY, 0.f};
    if (in) { mv_carve_world(a.c, i, pw); r = mv_carve_views(a.c, pw, w, 16); }
    part[w][0][lane] = r.tot_valid; part[w][1][lane] = r.tot_inside; part[w][2][lane] = r.pos_min; part[w][3][lane] = r.neg_max; part[w][4][lane] = r.tot_in;
    __syncthreads();
    if (w == 0 && in) {
        CarveAcc t = r;
        for (int k = 1; k < 16; ++k) {
            t.tot_valid += part[k][0][lane]; t.tot_inside += part[k][1][lane]; t.tot_in += part[k][4][lane];
            t.pos_min = fminf(t.pos_min, part[k][2][lane]); t.neg_max = fmaxf(t.neg_max, part[k][3][lane]);
        }
        mv_carve_finish(a.c, i, pw, t);
    }
}

struct LossScaleArgs {
    const float* g[6];                    // upstream of {loss, rgb, eikonal, depth, feat, surf}: one scalar each, null = 0
    float w_rgb, w_eik, w_surf, w_feat, w_depth;
    const float* src[5]; float* dst[5]; int n[5];      // unit gradients of rgb / grad_theta / eikonal_output / surf / diff_surf_pts (k_feat_corr's dpts) -> scaled copies
    float* coef_feat;                     // [1]: dL/d(sum of the per-point feature terms)
    MvDevCounts dc;                       // dc.p set (deferred step): n[1..4] are upper bounds
};
__global__ void k_loss_scale(LossScaleArgs a) {
    float g[6];
    for (int k = 0; k < 6; ++k) g[k] = a.g[k] ? a.g[k][0] : 0.0f;
    const float g0 = g[0];
    const float c[5] = {g0 * a.w_rgb + g[1], g0 * a.w_eik + g[2], g0 * a.w_depth + g[3], g0 * a.w_surf + g[5], g0 * a.w_feat + g[4]};
    int n[5] = {a.n[0], a.n[1], a.n[2], a.n[3], a.n[4]};
    if (a.dc.p) {
        n[1] = min(n[1], 3 * mv_dc_group_rows(a.dc, a.dc.e_mask)); n[2] = min(n[2], mv_dc_group_rows(a.dc, a.dc.d_mask));
        n[3] = min(n[3], (int)a.dc.p[1] + a.dc.n_eik); n[4] = min(n[4], 3 * (int)a.dc.p[0]);
    }
    const int total = n[0] + n[1] + n[2] + n[3] + n[4];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int k = i, t = 0;
        while (t < 4 && k >= n[t]) { k -= n[t]; ++t; }
        a.dst[t][k] = a.src[t][k] * c[t];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.coef_feat) a.coef_feat[0] = g0 * a.w_feat + g[4];
}

extern "C" {

int mvsdf_loss_prep(const uint8_t* net_mask, const uint8_t* obj_mask, const uint8_t* true_mask, int R, int B, uint8_t* hit, int* view_start,
                    long long* n_pos, void* stream) {
    if (!net_mask || !obj_mask || !true_mask || !hit || !view_start || !n_pos || R <= 0 || B <= 0 || B > 1024 || R % B)
        return mv_fail(-1, "mvsdf_loss_prep: bad arguments");
    hipLaunchKernelGGL(k_loss_prep, dim3(1), dim3(1024), 0, (hipStream_t)stream, net_mask, obj_mask, true_mask, R, B, hit, view_start, n_pos);
    return mv_check(hipGetLastError(), "mvsdf_loss_prep");
}

int mvsdf_loss_scale(const float* const* g, float w_rgb, float w_eik, float w_surf, float w_feat, float w_depth, const float* d_rgb, float* g_rgb,
                     int n_rgb, const float* d_grad, float* g_grad, int n_grad, const float* d_eo, float* g_eo, int n_eo, const float* d_sf,
                     float* g_sf, int n_sf, float* coef_feat, void* stream) {
    if (!g || n_rgb < 0 || n_grad < 0 || n_eo < 0 || n_sf < 0) return mv_fail(-1, "mvsdf_loss_scale: bad arguments");
    LossScaleArgs a;
    for (int k = 0; k < 6; ++k) a.g[k] = g[k];
    a.w_rgb = w_rgb; a.w_eik = w_eik; a.w_surf = w_surf; a.w_feat = w_feat; a.w_depth = w_depth;
    const float* src[4] = {d_rgb, d_grad, d_eo, d_sf};
    float* dst[4] = {g_rgb, g_grad, g_eo, g_sf};
    const int n[4] = {n_rgb, n_grad, n_eo, n_sf};
    int total = 0;
    for (int t = 0; t < 4; ++t) {
        if (n[t] > 0 && (!src[t] || !dst[t])) return mv_fail(-1, "mvsdf_loss_scale: null tensor with a positive count");
        a.src[t] = src[t]; a.dst[t] = dst[t]; a.n[t] = n[t]; total += n[t];
    }
    a.src[4] = nullptr; a.dst[4] = nullptr; a.n[4] = 0;
    a.coef_feat = coef_feat;
    memset(&a.dc, 0, sizeof(a.dc));
    const int blocks = total > 0 ? (total + 255) / 256 : 1;
    hipLaunchKernelGGL(k_loss_scale, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_loss_scale");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// IDRLoss.forward / backward as one C call each (include/mvsdf_hip.h, "native step driver"): the same kernels as above, enqueued from C++.
static inline size_t mv_al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

int mvsdf_loss_layout(const MvsdfLossArgs* a, MvsdfLossLayout* lo) {
    if (!a || !lo || a->R <= 0 || a->B <= 0 || a->N < 0 || a->n_grad < 0 || a->n_depth < 0 || a->n_surf < 0) return mv_fail(-1, "mvsdf_loss_layout: bad arguments");
    size_t p = 0;
    auto take = [&](size_t bytes) { const size_t o = p; p += mv_al256(bytes ? bytes : 4); return o; };
    lo->out = take(6 * 4);
    lo->hit = take((size_t)a->R);
    lo->view_start = take(((size_t)a->B + 1) * 4);
    lo->n_pos = take(8);
    lo->loss_pp = take((size_t)a->N * 4);
    lo->dpts = take((size_t)a->N * 12);
    lo->dist_r = take((size_t)a->n_depth * 4);
    lo->weight = take((size_t)a->n_depth * 4);
    lo->d_rgb = take((size_t)a->R * 12);
    lo->d_grad = take((size_t)a->n_grad * 12);
    lo->d_eo = take((size_t)a->n_depth * 4);
    lo->d_sf = take((size_t)a->n_surf * 4);
    lo->s_rgb = take((size_t)a->R * 12);
    lo->s_grad = take((size_t)a->n_grad * 12);
    lo->s_eo = take((size_t)a->n_depth * 4);
    lo->s_sf = take((size_t)a->n_surf * 4);
    lo->s_diff = take((size_t)a->N * 12);
    take(MV_LOSS_SLICES * 8 * 4 + 16);                            // scratch of k_loss_terms (slice sums + ticket): loss_scratch() below
    lo->bytes = p;
    return 0;
}
// the scratch region behind s_diff (not part of the public layout)
static size_t loss_scratch(const MvsdfLossArgs* a, const MvsdfLossLayout& lo) { return lo.s_diff + mv_al256((size_t)a->N * 12 ? (size_t)a->N * 12 : 4); }

int mvsdf_loss_forward(const MvsdfLossArgs* a, void* blk, void* stream) {
    MvsdfLossLayout lo;
    int rc = mvsdf_loss_layout(a, &lo);
    if (rc) return rc;
    if (!blk || !a->net_mask || !a->obj_mask || !a->true_mask || !a->rgb || !a->rgb_gt) return mv_fail(-1, "mvsdf_loss_forward: null argument");
    char* b = (char*)blk;
    uint8_t* hit = (uint8_t*)(b + lo.hit);
    int* view_start = (int*)(b + lo.view_start);
    long long* n_pos = (long long*)(b + lo.n_pos);
    if (a->B > 1024 || a->R % a->B) return mv_fail(-1, "mvsdf_loss_forward: R must be a multiple of B <= 1024");
    if (a->n_depth > 0 && (!a->points_hom || !a->depths || !a->depth_cams || !a->size || !a->center || a->dB <= 0 || a->dh <= 0 || a->dw <= 0))
        return mv_fail(-1, "mvsdf_loss_forward: depth term without depth maps / cameras");
    // deferred step: the counts above are upper bounds, the kernels take {N, n_true} from the device
    MvDevCounts dc;
    memset(&dc, 0, sizeof(dc));
    if (a->counts_dev) {
        if (a->n_eik < 0 || a->n_ds < 0 || (a->d_mask & ~15) || (a->e_mask & ~15)) return mv_fail(-1, "mvsdf_loss_forward: bad point groups beside counts_dev");
        dc.p = a->counts_dev; dc.n_eik = a->n_eik; dc.n_ds = a->n_ds; dc.d_mask = a->d_mask; dc.e_mask = a->e_mask;
    }
    {
        PrepCarveArgs pc;
        memset(&pc, 0, sizeof(pc));
        pc.net_mask = a->net_mask; pc.obj_mask = a->obj_mask; pc.true_mask = a->true_mask; pc.R = a->R; pc.B = a->B;
        pc.hit = hit; pc.view_start = view_start; pc.n_pos = n_pos;
        pc.ticket = (unsigned*)(b + loss_scratch(a, lo) + MV_LOSS_SLICES * 8 * 4);
        CarveArgs& c = pc.c;
        c.pts = a->points_hom; c.M = a->n_depth; c.pts_ld = 4; c.pts_world = a->points_hom;      // rescaled in place (loss.py:38,42)
        c.depths = a->depths; c.B = a->dB; c.h = a->dh; c.w = a->dw; c.cams = a->depth_cams; c.size = a->size; c.center = a->center;
        c.out_thresh_perc = a->out_thresh_perc; c.far_thresh = a->far_thresh; c.far_att = a->far_att; c.near_thresh = a->near_thresh; c.near_att = a->near_att;
        c.use_invalid = a->use_invalid ? 1 : 0;
        c.dist_r = (float*)(b + lo.dist_r); c.weight = (float*)(b + lo.weight);
        pc.dc = dc;
        hipLaunchKernelGGL(k_loss_prep_carve, dim3(1 + (a->n_depth + 63) / 64), dim3(1024), 0, (hipStream_t)stream, pc);
        rc = mv_check(hipGetLastError(), "mvsdf_loss_forward (prep + carve)");
        if (rc) return rc;
    }
    const bool feat = a->feat_on && a->N > 0;
    if (feat) {
        rc = mv_feat_corr_cnt(a->diff_pts, a->N, view_start, a->B, a->V, a->C, a->H, a->W, a->feat, a->feat_strides, a->feat_src, a->src_strides, a->cam,
                              a->src_cams, a->size, a->center, (float*)(b + lo.loss_pp), (float*)(b + lo.dpts), a->counts_dev, stream);
        if (rc) return rc;
    }
    {
        if (a->R <= 0 || (a->n_grad > 0 && !a->grad_theta) || (a->n_depth > 0 && !a->eik_out) || (a->n_surf > 0 && !a->surf))
            return mv_fail(-1, "mvsdf_loss_forward: null term input");
        LossArgs k;
        memset(&k, 0, sizeof(k));
        k.rgb = a->rgb; k.rgb_gt = a->rgb_gt; k.rgb_mask = hit; k.R = a->R;
        k.grad_theta = a->n_grad > 0 ? a->grad_theta : nullptr; k.n_eik = a->n_grad;
        k.eik_out = a->eik_out; k.dist_r = (const float*)(b + lo.dist_r); k.dweight = (const float*)(b + lo.weight); k.n_depth = a->n_depth;
        k.surf = a->n_surf > 0 ? a->surf : nullptr; k.n_surf = a->n_surf; k.n_pos = n_pos;
        k.feat_pp = feat ? (const float*)(b + lo.loss_pp) : nullptr; k.n_feat = feat ? a->N : 0;
        k.w_rgb = a->w_rgb; k.w_eik = a->w_eik; k.w_surf = a->w_surf; k.w_feat = a->w_feat; k.w_depth = a->w_depth;
        k.smooth = a->smooth > 0.f ? a->smooth : 0.f;
        k.surf_on = a->surf_on; k.feat_on = a->feat_on; k.inv_counts = a->inv_counts;
        k.out = (float*)(b + lo.out); k.d_rgb = (float*)(b + lo.d_rgb); k.d_grad = (float*)(b + lo.d_grad); k.d_eik_out = (float*)(b + lo.d_eo);
        k.d_surf = (float*)(b + lo.d_sf);
        // + the gradients of the total loss itself (weights folded in): `loss.backward()` with the plain upstream 1 then needs no launch of its own
        k.s_rgb = (float*)(b + lo.s_rgb); k.s_grad = (float*)(b + lo.s_grad); k.s_eik_out = (float*)(b + lo.s_eo); k.s_surf = (float*)(b + lo.s_sf);
        k.dpts = feat ? (const float*)(b + lo.dpts) : nullptr; k.s_diff = feat ? (float*)(b + lo.s_diff) : nullptr; k.n_dpts = feat ? a->N * 3 : 0;
        k.partial = (float*)(b + loss_scratch(a, lo)); k.ticket = (unsigned*)(b + loss_scratch(a, lo) + MV_LOSS_SLICES * 8 * 4);
        k.dc = dc;
        hipLaunchKernelGGL(k_loss_terms, dim3(MV_LOSS_SLICES), dim3(1024), 0, (hipStream_t)stream, k);
        return mv_check(hipGetLastError(), "mvsdf_loss_forward (terms)");
    }
}

int mvsdf_loss_backward(const MvsdfLossArgs* a, const void* blk, const float* const* g, float* g_rgb, float* g_grad, float* g_eo, float* g_sf,
                        float* g_diff, void* stream) {
    MvsdfLossLayout lo;
    int rc = mvsdf_loss_layout(a, &lo);
    if (rc) return rc;
    if (!blk || !g) return mv_fail(-1, "mvsdf_loss_backward: null argument");
    const char* b = (const char*)blk;
    LossScaleArgs s;
    for (int k = 0; k < 6; ++k) s.g[k] = g[k];
    s.w_rgb = a->w_rgb; s.w_eik = a->w_eik; s.w_surf = a->w_surf; s.w_feat = a->w_feat; s.w_depth = a->w_depth;
    const bool feat = a->feat_on && a->N > 0;
    const float* src[5] = {(const float*)(b + lo.d_rgb), (const float*)(b + lo.d_grad), (const float*)(b + lo.d_eo), (const float*)(b + lo.d_sf),
                           (const float*)(b + lo.dpts)};
    float* dst[5] = {g_rgb, g_grad, g_eo, (a->surf_on ? g_sf : nullptr), (feat ? g_diff : nullptr)};
    const int n[5] = {a->R * 3, a->n_grad * 3, a->n_depth, a->n_surf, a->N * 3};
    int total = 0;
    for (int t = 0; t < 5; ++t) {
        s.src[t] = src[t]; s.dst[t] = dst[t]; s.n[t] = dst[t] ? n[t] : 0; total += s.n[t];
    }
    s.coef_feat = nullptr;
    memset(&s.dc, 0, sizeof(s.dc));
    if (a->counts_dev) { s.dc.p = a->counts_dev; s.dc.n_eik = a->n_eik; s.dc.n_ds = a->n_ds; s.dc.d_mask = a->d_mask; s.dc.e_mask = a->e_mask; }
    const int blocks = total > 0 ? (total + 255) / 256 : 1;
    hipLaunchKernelGGL(k_loss_scale, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, (hipStream_t)stream, s);
    return mv_check(hipGetLastError(), "mvsdf_loss_backward");
}

}  // extern "C"
