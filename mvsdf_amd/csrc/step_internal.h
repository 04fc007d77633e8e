// step_internal.h -- library-internal entry points shared by step_kernels.hip / diff_mlp.hip and the step driver (step_driver.hip).
// Not part of the C ABI (include/mvsdf_hip.h): same conventions (device pointers, stream as void*, 0 on success).
#pragma once
#include <stdint.h>
#include "../../include/mvsdf_hip.h"

// optional row source of k_chain_fwd: the training step's evaluation rows [eikonal samples | on-surface samples | jittered samples | traced points of the
// rays in sorted order] gathered on the fly (and written to x_out for the later consumers) instead of by a separate launch
struct FwdGather { const float* eik; const float* on; const float* jit; const float* pts; const long long* perm; int n_eik, n_ds; float* x_out; };

// k_partition_rays with the step driver's extra outputs: true_rank[pos] (rank of sorted hit row pos among the true-mask hit rows, -1 outside the
// true mask) and counts[2..3] = extra_counts[0..1] (0 when NULL)
int mv_partition_rays_step(const uint8_t* net_mask, const uint8_t* object_mask, const uint8_t* true_mask, const float* ray_dirs, int R, long long* perm,
                           long long* inv, long long* true_rows, long long* counts, float* view_sorted, int* true_rank, const long long* extra_counts,
                           long long* counts_host, long long counts_seq, float* term_rows, int n_eik, int n_ds, int d_mask, int e_mask,
                           void* stream);   // term_rows (optional, [3] floats): rows of the eikonal / depth / surface terms for these counts; counts_host: optional device pointer of host-mapped pinned memory, receives the 4 counts, then counts_seq in entry 4 (system-scope release: the host polls it)
// dy / dn (full pass A upstream without SampleNetwork's scalar) and dy_x / dn_x (rendering-net adjoints alone on the hit rows) in one gather pass
int mv_step_backward_assemble(int n_eik, int n_ds, int N, int Nout, int n_true, const float* din, int din_ld, int din_feat0, int din_nrm0, int use_geo,
                              const int* true_rank, const float* d_eo, const float* d_gth, const float* d_si, int d_mask, int e_mask, float* dy,
                              float* dn, float* dy_x, float* dn_x, const long long* cnt, void* stream);

// ---- device-side counts (the deferred step) ------------------------------------------------------------------------------------------------
// A training step whose host never learns the hit counts: k_partition_rays leaves counts = {N hit rows, n_true of them inside the true mask} in the forward
// block, and every N-dependent launch behind it takes them from there.  Convention of the `cnt` parameters below and of the kernels' argument structs:
// cnt == NULL: the row counts passed by value are exact (the classic step; grids, layouts and bounds follow them).  cnt != NULL: cnt[0] = N, cnt[1] = n_true on
// the DEVICE; the by-value counts are then UPPER BOUNDS (N = R): workspaces, their layouts and the grids are sized for them, every kernel bounds its rows by
// the device values, and workgroups wholly beyond them leave at once.  Results are bit-identical to the classic step's (tests/test_gpu_deferred.py).
int mv_sdf_backward_pair_cnt(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int M, int Mg, int MbA, const float* dyA, const float* dnA, float* wsA,
                             int row0X, int MbX, const float* dyX, const float* dnX, float* wsX, float* dx, const float* ctx, const long long* cnt, int n_hint,
                             void* stream);
// 1 when every launch of mvsdf_step_backward has a device-count form for these networks (the fused chain kernels cover them), else 0
int mv_step_can_defer(const MvsdfNetDesc* sdf, const MvsdfNetDesc* sdfT, const MvsdfNetDesc* rnd, const MvsdfNetDesc* rndT);

// mvsdf_sdf_forward with the step's evaluation rows gathered inside the fused chain kernel; `gather` = const FwdGather* (layer_kernels.h) or NULL;
// -> 1 when the per-layer route would run (nothing launched: gather yourself and call with x)
// [r_begin, r_end): the rows this launch evaluates (a proper sub-range only through the fused chain; -> 1 otherwise)
int mv_sdf_forward_gather(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, const float* x, const void* gather, int M, int Mg, int r_begin, int r_end,
                          float* y, float* nrm, float* ctx, void* stream);
// 1 when the fused chain over the rows [E, M) alone is a shorter launch than over [0, M) (diff_mlp.hip)
int mv_chain_split_pays(const MvsdfNetDesc* d, int E, int M);

// pieces of the training step's backward (diff_mlp.hip): the rendering net's descending chain alone, the SDF net's delta pass alone, and the weight
// gradients of BOTH networks as one k_wgrad_net / k_reduce_net pair
// drgb_rows (may be NULL): sorted row r takes its upstream from drgb[drgb_rows[r]]; -3 when that needs the fused chain kernel and it does not apply
int mv_render_backward_chain(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int N, int Nctx, const float* drgb, const long long* drgb_rows, const float* ctx,
                             float* din, float* ws, const long long* cnt, void* stream);
int mv_delta_is_chain();                                       // MVSDF_DELTA_CHAIN=1
int mv_sdf_backward_delta_fbar(const MvsdfNetDesc* d, int M, int Mg, int Mb, const float* ctx, float* ws, int row0D, int MbD, int Nout, const float* din,
                               int din_ld, int use_geo, const float* d_diff, const float* dx, const float* view_sorted, const float* n_eval, float* dy,
                               float* fbar, const long long* cnt, void* stream);
int mv_sdf_backward_delta(const MvsdfNetDesc* d, const MvsdfNetDesc* dT, int M, int Mg, int Mb, const float* ctx, float* ws, int row0D, int MbD,
                          const float* fbar, void* stream);
int mv_step_wgrad(const MvsdfNetDesc* sd, const MvsdfNetDesc* rd, int M, int Mg, int Mb, const float* dy, const float* ctx, float* wsA, int N, int Nctx,
                  const float* rctx, float* rws, float* dW_s, float* db_s, float* dW_r, float* db_r, const long long* cnt, int cnt_base, void* stream);

// fold + MFMA packs (+ bf16 packs where wp16[l] is set: nsplit[l] = its PE split width) of every layer + the camera rays in ONE launch
// (basic.hip::k_step_prologue): the work of mvsdf_fold_pack_net, mvsdf_pack_bf16_net_skips and mvsdf_camera_rays, same results
int mv_step_prologue(int n_layers, const float* const* v, const float* const* g, const int* N, const int* K, float* const* w, float* const* wp,
                     float* const* wpT, void* const* wp16, const int* nsplit, int wp16_mode, void* const* wx3, void* const* wx3T, const float* uv, const float* pose,
                     const float* intrinsics, int B, int P,
                     float* ray_dirs, float* cam_loc, uint8_t* ones, unsigned long long* counters, const float* stage_src, float* stage_a, int stage_na,
                     float* stage_b, int stage_nb, void* stream);   // stage_src (optional): device-visible pinned host memory [na | nb] -> stage_a, stage_b
                                                                    // wx3 / wx3T (optional, per layer, entries may be null): the three-term bf16 packs of W_l / W_l^T
                                                                    // of the differentiable chains (chain_x3.h; layouts of mvsdf_pack_bf16x3_net / _bf16x3t_net)
int mv_chain_x3_enabled();                                          // the fused SDF chains run in the three-term bf16 arithmetic when the packs exist (dev: MVSDF_CHAIN_X3=0)
// stage 1 of mvsdf_trace_stage for a caller whose previous launch (mv_step_prologue) zeroed the counters
extern "C" int mv_trace_stage1_prezeroed(const MvsdfNetDesc* desc, const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, const uint8_t* object_mask,
                              int B, int P, int training, const float* intervals, const float* minsdf_steps, float* points, uint8_t* mask, float* dists,
                              unsigned long long* counters, void* workspace, size_t workspace_bytes, int mt, int rpw, void* stream);
