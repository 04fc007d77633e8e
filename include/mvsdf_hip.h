/* mvsdf_hip.h -- C ABI of libmvsdf_hip.so, the MI355X (gfx950) native hot path of MVSDF.
 *
 * The reference (jzhangbs/MVSDF) has no FFI for this path: it is eager PyTorch behind nn.Module classes
 * (SURVEY.md section 8b).  Each entry point below replaces the PyTorch op sequence of the cited reference
 * lines; the Python mirror (mvsdf_amd/model/...) binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions: all pointers are DEVICE pointers unless marked host; tensors are dense row-major fp32;
 * masks are uint8 (0/1); `stream` is a hipStream_t passed as void*; every call is asynchronous on that
 * stream and returns 0 on success or a nonzero code (hipError_t value, or negative for bad arguments);
 * mvsdf_last_error() gives a message.  No call allocates or synchronises.
 *
 * What this boundary REFUSES (the call returns a negative code and mvsdf_last_error() names the reason; the Python mirror raises
 * NotImplementedError / ValueError before it gets here) -- every other constructor option of the reference's three modules is accepted:
 *      (a skip connection into the LAST Linear of the SDF network, `skip_in` containing num_layers - 2 (idr.py:46-49,86), is accepted since round 5:
 *      u_L = W_L[0, :] splits like any skip layer's adjoint, its PE columns start the PE adjoint of the normal chain)
 *   1. `d_in != 3` for ImplicitNetwork (idr.py:22,34): points are 3-vectors everywhere on this path (camera rays, PE of 3 + 6 * multires columns).
 *   2. hidden widths above 512 (column tiles per wave of the fused engines: 2 up to 256, 4 up to 512), and more than MVSDF_MAX_LAYERS = 12 Linears.
 *   3. quaternion poses in get_camera_params (rend_util.py:49-54: the `pose.shape[1] == 7` branch): the reference switches it off
 *      (exp_runner.py:40 train_cameras = False; scene_dataset.py hands 4 x 4 matrices); mvsdf_camera_rays takes pose[B][4][4] only.
 */
#ifndef MVSDF_HIP_H
#define MVSDF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVSDF_MAX_LAYERS 12

/* A weight-norm-folded MLP in MFMA-packed form (see mvsdf_fold_pack).  host struct, device pointers. */
typedef struct {
    int n_layers;
    int K[MVSDF_MAX_LAYERS];            /* in features per Linear */
    int N[MVSDF_MAX_LAYERS];            /* out features per Linear */
    const float* wp[MVSDF_MAX_LAYERS];  /* packed weights, mvsdf_packed_floats(N, K) floats */
    const float* bias[MVSDF_MAX_LAYERS]; /* [N] floats.  trace_dtype 3 / 4 / 5 read each vector with 16-byte loads up to the next multiple of 16 entries (the
                                         * entries past N are loaded and never used): the buffer must be READABLE that far.  Every hipMalloc'd or torch-allocated
                                         * buffer is (allocation granularity >= 64 bytes); a caller that carves biases out of its own arena pads each vector to a
                                         * multiple of 16 floats (a vector that ends exactly at the end of a mapped range would fault otherwise) */
    const float* w[MVSDF_MAX_LAYERS];   /* folded weights, row-major [N][K] (only the last layer's row 0 is read: u_L = W_L[0,:]); may be NULL when no normals are needed */
    int skip_layer;                     /* layer whose input is cat([x, PE(x)])/sqrt(2) (idr.py:86-87), -1 if none (see skip_mask for several) */
    int multires;                       /* positional-encoding frequencies (embedder.py:38-50) */
    const void* wp16[MVSDF_MAX_LAYERS]; /* trace_dtype 2: fp32 packs of the rounded weights
                                         * (mvsdf_pack_bf16w_net); 3 / 4: bf16 packs made by mvsdf_pack_bf16s_net; 5: three-term packs made by mvsdf_pack_bf16x3_net;
                                         * NULL for trace_dtype 0 */
    int trace_dtype;                    /* arithmetic of the no-grad tracing MLP (mvsdf_trace, mvsdf_sdf_col0): 0 = fp32 weights and fp32-input
                                         * MFMA (bit-exact against the fmaf-chain oracle), 1 = REMOVED in round 5 (bf16 weights AND 8-bit activations: every entry point refuses it),
                                         * 2 = bf16-ROUNDED WEIGHTS ONLY: wp16[l] holds an fp32 pack (mvsdf_packed_floats(N, K) floats, made by
                                         * mvsdf_pack_bf16w_net) of the weights rounded to bf16, activations stay fp32 on the fp32-input MFMA -- bit-exact
                                         * against the oracle run on the rounded weights,
                                         * 3 / 4 = bf16 weights on the bf16 MFMA with every activation carried as 2 / 3 bf16 TERMS (a = t0 + t1 [+ t2], 16 / all 24
                                         * mantissa bits; csrc/tile_engine_bf16s.h): the arithmetic of mode 2 (idr.py:77-94 on bf16-rounded weights) up to the order
                                         * of the fp32 additions inside the matrix core -- the fast mode that is parity-checked against that oracle (hit masks
                                         * equal except at recorded ties, depths 1e-4).  bias[] is read with 16-byte loads up to the next multiple of 16 entries,
                                         * 5 = the fp32 weights UNROUNDED as three bf16 terms too (w = w0 + w1 + w2 exactly): the reference's fp32 Linear
                                         * (idr.py:89) from the six exact products a_s w_j, s + j <= 2, on the bf16 MFMA -- fp32-accurate (measured closer to an
                                         * fp64 evaluation than mode 0's fmaf chain) and reproduced BIT FOR BIT by the oracle's model of the matrix instruction
                                         * (oracle_mvsdf.c::sdf_row_f32x3); det_math's softplus like mode 0; values below 2^-40 count as zero.  Not bit-identical
                                         * to mode 0 (another summation order).  bias[] is read like in modes 3 / 4 */
    unsigned skip_mask;                 /* several skip connections (skip_in with more than one entry, idr.py:46,86): bit l set = the input of layer l
                                         * is cat([x, PE(x)])/sqrt(2).  0 = use skip_layer alone.  Any layer but layer 0 (the last Linear included). */
    const void* wx3[MVSDF_MAX_LAYERS];  /* optional: three-term bf16 packs of THIS descriptor's matrices (mvsdf_pack_bf16x3_net for W_l; mvsdf_pack_bf16x3t_net for the
                                         * transposed descriptor's W_l^T).  When every layer of both descriptors has one, the differentiable passes of the SDF network
                                         * (mvsdf_sdf_forward / _backward / _backward_pair, the training step) run their fused chains in the three-term fp32 arithmetic
                                         * on the bf16 matrix cores (csrc/chain_x3.h) instead of the fp32-input MFMA chains; NULL: the fp32 chains.  bias[] is then read
                                         * like in trace_dtype 3 / 4 / 5 */
} MvsdfNetDesc;

/* RayTracing constructor arguments (ray_tracing.py:7-25) + the hard-coded dist_clip (ray_tracing.py:127-131). */
typedef struct {
    float r;                 /* object_bounding_sphere */
    float thr;               /* sdf_threshold */
    float line_search_step;
    int line_step_iters;
    int st_iters;            /* sphere_tracing_iters */
    int n_steps;
    int n_secant;            /* n_secant_steps */
    float dist_clip;         /* 0.5 (0.05 in IDR_RENDER mode) */
} MvsdfTraceParams;

/* indices into the uint64 counters[16] array filled by mvsdf_trace (device memory) */
enum {
    MVSDF_CNT_ROWS_SPHERE = 0,  /* sdf() rows evaluated by sphere_tracing   (ray_tracing.py:134,137,168,171,185,186) */
    MVSDF_CNT_ROWS_SAMPLER = 1, /* ... by ray_sampler                       (ray_tracing.py:218) */
    MVSDF_CNT_ROWS_SECANT = 2,  /* ... by secant                            (ray_tracing.py:266) */
    MVSDF_CNT_ROWS_MINSDF = 3,  /* ... by minimal_sdf_points                (ray_tracing.py:301) */
    MVSDF_CNT_N_SECANT = 4,     /* rays that ran the secant */
    MVSDF_CNT_N_SAMPLER = 5,    /* rays on the sampler work list */
    MVSDF_CNT_N_MINSDF = 6,     /* rays on the min-sdf work list */
    MVSDF_CNT_N_SAMPLER_REST = 7,    /* sampler rays whose first sample window did not settle them (no sign change yet) */
    MVSDF_CNT_ROWS_SAMPLER_EVAL = 8, /* ray_sampler rows actually evaluated here: the samples AFTER a ray's first sign change cannot
                                      * influence any output (ray_tracing.py:221-256 reads only sdf_val[ind-1], sdf_val[ind]) and are
                                      * skipped.  counters[1] keeps the reference's count (n_rays * n_steps). */
    /* tail filling of the sphere-tracing kernel (training): the min-sdf rows are a queue that finished sphere-tracing workgroups serve until
     * the slowest one is done, the min-sdf launch takes the rest */
    MVSDF_CNT_TAIL_NEXT = 9,    /* next unit of min-sdf rows to hand out */
    MVSDF_CNT_TAIL_READY = 10,  /* min-sdf list items completely written */
    MVSDF_CNT_TAIL_WGS = 11,    /* sphere-tracing workgroups whose rays are done */
    MVSDF_CNT_TAIL_ROWS = 12    /* min-sdf rows evaluated inside the sphere-tracing kernel (a subset of counters[3]) */
};

int mvsdf_version(void);
/* sizeof() of the structs of this header as the library was compiled, in the order MvsdfNetDesc, MvsdfTraceParams, MvsdfStepDesc, MvsdfStepParams,
 * MvsdfStepInputs, MvsdfStepLayout, MvsdfLossArgs, MvsdfLossLayout -> out[8]; a binding checks its own struct definitions against it
 * (tests/test_abi.py does for the ctypes binding).  Returns 8. */
int mvsdf_abi_struct_sizes(size_t* out);
const char* mvsdf_last_error(void);

/* number of floats of the packed form of an [N][K] Linear (both dims rounded up to 16) */
size_t mvsdf_packed_floats(int N, int K);

/* weight_norm fold  w = v * (g / ||v||_row)  (idr.py:70-71, torch._weight_norm dim=0) + MFMA packing.
 * v[N][K], g[N] -> w[N][K] (row-major, required), wp (packed W, mvsdf_packed_floats(N, K) floats, may be NULL),
 * wpT (packed W^T for contractions over the OUT dimension, mvsdf_packed_floats(K, N) floats, may be NULL). */
int mvsdf_fold_pack(const float* v, const float* g, int N, int K, float* w, float* wp, float* wpT, void* stream);
/* all layers of a network -- or of several networks, up to 24 layers in total -- at once (one fold launch + one pack launch).  g[l] may be
 * NULL for a layer without weight norm (weight_norm=False, idr.py:70-71 skipped): w = v, and the backward gives dv = dW, no dg.  The
 * pointer arrays are HOST arrays of n_layers device pointers; wp[l] / wpT[l] may be NULL. */
int mvsdf_fold_pack_net(int n_layers, const float* const* v, const float* const* g, const int* N, const int* K, float* const* w,
                        float* const* wp, float* const* wpT, void* stream);
/* backward of every fold in one launch.  db / dbias (both NULL or both given; entries may be NULL pairwise) route the bias
 * gradients db[l][N_l] to dbias[l] in the same launch; accumulate != 0 adds into dv / dg / dbias (the targets may be the
 * parameters' .grad buffers: what autograd's AccumulateGrad would do with ~3 launches per layer). */
int mvsdf_fold_backward_net(int n_layers, const float* const* v, const float* const* g, const float* const* dW, const float* const* db,
                            const int* N, const int* K, float* const* dv, float* const* dg, float* const* dbias, int accumulate,
                            void* stream);
/* backward of the fold: dW[N][K] -> dv[N][K], dg[N]    */
int mvsdf_fold_backward(const float* v, const float* g, const float* dW, int N, int K, float* dv, float* dg, void* stream);

/* bytes of one layer's bf16 pack in the layout of v_mfma_f32_16x16x32_bf16 (wp16[l] of trace_dtype 3 / 4; three times that for trace_dtype 5);
 * nsplit = 0 (duplicated hi / lo input columns belonged to trace_dtype 1, removed in round 5). */
size_t mvsdf_packed_bf16_bytes(int N, int K, int nsplit);
/* trace_dtype = 2: fp32 MFMA packs (layout of mvsdf_fold_pack's wp) of the folded weights w[l] rounded to bf16 (nearest even) */
int mvsdf_pack_bf16w_net(int n_layers, const float* const* w, const int* N, const int* K, float* const* wp_rounded, void* stream);
/* trace_dtype = 3 / 4: bf16 packs (v_mfma_f32_16x16x32_bf16 B-operand layout) WITHOUT duplicated columns (the positional-encoding inputs are split into bf16 terms like
 * every other activation); wp16[l]: mvsdf_packed_bf16_bytes(N, K, 0) bytes.  idr.py:77-94 on bf16-rounded weights. */
int mvsdf_pack_bf16s_net(int n_layers, const float* const* w, const int* N, const int* K, void* const* wp16, void* stream);
/* trace_dtype = 5: the folded fp32 weights as three bf16 terms (t0 = bf16(w), t1 = bf16(w - t0), t2 = bf16(w - t0 - t1)), layout of mvsdf_pack_bf16s_net
 * with a k-block's three term fragments behind each other; wp16[l]: 3 * mvsdf_packed_bf16_bytes(N, K, 0) bytes.  idr.py:77-94 on the fp32 weights. */
int mvsdf_pack_bf16x3_net(int n_layers, const float* const* w, const int* N, const int* K, void* const* wp16, void* stream);
/* the same three-term pack of W_l^T (N, K are W_l's dims): 3 * mvsdf_packed_bf16_bytes(K, N, 0) bytes per layer -- MvsdfNetDesc.wx3 of the transposed descriptor */
int mvsdf_pack_bf16x3t_net(int n_layers, const float* const* w, const int* N, const int* K, void* const* wx3t, void* stream);

/* ImplicitNetwork.forward(x)[:, 0] (idr.py:77-94) for n points: the tracing MLP alone. */
int mvsdf_sdf_col0(const MvsdfNetDesc* net, const float* x, int n, float* y, int mt, void* stream);

/* rend_util.get_camera_params + lift, pose-matrix branch (rend_util.py:48-75, 87-100):
 * uv[B][P][2], pose[B][4][4], intrinsics[B][4][4] -> ray_dirs[B][P][3], cam_loc[B][3]. */
int mvsdf_camera_rays(const float* uv, const float* pose, const float* intrinsics, int B, int P, float* ray_dirs, float* cam_loc,
                      void* stream);

/* rend_util.get_sphere_intersection (rend_util.py:141-162): -> t[B*P][2] (near, far; clamped at 0), mask[B*P] (disc > 0). */
int mvsdf_sphere_intersection(const float* cam_loc, const float* ray_dirs, int B, int P, float r, float* t, uint8_t* mask, void* stream);

/* RayTracing.forward (ray_tracing.py:27-98) with the SDF given by `net`:
 * cam_loc[B][3], ray_dirs[B][P][3], object_mask[B*P] -> points[B*P][3], mask[B*P], dists[B*P].
 * intervals[n_steps] = torch.linspace(0, 1, n_steps) (ray_tracing.py:206); minsdf_steps[n_steps] = the uniform draws
 * of minimal_sdf_points (ray_tracing.py:287), used only when training != 0.
 * counters: uint64[16], zeroed by the call.  workspace: mvsdf_trace_workspace_bytes(B*P) bytes (n_steps <= 128) or
 * mvsdf_trace_workspace_bytes_n(B*P, n_steps).
 * mt: row tiles (16 rows) per workgroup of the sphere-tracing kernel (8*mt rays), 1..4;
 * mt_samples: row tiles per chunk of the flattened sample-row kernels (sampler / min-sdf), 1..4. */
size_t mvsdf_trace_workspace_bytes(int R);   /* = mvsdf_trace_workspace_bytes_n(R, 128) */
size_t mvsdf_trace_workspace_bytes_n(int R, int n_steps);
int mvsdf_trace(const MvsdfNetDesc* net, const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs,
                const uint8_t* object_mask, int B, int P, int training, const float* intervals, const float* minsdf_steps,
                float* points, uint8_t* mask, float* dists, unsigned long long* counters, void* workspace,
                size_t workspace_bytes, int mt, int mt_samples, void* stream);
/* the launches of mvsdf_trace separately, same arguments and workspace.  stage 1: sphere tracing (zeroes the counters);
 * stage 2: ray sampler + secant + min-sdf; or stage 3: ray sampler rows only -- `mask` is FINAL after it (ray_tracing.py:61) --
 * followed by stage 4: secant + min-sdf (only points / dists still change, ray_tracing.py:63-96); stage 4 may be split further into
 * stage 5: min-sdf rows + their reduction alone (independent of stage 3: they only need stage 1's work list, and use their own sample-value
 * buffer, so a caller may run them on another stream concurrently with stage 3) and stage 6: secant alone.  Lets a caller bracket each
 * kernel with events, and fetch the hit count to the host while stage 4 still runs. */
int mvsdf_trace_stage(int stage, const MvsdfNetDesc* net, const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs,
                      const uint8_t* object_mask, int B, int P, int training, const float* intervals, const float* minsdf_steps,
                      float* points, uint8_t* mask, float* dists, unsigned long long* counters, void* workspace,
                      size_t workspace_bytes, int mt, int mt_samples, void* stream);

/* ---- RayTracing.forward for an OPAQUE `sdf` callable (ray_tracing.py:27-32, called with a lambda at idr.py:194) ----
 * The per-ray state machine of mvsdf_trace split at its evaluation points, so that a host-side callable can be run between the launches:
 *   init   : sphere intersection + first requests.  state: mvsdf_tracegen_state_bytes(R) bytes; req[R][2] (uint8: start / end side wants a
 *            value), pts[R][2][3] (the requested points, dense); zeroes counters[16].
 *   step   : consumes vals[R][2] (the callable's values at the requested entries, anything elsewhere), advances every ray by one round
 *            (ray_tracing.py:139-194) and emits the next requests.  The caller loops until no entry of req is set.
 *   finish : masks / dists / points of the sphere-tracing stage, sampler and min-sdf work lists IN RAY ORDER (counters[MVSDF_CNT_N_SAMPLER],
 *            [MVSDF_CNT_N_MINSDF]); workspace as for mvsdf_trace.
 *   rows   : the n_steps sample points of each listed ray (kind 0: ray_sampler with `zs` = intervals, ray_tracing.py:206-213; kind 1:
 *            minimal_sdf_points with `zs` = the uniform draws, 287-297) -> out_pts[n_list * n_steps][3].
 *   reduce : the per-ray decisions on the callable's values sv[n_list][n_steps] (kind 0: ray_tracing.py:221-256, fills the secant list sorted
 *            by ray, marks[R] is scratch; kind 1: 303-307).
 *   secant : op 0 emits the n_sec points of one secant round -> pts_out[n_sec][3]; op 1 consumes vals[n_sec]; op 2 writes the final result. */
size_t mvsdf_tracegen_state_bytes(int R);
int mvsdf_tracegen_init(const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, const uint8_t* object_mask, int B, int P,
                        void* state, uint8_t* req, float* pts, unsigned long long* counters, void* stream);
int mvsdf_tracegen_step(const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, int B, int P, void* state, const float* vals,
                        uint8_t* req, float* pts, unsigned long long* counters, void* stream);
int mvsdf_tracegen_finish(const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, int B, int P, int training, const void* state,
                          float* points, uint8_t* mask, float* dists, unsigned long long* counters, void* workspace, size_t workspace_bytes,
                          void* stream);
int mvsdf_tracegen_rows(const MvsdfTraceParams* tp, int kind, const float* cam_loc, const float* ray_dirs, int B, int P, const float* zs, int n_list,
                        void* workspace, float* out_pts, void* stream);
int mvsdf_tracegen_reduce(const MvsdfTraceParams* tp, int kind, const float* cam_loc, const float* ray_dirs, int B, int P, int training,
                          const float* intervals, const float* minsdf_steps, const float* sv, float* points, uint8_t* mask, float* dists,
                          unsigned long long* counters, void* workspace, uint8_t* marks, void* stream);
int mvsdf_tracegen_secant(const MvsdfTraceParams* tp, int op, const float* cam_loc, const float* ray_dirs, int B, int P, int n_sec, const float* vals,
                          float* pts_out, float* points, float* dists, unsigned long long* counters, void* workspace, void* stream);

/* ---- differentiable SDF network: value + normal and their first/second-order backward ----
 * Replaces ImplicitNetwork.forward / .gradient (idr.py:77-107) and autograd's (double) backward through them.
 * net: packs of W_l; netT: packs of W_l^T (K and N swapped per layer).  x[M][3].  The first Mg rows also get the normal
 * n = d out[:,0] / dx (NOT normalised, idr.py:327).  y[M][Nout], nrm[Mg][3].  ctx: mvsdf_sdf_ctx_floats(net, M, Mg) floats,
 * kept by the caller until the backward.  */
size_t mvsdf_sdf_ctx_floats(const MvsdfNetDesc* net, int M, int Mg);
int mvsdf_sdf_forward(const MvsdfNetDesc* net, const MvsdfNetDesc* netT, const float* x, int M, int Mg, float* y, float* nrm, float* ctx,
                      void* stream);
/* Backward over rows [row0, row0 + Mb) (inside [0, M); inside [0, Mg) when dn is given): dy[Mb][Nout], dn[Mb][3] or NULL ->
 * dW_cat (all layers, row-major [N][K], concatenated), db_cat (both NULL: input adjoint only), dx[Mb][3] or NULL.  ws: mvsdf_sdf_bwd_ws_floats(net, Mb) floats. */
size_t mvsdf_sdf_bwd_ws_floats(const MvsdfNetDesc* net, int Mb);
int mvsdf_sdf_backward(const MvsdfNetDesc* net, const MvsdfNetDesc* netT, const float* x, int M, int Mg, int row0, int Mb, const float* dy,
                       const float* dn, const float* ctx, float* dW_cat, float* db_cat, float* dx, float* ws, void* stream);

/* The training step's SDF backward in two chain launches instead of three sequential ones (functional._IdrStep.backward):
 *   pair   : pass A = full backward over rows [0, MbA) with (dyA[MbA][Nout], dnA[MbA][3]), per-layer adjoints kept in wsA
 *            (mvsdf_sdf_bwd_ws_floats(net, MbA)); pass X = input adjoint only over rows [row0X, row0X + MbX) with (dyX, dnX) -> dx[MbX][3]
 *            (scratch wsX).  Independent, ONE grid.  Returns -3 when the fused chain kernels do not cover the network (use mvsdf_sdf_backward).
 *   finish : delta pass over rows [row0D, row0D + MbD): extra upstream fbar[MbD] on output column 0 only (SampleNetwork's scalar, known
 *            after pass X), its first-order adjoints are ADDED to wsA's (linearity); dy must already include fbar in column 0 of those rows.
 *            Then the weight / bias gradients of every layer -> dW_cat, db_cat. */
int mvsdf_sdf_backward_pair(const MvsdfNetDesc* net, const MvsdfNetDesc* netT, int M, int Mg, int MbA, const float* dyA, const float* dnA, float* wsA,
                            int row0X, int MbX, const float* dyX, const float* dnX, float* wsX, float* dx, const float* ctx, void* stream);
int mvsdf_sdf_backward_finish(const MvsdfNetDesc* net, const MvsdfNetDesc* netT, int M, int Mg, int Mb, const float* dy, const float* ctx, float* ws,
                              int row0D, int MbD, const float* fbar, float* dW_cat, float* db_cat, void* stream);

/* ---- rendering network (idr.py:145-167): rgb = tanh(MLP(cat[points, PE(view), normals, feat])) ----
 * multires_view: low 8 bits = positional-encoding frequencies of the view direction (0: the raw direction); bit 8 (0x100) = mode 'no_view_dir'
 * (input cat[points, normals, feat]); bit 9 (0x200) = mode 'no_normal' (cat[points, PE(view), feat]); neither = mode 'idr'.  The unused input
 * pointer must still be valid device memory. */
size_t mvsdf_render_ctx_floats(const MvsdfNetDesc* net, int N);
size_t mvsdf_render_bwd_ws_floats(const MvsdfNetDesc* net, int N);
int mvsdf_render_forward(const MvsdfNetDesc* net, const float* points, const float* view, const float* normals, const float* feat,
                         int ldfeat, int N, int multires_view, float* rgb, float* ctx, void* stream);
/* din[N][K0]: adjoint of the concatenated input (points = [:,0:3], normals = [:,3+dv:6+dv], feat = [:,6+dv:], dv = 3+6*multires_view).
 * ctx was made by mvsdf_render_forward over Nctx >= N rows; the backward covers its first N rows (a training step renders every sorted
 * ray before the host knows how many of them hit). */
int mvsdf_render_backward(const MvsdfNetDesc* net, const MvsdfNetDesc* netT, int N, int Nctx, const float* drgb, const float* ctx,
                          float* dW_cat, float* db_cat, float* din, float* ws, void* stream);

/* ---- IDRLoss.get_feat_loss_corr (model/loss.py:115-165 + utils/my_utils.py:98-110,152-165 + F.grid_sample) ----
 * forward AND analytic d(loss)/d(points) in one launch (feature maps are constants).  pts[N][3] = diff_surf_pts (hit points,
 * view-major); view_start[B+1] (device int32) = prefix sums of per-view hit counts; feat[B][C][H][W], feat_src[B][V][C][H][W]
 * addressed through element strides (NCHW or channels_last, C <= 32); cam[B][2][4][4], src_cams[B][V][2][4][4], size[1],
 * center[3] on the device.  loss_pp[N]: per-point terms (their sum is the loss); dpts[N][3]. */
int mvsdf_feat_corr(const float* pts, int N, const int* view_start, int B, int V, int C, int H, int W, const float* feat,
                    const long long* feat_strides, const float* feat_src, const long long* src_strides, const float* cam,
                    const float* src_cams, const float* size, const float* center, float* loss_pp, float* dpts, void* stream);

/* ---- depth-carving target of IDRLoss.get_depth_loss (loss.py:37-63 -> my_utils.py:269-331 carving_t2) ----
 * pts[M][pts_ld] (first 3 columns: normalised sample points, e.g. eikonal_points_hom with pts_ld = 4), depths[B][h][w],
 * cams[B][2][4][4] -> dist_r[M], weight[M]; pts_world (may be NULL, may alias pts, same row stride) receives the points rescaled to
 * world coordinates -- the reference does that in place on eikonal_points_hom (loss.py:38,42);
 * loss = mean(|eikonal_output + dist_r| * weight)  (weight = far/near attenuation * in_range).
 * use_invalid != 0: carving_t (conf.use_invalid, loss.py:43-44 -> my_utils.py:204-266): a view that sees the point but has no depth there counts as half an
 * 'outside' vote, and `in_range` is "some view sees the point" instead of "some view has a depth there". */
int mvsdf_depth_carve(const float* pts, int pts_ld, int M, const float* depths, int B, int h, int w, const float* cams, const float* size,
                      const float* center, float out_thresh_perc, float far_thresh, float far_att, float near_thresh, float near_att, int use_invalid,
                      float* dist_r, float* weight, float* pts_world, void* stream);

/* ---- the elementwise terms of IDRLoss.forward + weighted total (loss.py:21-35, 58-61, 167-174, 206-210), one launch ----
 * rgb[R][3], rgb_gt[R][3], rgb_mask[R] (network_object_mask & object_mask); grad_theta[n_eik][3]; eik_out / dist_r / dweight[n_depth]
 * (dist_r, dweight from mvsdf_depth_carve); surf[n_surf] logits with targets (i < *n_pos); feat_pp[n_feat] from mvsdf_feat_corr or NULL.
 * smooth: 0 = the L1 depth term (loss.py:60), s > 0 = SmoothL1(eikonal_output / s, -dist_r / s) * s (loss.py:57-58).
 * out[6] = {loss, rgb_loss, eikonal_loss, depth_loss, feat_loss, surf_loss}; d_*: unit gradients of each term w.r.t. its input.
 * inv_counts (device, [3], may be NULL): replaces 1/n_eik, 1/n_depth, 1/n_surf of the three count-normalised means -- with
 * world_size / (count summed over the data-parallel ranks) the rank-averaged gradient equals the single-process one. */
int mvsdf_loss_terms(const float* rgb, const float* rgb_gt, const uint8_t* rgb_mask, int R, const float* grad_theta, int n_eik,
                     const float* eik_out, const float* dist_r, const float* dweight, int n_depth, const float* surf, int n_surf,
                     const long long* n_pos, const float* feat_pp, int n_feat, float w_rgb, float w_eik, float w_surf, float w_feat,
                     float w_depth, float smooth, int surf_on, int feat_on, const float* inv_counts, float* out, float* d_rgb, float* d_grad,
                     float* d_eik_out, float* d_surf, void* stream);

/* ---- bookkeeping of one training step of IDRNetwork.forward (idr.py:202-304) between the big kernels ----
 * mvsdf_partition_rays: stable partition of the R rays by surface = net_mask & object_mask (object_mask / true_mask may be NULL = all
 * ones): perm[R] = [surface rays in ray order | the others in ray order], inv[R] its inverse, true_rows[R] = ranks among the surface
 * rays of those inside true_mask (first counts[1] entries valid), counts[2] = {#surface, #surface & true}; view_sorted[R][3]
 * (may be NULL) = -ray_dirs[perm[r]].  Replaces the boolean-mask indexing of idr.py:202-213, 272 (which syncs the host per mask). */
int mvsdf_partition_rays(const uint8_t* net_mask, const uint8_t* object_mask, const uint8_t* true_mask, const float* ray_dirs, int R,
                         long long* perm, long long* inv, long long* true_rows, long long* counts, float* view_sorted, void* stream);
/* Output tensors of the training forward gathered from ONE fused evaluation over the rows [E sample points | R rays sorted, hit first]
 * (x_eval[E+R][3], y_eval[E+R][Nout], n_eval[E+R][3]).  The sample rows are [n_eik eikonal | n_ds on-surface | n_ds jittered]
 * (E = n_eik + 2 n_ds).  Point groups in the reference's order: 0 = hit rays, 1 = eikonal, 2 = on-surface, 3 = jittered;
 * d_mask / e_mask (bit g = group g) select the groups of the depth term (eikonal_output, eikonal_points_hom) and of the eikonal term
 * (grad_theta), idr.py:258-286.  counts (DEVICE, {N, n_true} from mvsdf_partition_rays): the kernel reads the hit counts there, so it
 * can be enqueued before the host knows them; every output is sized for the worst case (N = R) and the first
 * [N + selected samples] rows are valid.  rgb_values[R][3] (rgb_sorted[r] for sorted rows r < N, 1 elsewhere; idr.py:302-304),
 * sdf_output[R], diff_pts[R][3], eik_out[.], points_hom[.][4] (x, 1), grad_theta[.][3], surf[R + n_eik] = column 1 at the true-mask hit
 * rows, then at the n_eik eikonal rows (idr.py:270-276). */
int mvsdf_step_outputs(int R, int n_eik, int n_ds, int Nout, const long long* counts, const float* x_eval, const float* y_eval,
                       const float* n_eval, const long long* inv, const long long* true_rows, const float* rgb_sorted, int d_mask,
                       int e_mask, float* rgb_values, float* sdf_output, float* diff_pts, float* eik_out, float* points_hom, float* grad_theta,
                       float* surf, void* stream);
/* Upstream gradients dy[E+N][Nout], dn[E+N][3] of the fused SDF backward (N, n_true known on the host by now).  stage 0: zero + the
 * rendering net's input adjoint din[N][din_ld] (features from column din_feat0 -> dy[:, 2:], normals from din_nrm0 -> dn when use_geo).
 * stage 1 (after the input-adjoint pass produced dx[N][3]): SampleNetwork's scalar -(xbar . v)/(n . v), xbar = d_diff + din[:, 0:3]
 * (use_geo) + dx (sample_network.py:10-20), added to dy[E+i][0]; d_eo / d_gth / d_si (upstream of eikonal_output / grad_theta /
 * surf_indicator_output, any may be NULL) scattered over the same groups as mvsdf_step_outputs. */
int mvsdf_step_backward_inputs(int stage, int n_eik, int n_ds, int N, int Nout, int n_true, const float* din, int din_ld, int din_feat0,
                               int din_nrm0, int use_geo, const float* d_diff, const float* dx, const float* view_sorted, const float* n_eval,
                               const long long* true_rows, const float* d_eo, const float* d_gth, const float* d_si, int d_mask, int e_mask,
                               float* dy, float* dn, void* stream);
/* stage 2 of mvsdf_step_backward_inputs = stage 1 without SampleNetwork's scalar; this adds it afterwards: fbar[i] = -(xbar_i . v_i) / (n_i . v_i)
 * (sample_network.py:10-20), xbar = d_diff + din[:, 0:3] (if use_geo) + dx, written to fbar[N] and added to dy[(E + i) * Nout]. */
int mvsdf_step_backward_fbar(int n_eik, int n_ds, int N, int Nout, const float* din, int din_ld, int use_geo, const float* d_diff, const float* dx,
                             const float* view_sorted, const float* n_eval, float* dy, float* fbar, void* stream);

/* ---- phase-0 depth-surface sampling of IDRNetwork.forward (idr.py:226-247, my_utils.py:71-95) ----
 * Two uniformly random n-subsets (without replacement) of the depth pixels (depths[N][H][W] > 0) whose unprojected, normalised point
 * -- set 0 -- or jittered point (+U(-jitter_rad, jitter_rad)^3) -- set 1 -- lies inside the box |x| < bb: the kernel visits the
 * pixels in a keyed random order and unprojects only the candidates it visits.  kinv[N][3][3] / einv[N][4][4]: inverse intrinsics /
 * extrinsics of the depth cameras; size[1], center[3] on the device.  idx[2][n] (pre-filled by the caller with a value >= N*H*W)
 * receives the pixel indices in visiting order, counts[2] how many were found (n unless the image holds fewer valid pixels).
 * mvsdf_dsurf_points turns the SORTED indices (the reference sorts, np.sort) into the points pts_on[n][3], pts_jit[n][3]; same
 * seed = same jitter as during the selection. */
int mvsdf_dsurf_select(const float* depths, const float* kinv, const float* einv, int N, int H, int W, const float* size, const float* center,
                       float bb, float jitter_rad, unsigned long long seed, int n, long long* idx, long long* counts, void* stream);
int mvsdf_dsurf_points(const float* depths, const float* kinv, const float* einv, int N, int H, int W, const float* size, const float* center,
                       float bb, float jitter_rad, unsigned long long seed, int n, const long long* idx_sorted, const long long* counts,
                       float* pts_on, float* pts_jit, void* stream);

/* ---- optimiser tail on flat buffers (idr_train.py:289-302: all_norm, clip_grad_norm_(grad_cap), Adam.step), two launches ----
 * p, g, m, v: flat fp32 buffers of n elements (parameters, gradients, exp_avg, exp_avg_sq).  step >= 1 is the Adam step count AFTER
 * this update.  max_norm <= 0 disables clipping; otherwise g is scaled in place by min(1, max_norm / (||g|| + 1e-6)).
 * norm_out (device, 2 floats, may be NULL) receives {||g||, clip coefficient}; ws: mvsdf_adam_ws_floats() floats. */
size_t mvsdf_adam_ws_floats(void);
int mvsdf_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step, float max_norm,
                    float* norm_out, float* ws, void* stream);
/* Same with every gradient multiplied by grad_scale first (norm and clip see the scaled gradient): grad_scale = 1 / world size turns the
 * SUM of a data-parallel all-reduce into the rank average inside the optimiser launch -- no separate division pass over the bucket. */
int mvsdf_adam_step_scaled(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                           float max_norm, float grad_scale, float* norm_out, float* ws, void* stream);
/* ... and, with zero_grad != 0, ZEROS left in g instead of the scaled / clipped gradient: the `optimizer.zero_grad()` that opens the next iteration
 * (idr_train.py:283) then needs no launch of its own (the update pass touches every gradient element anyway). */
int mvsdf_adam_step_fused(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                          float max_norm, float grad_scale, int zero_grad, float* norm_out, float* ws, void* stream);

/* masks -> what IDRLoss.forward needs from them, one launch: hit[R] = network_object_mask & object_mask (loss.py:21,206), view_start[B+1]
 * = prefix sums of the per-view hit counts (rows of diff_surf_pts per view, loss.py:119-127; R = B * P rays, view-major),
 * n_pos = #(network_object_mask & object_mask_true) (positives of the surface-indicator BCE, loss.py:167-173). */
int mvsdf_loss_prep(const uint8_t* net_mask, const uint8_t* obj_mask, const uint8_t* true_mask, int R, int B, uint8_t* hit, int* view_start,
                    long long* n_pos, void* stream);
/* backward of the weighted total of mvsdf_loss_terms: g = HOST array of 6 device pointers to the scalar upstream gradients of {loss, rgb,
 * eikonal, depth, feat, surf} (NULL = no gradient, the usual case for all but `loss`); every unit
 * gradient tensor is scaled by (g[0] * weight + g[term]) into its output; coef_feat[1] (may be NULL) = g[0] * w_feat + g[4]. */
int mvsdf_loss_scale(const float* const* g, float w_rgb, float w_eik, float w_surf, float w_feat, float w_depth, const float* d_rgb, float* g_rgb,
                     int n_rgb, const float* d_grad, float* g_grad, int n_grad, const float* d_eo, float* g_eo, int n_eo, const float* d_sf,
                     float* g_sf, int n_sf, float* coef_feat, void* stream);

/* device self-test of the deterministic math: op 0 softplus100, 1 expneg, 2 log1p01, 3 sincos (y0=sin, y1=cos),
 * 4 div100 / div_sqrt2 (y0, y1), 5 sqrt / reciprocal (y0, y1), 6 two-wide softplus100 (y0 = f(x), y1 = f(-x)). */
int mvsdf_det_math(int op, const float* x, int n, float* y0, float* y1, void* stream);

/* ==== native step driver: one training step as a handful of calls, no host code between the launches ====
 * Replaces the host side of IDRNetwork.forward in training mode (idr.py:179-322), of IDRLoss.forward (loss.py:176-219) and of
 * `loss.backward()` through both (idr_train.py:283-287) -- the reference issues ~2 000 framework ops there; the Python mirror of this
 * library used to issue ~45 ctypes calls plus autograd glue (1.6 ms of host time per step).  Each call below enqueues ALL launches of its
 * phase on `stream` from C++; launch shapes that depend on the hit count N read it from device memory or are sized for the worst case.
 *
 * Memory: the caller owns two blocks.  `fwd` (MvsdfStepLayout.fwd_bytes, one per forward, kept until its backward has run) receives every
 * output tensor of the forward at the byte offsets of the layout and everything the backward reads (folded weights, packs, saved
 * activations); `bwd` (bwd_bytes) is scratch of the backward.  Offsets are multiples of 256 bytes.  */
#define MVSDF_STEP_MAX_LAYERS 24

typedef struct {
    int B, P;                              /* this batch: B views x P pixels, R = B * P rays (uv[B][P][2], idr.py:183-190) */
    int n_eik;                             /* eikonal sample points (idr.py:216-217: R / 2) */
    int n_ds;                              /* depth-surface samples per set (idr.py:226-247), 0 outside phase 0 */
    int n_sdf, n_render;                   /* Linear layers of ImplicitNetwork / RenderingNetwork */
    int N[MVSDF_STEP_MAX_LAYERS];          /* out features per Linear: the n_sdf SDF layers, then the n_render rendering layers */
    int K[MVSDF_STEP_MAX_LAYERS];          /* in features */
    unsigned skip_mask;                    /* bit l: the input of SDF layer l is cat([x, PE(x)]) / sqrt(2) (idr.py:86-87) */
    int multires;                          /* PE frequencies of the SDF net */
    int view_spec;                         /* multires_view | mode bits, as mvsdf_render_forward takes them */
    int trace_dtype;                       /* 0: fp32 tracing MLP; 1: bf16 packs are made each forward and the tracer uses them; 2: fp32 packs of the bf16-rounded weights; 3 / 4: bf16 packs without duplicated columns, activations as 2 / 3 bf16 terms (MvsdfNetDesc.trace_dtype) */
    int use_object_mask;                   /* conf.use_mask (idr.py:187): 0 = the ray partition ignores object_mask */
    MvsdfTraceParams tp;
    int mt, mt_samples;                    /* tiling of the tracer kernels (see mvsdf_trace) */
} MvsdfStepDesc;

typedef struct {                           /* raw parameters (device pointers), SDF layers first; g[l] NULL = no weight norm */
    const float* v[MVSDF_STEP_MAX_LAYERS];
    const float* g[MVSDF_STEP_MAX_LAYERS];
    const float* b[MVSDF_STEP_MAX_LAYERS];
} MvsdfStepParams;

typedef struct {
    const float* uv; const float* pose; const float* intrinsics;         /* [B][P][2], [B][4][4], [B][4][4] */
    const uint8_t* object_mask;            /* [R] the mask the tracer sees (all ones when conf.use_mask is off, idr.py:187) */
    const uint8_t* object_mask_true;       /* [R] input['object_mask'] */
    const float* intervals;                /* [n_steps] linspace(0, 1) (ray_tracing.py:206) */
    const float* minsdf_steps;             /* [n_steps] uniform draws of minimal_sdf_points (ray_tracing.py:287) */
    const float* eik_points;               /* [n_eik][3] uniform draws in the eikonal box (idr.py:216-221) */
    const float* ds_on; const float* ds_jit;  /* [n_ds][3] each (mvsdf_dsurf_points), NULL when n_ds == 0 */
    const long long* ds_counts;            /* [2] device: samples found per set (mvsdf_dsurf_select), NULL when n_ds == 0 */
    const float* host_stage;               /* optional: PINNED host memory (hipHostMalloc / a torch pinned tensor) holding [minsdf_steps | eik_points]
                                            * = tp.n_steps + 3 n_eik floats.  When given, minsdf_steps / eik_points are uninitialised device buffers and
                                            * the forward's first kernel fills them from here (no copy node, and no copy-to-kernel bubble, in front of
                                            * the step); the host may rewrite the memory once mvsdf_step_wait_counts has returned.  NULL: the two
                                            * device buffers already hold the draws. */
} MvsdfStepInputs;

typedef struct {
    size_t fwd_bytes, bwd_bytes;
    /* byte offsets into `fwd` of the tensors of the reference's output dict (idr.py:306-322); every N-dependent tensor is sized for N = R
     * and its first rows are valid (see mvsdf_step_outputs) */
    size_t ray_dirs, cam_loc;              /* [R][3], [B][3] */
    size_t points, mask, dists, counters;  /* tracer outputs: [R][3] f32, [R] u8, [R] f32, uint64[16] */
    size_t object_mask_out;                /* [R] u8: all ones (out['object_mask'] when conf.use_mask is off) */
    size_t rgb_values, sdf_output, diff_pts, eik_out, points_hom, grad_theta, surf;
    size_t perm;                           /* int64 [R]: sorted row -> ray (hit rays first) */
    size_t dflat;                          /* offset into `bwd`: the gradient of the folded parameters [dW | db of the SDF net | dW | db of the rendering net] */
    size_t dflat_floats;
} MvsdfStepLayout;

/* create / destroy the host-side state of a step (a pinned 4 x int64 staging buffer and HIP events; no device memory) */
int mvsdf_step_create(const MvsdfStepDesc* desc, MvsdfStepLayout* layout, void** step);
void mvsdf_step_destroy(void* step);
/* IDRNetwork.forward, training mode: fold (+ bf16 packs) -> camera rays -> RayTracing.forward -> ray partition (hit counts start travelling
 * to the pinned buffer) -> ONE fused value + normal evaluation over [sample points | rays, hit first] -> rendering net -> output gather.
 * d_mask / e_mask: point groups of the depth / eikonal terms (see mvsdf_step_outputs). */
int mvsdf_step_forward(void* step, const MvsdfStepParams* prm, const MvsdfStepInputs* in, int d_mask, int e_mask, void* fwd, void* stream);
/* blocks until the counts of the last mvsdf_step_forward are on the host: {N hit, N hit & true mask, depth-surface samples found per set x 2}.
 * The one host wait of a CLASSIC training step (the fused evaluation enqueued behind the count record keeps the GPU busy meanwhile); a DEFERRED
 * step (below) never calls it. */
int mvsdf_step_wait_counts(void* step, long long counts[4]);
/* ---- the deferred step: no host wait between the forward and the optimiser ----
 * The counts stay on the device: mvsdf_loss_forward with MvsdfLossArgs.counts_dev set and mvsdf_step_backward with N < 0 take {N, n_true} from
 * the forward block (written by the ray partition), size their launches for N = R and bound every row loop by the device values, so a training
 * loop `forward -> loss -> backward -> optimiser` enqueues whole steps ahead of the GPU and its rate no longer depends on the host's latency
 * (idr_train.py:253-315 is the loop; its per-step print is the only reader of host-side numbers).  Results are bit-identical to the classic step.
 * mvsdf_step_seq: sequence number (1, 2, ...) of the last mvsdf_step_forward of this step object.
 * mvsdf_step_counts_offset: byte offset inside `fwd` of its int64 counts[4] = {N, n_true, ds found x 2} (the `counts_dev` of that forward); 32 bytes behind
 *   it float term_rows[3] = rows of grad_theta / eikonal_output / surf_indicator_output for these counts (what a data-parallel step all-reduces to
 *   normalise the three count-based means by the global counts, MvsdfLossArgs.inv_counts).
 * mvsdf_step_wait_counts_seq: blocks until forward `seq` has delivered its counts (a short spin, then sleeps); -4 when that record was overwritten
 *   (more than MVSDF_STEP_COUNT_RING forwards ago: read the counts from the forward block instead).
 * mvsdf_step_done_seq: newest forward whose partition kernel is known to have run (non-blocking; its counts -> counts[4] when not NULL).
 * mvsdf_step_can_defer: 1 when mvsdf_step_backward accepts N < 0 for this step's networks (every launch has a fused device-count form). */
#define MVSDF_STEP_COUNT_RING 64
long long mvsdf_step_seq(void* step);
size_t mvsdf_step_counts_offset(void* step);
int mvsdf_step_wait_counts_seq(void* step, long long seq, long long counts[4]);
long long mvsdf_step_done_seq(void* step, long long counts[4]);
int mvsdf_step_can_defer(void* step);
/* byte offsets inside `fwd` of what the forward saved for the backward, for inspection (tests read the rendering net's own ReLU masks there):
 * out[6] = {x_eval [E+R][3] evaluation rows [samples | rays, hit first], y_eval [E+R][Nout], n_eval [E+R][3], view_sorted [R][3],
 * render_ctx (mvsdf_render_forward's context for R rows: the input [R][K_0], then the post-ReLU activations [R][K_l] of layers 1.., then rgb),
 * rgb_sorted [R][3]} */
int mvsdf_step_saved_offsets(void* step, size_t out[6]);
/* backward of mvsdf_step_forward: upstream gradients of diff_surf_pts [N][3], rgb_values [R][3], grad_theta, eikonal_output,
 * surf_indicator_output (any may be NULL = zero) -> gradient of every raw parameter.  N, n_true: the counts mvsdf_step_wait_counts returned;
 * N < 0: the deferred step -- both counts are read on the device from `fwd`, the upstream tensors hold their valid rows first (sized for N = R),
 * and n_true carries a HINT of N (a recent step's, or < 0 for none) that only selects kernel forms.
 * use_geo: 0 = points / normals / view directions detached in front of the rendering net (idr.py:331-334).
 * dv / dg / db: per-layer targets (device pointers; dg[l] NULL where g[l] is); accumulate != 0 ADDS into them (the parameters' .grad). */
int mvsdf_step_backward(void* step, const MvsdfStepParams* prm, int N, int n_true, int d_mask, int e_mask, int use_geo, const float* d_diff,
                        const float* d_rgb, const float* d_gth, const float* d_eo, const float* d_si, const void* fwd, void* bwd,
                        float* const* dv, float* const* dg, float* const* db, int accumulate, void* stream);
/* per-kernel timing of the tracer for bench.py's roofline: enable != 0 makes every forward record HIP events on the launch stream around
 * k_sphere_trace, the sampler launches and the secant / min-sdf launch; mvsdf_step_trace_times (after the stream was synchronised) ->
 * ms[3] = {sphere tracing, sampler rows, secant + min-sdf rows} of the last forward. */
int mvsdf_step_set_timing(void* step, int enable);
int mvsdf_step_trace_times(void* step, float ms[3]);
/* ... and of the differentiable half (idr.py:240-322 forward, its backward): ms[6] = {the three above, the forward behind the tracer (fused value +
 * normal evaluation, rendering net, output gather: no bubbles on the stream, so the event distance IS the kernel time; sample rows evaluated on the side
 * stream beside the tracer are not in it), mvsdf_step_backward, the whole mvsdf_step_forward from its first launch (fold + packs + rays) to its last} of the
 * last step. */
int mvsdf_step_times(void* step, float ms[6]);

/* ---- IDRLoss.forward / backward (loss.py:176-219) as one call each ---- */
typedef struct {
    int R, B;                              /* rays, views of this batch */
    int N, n_grad, n_depth, n_surf;        /* rows of diff_surf_pts / grad_theta / eikonal_output / surf_indicator_output */
    const uint8_t* net_mask; const uint8_t* obj_mask; const uint8_t* true_mask;      /* [R] */
    const float* rgb; const float* rgb_gt;                                           /* [R][3] */
    const float* grad_theta; const float* eik_out; const float* surf; const float* diff_pts;
    float* points_hom;                     /* [n_depth][4]: rescaled to world coordinates IN PLACE (loss.py:38,42) */
    int feat_on, surf_on;                  /* phase switches (loss.py:195-204) */
    int V, C, H, W;                        /* source views, feature channels, feature map size */
    const float* feat; long long feat_strides[4]; const float* feat_src; long long src_strides[5];
    const float* cam; const float* src_cams; const float* size; const float* center;
    const float* depths; int dB, dh, dw; const float* depth_cams;                    /* [dB][dh][dw], [dB][2][4][4] */
    float out_thresh_perc, far_thresh, far_att, near_thresh, near_att;
    float w_rgb, w_eik, w_surf, w_feat, w_depth;
    int use_invalid;                       /* conf.use_invalid: carving_t instead of carving_t2 (see mvsdf_depth_carve) */
    float smooth;                          /* depth term: 0 = L1, s > 0 = SmoothL1(eikonal_output / s, -dist_r / s) * s (loss.py:57-58: conf.smooth(train_progress)) */
    const float* inv_counts;               /* see mvsdf_loss_terms */
    /* deferred step: device pointer to {N, n_true} (int64, the forward block's counts).  N / n_grad / n_depth / n_surf above are then UPPER BOUNDS
     * (N = R) that size the block's layout and the grids; the kernels derive the true row counts from the device values and the step's point groups:
     * n_grad / n_depth = rows of the groups e_mask / d_mask select among [N hit | n_eik | n_ds | n_ds] (mvsdf_step_outputs), n_surf = n_true + n_eik. */
    const long long* counts_dev;
    int n_eik, n_ds, d_mask, e_mask;
} MvsdfLossArgs;
typedef struct {
    size_t bytes, out, hit, view_start, n_pos, loss_pp, dpts, dist_r, weight, d_rgb, d_grad, d_eo, d_sf;
    /* gradients of the TOTAL loss w.r.t. rgb_values / grad_theta / eikonal_output / surf_indicator_output / diff_surf_pts (unit gradient x term weight:
     * exactly what mvsdf_loss_backward returns for an upstream of 1 on `loss` alone), written by mvsdf_loss_forward */
    size_t s_rgb, s_grad, s_eo, s_sf, s_diff;
} MvsdfLossLayout;
/* layout of the block both calls work in (out: float[6] = loss, rgb, eikonal, depth, feat, surf) */
int mvsdf_loss_layout(const MvsdfLossArgs* a, MvsdfLossLayout* lo);
/* mask bookkeeping || depth carving, feature consistency, all terms and their unit gradients: 3 launches */
int mvsdf_loss_forward(const MvsdfLossArgs* a, void* blk, void* stream);
/* g: HOST array of 6 device pointers (upstream of the six scalars, NULL = none) -> gradients of rgb_values [R][3], grad_theta [n_grad][3],
 * eikonal_output [n_depth], surf_indicator_output [n_surf], diff_surf_pts [N][3] (any target may be NULL): one launch */
int mvsdf_loss_backward(const MvsdfLossArgs* a, const void* blk, const float* const* g, float* g_rgb, float* g_grad, float* g_eo, float* g_sf,
                        float* g_diff, void* stream);

#ifdef __cplusplus
}
#endif
#endif
