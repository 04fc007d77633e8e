// trace.hip -- RayTracing.forward (reference code/model/ray_tracing.py:27-98) as 7 launches of three kernels (k_sphere_trace, then
// k_ray_samples / k_reduce_items three times each: sampler first window, sampler rest, secant || min-sdf):
//
//   k_sphere_trace : one workgroup owns NR = 8*MT rays = 16*MT "half-rays" (start/end side).  Per-ray state
//                    lives in wave 0's registers; every round each ray requests 0/1/2 SDF evaluations, the
//                    requests are compacted with ballot + prefix-popcount into the first n rows of the LDS
//                    point tile, and the fused 9-layer MLP (tile_engine.h) runs on ceil(n/16) row tiles only --
//                    converged rays drop out and the MFMA work shrinks with them.  Rays advance independently
//                    (own iteration and line-search counters): the reference's global loop conditions
//                    (ray_tracing.py:153,176) are per-ray no-ops for finished rays, so results are identical.
//                    Also: sphere intersection (rend_util.py:141-162), left-out projection (ray_tracing.py:79-84),
//                    and appending unfinished / non-hit rays to the sample work list.
//   k_ray_samples  : ray_sampler + secant (ray_tracing.py:198-278) and minimal_sdf_points (280-308): each work item
//                    is one ray x n_steps samples; rows of RPW rays are evaluated in full tiles, then one thread per
//                    ray does the argmin / first-sign-change logic and the (dependent) secant rounds run compacted.
//
// No host synchronisation anywhere: list lengths stay on the device, grids are sized for the worst case.
#include <stdlib.h>
#include <type_traits>
#include "tile_engine.h"
#include "tile_engine_bf16.h"
#include "tile_engine_bf16s.h"
#include "trace_params.h"
#include "capi_util.h"

struct RayCommon {
    float c[3], d[3];
};

__device__ __forceinline__ float mv_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// rend_util.get_sphere_intersection for one ray (same op order as the CPU restatement)
__device__ __forceinline__ bool mv_sphere_isect(const float* c, const float* d, float r, float& t0, float& t1) {
    const float dot = fmaf(d[2], c[2], fmaf(d[1], c[1], d[0] * c[0]));
    const float nrm = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float under = dot * dot - (nrm * nrm - r * r);
    const bool hit = under > 0.0f;
    float a = 0.0f, b = 0.0f;
    if (hit) {
        const float s = sqrtf(under);
        a = s * -1.0f - dot;
        b = s * 1.0f - dot;
    }
    t0 = a < 0.0f ? 0.0f : a;
    t1 = b < 0.0f ? 0.0f : b;
    return hit;
}

// NET = MvNet (fp32 weights, fp32-input MFMA, bit-exact vs the oracle), MvNetBf (bf16 weights / activations, bf16 MFMA) or MvNetBs<NS, WT> (bf16 weights, or with WT = 3 fp32 weights as three bf16 terms;
// activations as NS bf16 terms, bf16 MFMA: tile_engine_bf16s.h): overloads of mv_sdf_eval_col0
// XR: the weight fetch of the next layer runs under this layer's work (fp32 engine: ring carried across layers, one row tile only; bf16 engine:
// the CARRIED scheme of tile_engine_bf16.h) -- for k_sphere_trace, whose evaluations wait for each other; costs registers
template <int MT, int NTW, int NW, class NET, bool XR = false>
__device__ __forceinline__ void mv_eval_dispatch(const NET& net, int ntiles, float* act, float* pe, const float* pts, float* out, int tid) {
    if constexpr (std::is_same<NET, MvNet>::value) {
        if (MT >= 4 && ntiles == 4) mv_sdf_eval_col0<(MT >= 4 ? 4 : MT), NTW, NW>(net, act, pe, pts, out, tid);
        else if (MT >= 3 && ntiles == 3) mv_sdf_eval_col0<(MT >= 3 ? 3 : MT), NTW, NW>(net, act, pe, pts, out, tid);
        else if (MT >= 2 && ntiles == 2) mv_sdf_eval_col0<(MT >= 2 ? 2 : MT), NTW, NW>(net, act, pe, pts, out, tid);
        else if constexpr (XR) mv_sdf_eval_col0<1, NTW, NW, true>(net, act, pe, pts, out, tid);
        else mv_sdf_eval_col0<1, NTW, NW>(net, act, pe, pts, out, tid);
    } else {
        // (three or four row tiles: the carried scheme's 64 weight registers on top of 32-48 accumulators / activation registers spill)
        if (MT >= 4 && ntiles == 4) mv_sdf_eval_col0<(MT >= 4 ? 4 : MT), NTW, NW, false>(net, act, pe, pts, out, tid);
        else if (MT >= 3 && ntiles == 3) mv_sdf_eval_col0<(MT >= 3 ? 3 : MT), NTW, NW, false>(net, act, pe, pts, out, tid);
        // (four column tiles per wave = 512-wide nets: the carried scheme would hold 4 of their 16 k-blocks and fetch the rest on the spot)
        else if (MT >= 2 && ntiles == 2) {
            // (the sphere tracer's two-tile workgroups -- XR, MT == 2: one per CU -- keep a second pair of tiles: mv_act_rows; not the 512-wide nets, NTW == 4: 224 KB)
            if constexpr (XR && MT == 2 && NTW < 4 && mv_bs_pp<NET>::v) mv_sdf_eval_col0_pp<2, NTW, NW, (XR && NTW < 4)>(net, act, pe, pts, out, tid);
            else mv_sdf_eval_col0<(MT >= 2 ? 2 : MT), NTW, NW, (XR && NTW < 4)>(net, act, pe, pts, out, tid);
        }
        else if constexpr (mv_bs_pp<NET>::v) mv_sdf_eval_col0_pp<1, NTW, NW, (XR && NTW < 4)>(net, act, pe, pts, out, tid);   // (second activation tile behind the first: mv_act_rows)
        else mv_sdf_eval_col0<1, NTW, NW, (XR && NTW < 4)>(net, act, pe, pts, out, tid);
    }
}

// LDS carve shared by both kernels
struct TraceLds {
    float* act; float* pe; float* pts; float* sdfv; float* sv; int* misc;
};
// rows of activation tiles a kernel keeps for `rows` evaluation rows: the engines that alternate between two tiles when they evaluate ONE row tile
// (tile_engine_bf16s.h: mv_bs_pp) need the second one even in a one-tile kernel
// (sphere: k_sphere_trace, whose two-tile evaluations alternate too)
template <class NET> __host__ __device__ constexpr int mv_act_rows(int rows, bool sphere = false) {
    return !mv_bs_pp<NET>::v ? rows : (rows < 32 ? 32 : ((sphere && rows == 32) ? 64 : rows));
}
template <class NET, bool SPHERE = false>
__device__ __forceinline__ TraceLds mv_carve(float* base, int rows, int S, int d0, int sv_floats) {
    TraceLds l;
    l.act = base;
    l.pe = l.act + mv_act_rows<NET>(rows, SPHERE) * S;
    l.pts = l.pe + ((rows * d0 + 3) & ~3);
    l.sdfv = l.pts + rows * 4;
    l.sv = l.sdfv + rows;
    l.misc = (int*)(l.sv + sv_floats);
    return l;
}

// Tail filling ("tail filling of k_sphere_trace" below): what a sphere-tracing workgroup needs to evaluate min-sdf rows once its own rays are done
struct TailCtx {
    const float* steps;               // the uniform draws of minimal_sdf_points (ray_tracing.py:287)
    float* sv;                        // min-sdf sample values [n_items][n_steps] (the second sample-value buffer of the workspace)
    int unit_rows;                    // rows per claimed unit (a multiple of the row tiles of both kernels)
    int enable, spin;                 // spin: a helper without a tile looks again a few times (every ~15 us, up to ~1 ms) before it exits
    int stop_left;                    // helpers take no new tile once at most this many workgroups still trace
    unsigned* probe;                  // optional [gridDim.x][4]: rounds, units helped, clock ticks tracing, clock ticks helping (dev)
};
// results of a workgroup's rays + work-list appends (ray_tracing.py:41-44, 73-94): the statement block k_sphere_trace runs once when its rays are done
#define MV_SPHERE_FINALIZE \
    if (tid == 0 && nrows_total) atomicAdd(&counters[MV_CNT_ROWS_SPHERE], nrows_total); \
    if (valid) { \
        bool net_mask = acc_s < acc_e; \
        const bool sampler = unf_s; \
        float dist = acc_s; \
        bool listed = false; \
        float zmin = acc_s, zmax = acc_e; \
        int kind = 0; \
        if (sampler) { listed = true; kind = MV_ITEM_SAMPLER | (om ? MV_ITEM_OM : 0); } \
        else if (training) { \
            const bool in_mask = !net_mask && om; \
            const bool out_mask = !om; \
            if (in_mask || out_mask) { \
                if (!isect) { \
                    const float dot = (d[0] * c[0] + d[1] * c[1]) + d[2] * c[2]; \
                    dist = -dot; \
                } else { \
                    listed = true; kind = MV_ITEM_MINSDF; \
                    zmin = (net_mask && out_mask) ? acc_s : t0; \
                    zmax = t1; \
                } \
            } \
        } \
        o_mask[gid] = net_mask ? 1 : 0; \
        o_dists[gid] = dist; \
        o_points[3 * (size_t)gid + 0] = c[0] + dist * d[0]; \
        o_points[3 * (size_t)gid + 1] = c[1] + dist * d[1]; \
        o_points[3 * (size_t)gid + 2] = c[2] + dist * d[2]; \
        if (listed) { \
            const bool smp = kind & MV_ITEM_SAMPLER; \
            w_zmin[gid] = zmin; \
            w_zmax[gid] = zmax; \
            const unsigned long long idx = atomicAdd(&counters[smp ? MV_CNT_N_SAMPLER : MV_CNT_N_MINSDF], 1ull); \
            (smp ? w_list : w_list_min)[idx] = gid | (kind << 28); \
            if (!smp) n_pub = 1; \
        } \
    }

template <int MT, int NTW, int NW, class NET>
__global__ __launch_bounds__(64 * NW) void k_sphere_trace(NET net, MvTraceParams tp, const float* __restrict__ cam_loc,
                                                            const float* __restrict__ dirs, const uint8_t* __restrict__ object_mask,
                                                            int R, int P, int training, float* __restrict__ o_points,
                                                            uint8_t* __restrict__ o_mask, float* __restrict__ o_dists,
                                                            float* __restrict__ w_zmin, float* __restrict__ w_zmax, int* __restrict__ w_list,
                                                            int* __restrict__ w_list_min, unsigned long long* __restrict__ counters, TailCtx tail) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, NR = 8 * MT;
    // (round 4, review item 8: the descriptor read at its uses through __builtin_amdgcn_kernarg_segment_ptr() instead of by value: SGPR spills 94 -> 88 in the
    // fp32 instance, 140 -> 74 in the bf16 one but 17-37 VGPR spills to scratch appear there: the spilled SGPRs are the ray state machine's, not the
    // descriptor's.  Not kept.)
    const long long clk0 = tail.probe ? (long long)wall_clock64() : 0;
    unsigned n_rounds = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    TraceLds lds = mv_carve<NET, (NTW < 4)>(smem, ROWS, net.S, 3 + 6 * net.multires, 0);
    int* s_n = lds.misc;

    // ---- per-ray state (threads 0..NR-1 of wave 0) ----
    const int gid = blockIdx.x * NR + tid;
    const bool valid = (tid < NR) && (gid < R);
    float c[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    float t0 = 0.f, t1 = 0.f;
    bool isect = false, om = false;
    if (valid) {
        const int b = gid / P;
        for (int i = 0; i < 3; ++i) { c[i] = cam_loc[3 * b + i]; d[i] = dirs[3 * (size_t)gid + i]; }
        isect = mv_sphere_isect(c, d, tp.r, t0, t1);
        om = object_mask[gid] != 0;
    }
    bool unf_s = isect, unf_e = isect;
    float acc_s = isect ? t0 : 0.f, acc_e = isect ? t1 : 0.f;
    float next_s = 0.f, next_e = 0.f, curr_s = 0.f, curr_e = 0.f;
    int iters = 0, k = 0, phase = isect ? 0 : 3;          // 0 init, 1 step, 2 line search, 3 done
    bool req_s = isect, req_e = isect;
    float ts = acc_s, te = acc_e;
    unsigned long long nrows_total = 0;

    // Speculative line search.  The back-off positions of a side that steps are known before its evaluation (acc -/+ 0.5*curr, then
    // -/+ 0.25*curr, -/+ 0.125*curr: ray_tracing.py:178-181), so they ride along as extra rows whenever the row tiles that run anyway have
    // free slots: a ray whose step overshoots then finishes its iteration in ONE dependent evaluation instead of up to 1 + line_step_iters.
    // Slots go first to sides that needed a back-off in their previous iteration (rays that overshoot tend to keep doing so), shallow
    // levels before deep ones.  Results are identical; speculative rows count only if the reference would have evaluated them.
    bool hard_s = false, hard_e = false, bo_s = false, bo_e = false;
    const float lss1 = 1.0f - tp.line_search_step;
    // helping mode (tail filling): once this workgroup's rays are done it evaluates min-sdf rows through the SAME evaluation call below (one
    // copy of the fused MLP in the kernel: a second call site costs 11 KB of code and the two copies evict each other from the instruction cache)
    bool helping = false;
    int n_pub = 0;                                                // this thread appended a min-sdf item
    int h_chunk = 0, h_left = 0, h_items = 0, h_nr = 0;
    long long h_svi = 0;
    unsigned h_units = 0;
    long long clk1 = 0;
    for (;;) {
        int row_s = 0, row_e = 0;
        int sr_s[3] = {-1, -1, -1}, sr_e[3] = {-1, -1, -1};       // rows of the speculative levels lvl0 + 1 .. lvl0 + 3 of each side
        int lvl0 = 0;
        if (!helping && w == 0) {
            const unsigned long long ms = __ballot(req_s), me = __ballot(req_e);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int ns = __popcll(ms), ne = __popcll(me), n = ns + ne;
            row_s = __popcll(ms & lt);
            row_e = ns + __popcll(me & lt);
            if (req_s) { float* p = lds.pts + row_s * 3; p[0] = c[0] + ts * d[0]; p[1] = c[1] + ts * d[1]; p[2] = c[2] + ts * d[2]; }
            if (req_e) { float* p = lds.pts + row_e * 3; p[0] = c[0] + te * d[0]; p[1] = c[1] + te * d[1]; p[2] = c[2] + te * d[2]; }
            int base = n, left = ((n + 15) & ~15) - n;
            lvl0 = (phase == 2) ? k + 1 : 0;                      // level of this round's mandatory request (back-offs already applied)
            const bool can = (phase == 1 || phase == 2);
            if (left > 0 && tp.line_step_iters > 0) {
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {            // 0: sides that backed off last iteration, 1: the others
#pragma unroll
                    for (int dd = 0; dd < 3; ++dd) {
                        const bool lv_ok = can && (lvl0 + dd + 1 <= tp.line_step_iters);
                        const bool cs = lv_ok && req_s && (pass == 0 ? hard_s : !hard_s);
                        const bool ce = lv_ok && req_e && (pass == 0 ? hard_e : !hard_e);
                        const unsigned long long bs = __ballot(cs), be = __ballot(ce);
                        int cnt = __popcll(bs), take = cnt < left ? cnt : left;
                        if (cs && __popcll(bs & lt) < take) sr_s[dd] = base + __popcll(bs & lt);
                        base += take; left -= take;
                        cnt = __popcll(be); take = cnt < left ? cnt : left;
                        if (ce && __popcll(be & lt) < take) sr_e[dd] = base + __popcll(be & lt);
                        base += take; left -= take;
                    }
                }
                // points of the granted slots: successive subtraction, exactly like `acc -= coef * curr` level by level
                float z = ts;
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    z = z - (lss1 / (float)(1 << (lvl0 + dd))) * curr_s;
                    if (sr_s[dd] >= 0) { float* p = lds.pts + sr_s[dd] * 3; p[0] = c[0] + z * d[0]; p[1] = c[1] + z * d[1]; p[2] = c[2] + z * d[2]; }
                }
                z = te;
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    z = z + (lss1 / (float)(1 << (lvl0 + dd))) * curr_e;
                    if (sr_e[dd] >= 0) { float* p = lds.pts + sr_e[dd] * 3; p[0] = c[0] + z * d[0]; p[1] = c[1] + z * d[1]; p[2] = c[2] + z * d[2]; }
                }
            }
            if (lane == 0) { s_n[0] = base; s_n[1] = n; }
        }
        int n = 0;
        if (!helping) {
            __syncthreads();
            n = s_n[0];
            if (n == 0) {                                         // every ray of this workgroup is done: write its results (once)
                MV_SPHERE_FINALIZE
                if (!tail.enable) break;
                // publish this workgroup's min-sdf items: ONE release per workgroup after all its appends, no acquire anywhere (an agent-scope
                // acquire invalidates the whole L2 of the XCD, i.e. the cached weights of every workgroup still tracing there: measured 44 -> 70 us
                // per round); the readers use cache-bypassing loads instead
                const int pub = (w == 0) ? __popcll(__ballot(n_pub != 0)) : 0;
                __syncthreads();
                if (tid == 0) {
                    __hip_atomic_fetch_add(&counters[MV_CNT_TAIL_READY], (unsigned long long)pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&counters[MV_CNT_TAIL_WGS], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                clk1 = tail.probe ? (long long)wall_clock64() : 0;
                helping = true;
            } else {
                ++n_rounds;
                if (tid == 0) nrows_total += (unsigned long long)s_n[1];
            }
        }
        if (helping) {
            // ---- tail filling: the next chunk of min-sdf rows (a new unit from the queue when the current one is used up)
            if (h_left == 0) {
                if (tid == 0) {
                    // A tile is claimed by fetch-add (a compare-and-swap loop hands out one tile per memory round trip with ~200 contenders: measured
                    // 150 tiles in 190 us) -- but only when it is CERTAIN to lie inside the rows that already exist: every list item published
                    // (ready == reserved, read in that order: no append in flight, so all `reserved` items are complete) and a margin of one tile per
                    // workgroup of the grid beyond the queue head (between this look and the fetch-add every other workgroup can claim at most one
                    // tile).  So a claim normally never has to wait for rows to appear and is never abandoned; the margin (<= 256 tiles of a few thousand) is
                    // simply left to the launch that follows.  The one exception -- a workgroup stalled between its look and its fetch-add (another process on
                    // the GPU) -- is the bounded-by-construction wait below: it ends when the tracing workgroups have published, and those never wait for a
                    // helper's slot because the grid fits the device (mv_tail_on: at most one workgroup per compute unit).
                    long long take = -1;
                    int n_items = 0;                              // the item count the tile was validated against (>= what it touches)
                    for (int tries = 0; tries < 64; ++tries) {
                        // no new tile once only a handful of workgroups still trace: a tile started in the kernel's last round outlives it (measured: the
                        // kernel then ends ~70 us after its slowest tracer)
                        if (__hip_atomic_load(&counters[MV_CNT_TAIL_WGS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (unsigned long long)tail.stop_left >= (unsigned long long)gridDim.x) break;
                        const long long head = (long long)__hip_atomic_load(&counters[MV_CNT_TAIL_NEXT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // a consistent look at (reserved, ready): reserved read BEFORE and AFTER ready, compiler barriers between the loads (a wave's loads
                        // return in order; relaxed atomics on different addresses may be reordered by the compiler only).  reserved only grows and ready
                        // follows it, so res0 == rdy == res1 means no append was in flight when `rdy` was read.
                        const unsigned long long res0 = __hip_atomic_load(&counters[MV_CNT_N_MINSDF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("" ::: "memory");
                        const unsigned long long rdy = __hip_atomic_load(&counters[MV_CNT_TAIL_READY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("" ::: "memory");
                        const unsigned long long res = __hip_atomic_load(&counters[MV_CNT_N_MINSDF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        long long rows = (long long)res * tp.n_steps;
                        if (res0 == res && rdy == res && head + ((long long)gridDim.x + 1) * ROWS <= rows) {
                            take = (long long)__hip_atomic_fetch_add(&counters[MV_CNT_TAIL_NEXT], (unsigned long long)ROWS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            n_items = (int)res;
                            // The margin assumes every other workgroup claims at most one tile between the look and the fetch-add.  A workgroup that was
                            // stalled longer than that (another process on the GPU) may have claimed rows that do not exist YET: it owns them all the same
                            // (the launch that follows starts behind TAIL_NEXT), so it waits until they are published or the list is final -- the tracing
                            // workgroups do not depend on any helper, so this ends -- and evaluates what exists of the tile.  Never a dropped row.
                            while (take + ROWS > rows) {
                                const unsigned long long wgs = __hip_atomic_load(&counters[MV_CNT_TAIL_WGS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                asm volatile("" ::: "memory");
                                const unsigned long long a0 = __hip_atomic_load(&counters[MV_CNT_N_MINSDF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                asm volatile("" ::: "memory");
                                const unsigned long long r1 = __hip_atomic_load(&counters[MV_CNT_TAIL_READY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                asm volatile("" ::: "memory");
                                const unsigned long long a1 = __hip_atomic_load(&counters[MV_CNT_N_MINSDF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (a0 == a1 && r1 == a1) {
                                    rows = (long long)a1 * tp.n_steps;
                                    n_items = (int)a1;
                                    if (wgs >= (unsigned long long)gridDim.x) break;      // every workgroup has appended: the list is final
                                }
                                __builtin_amdgcn_s_sleep(127);
                            }
                            if (take >= rows) take = -1;                                   // (claimed behind the end of the final list: nothing there)
                            break;
                        }
                        if (!tail.spin) break;
                        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);   // ~15 us
                    }
                    s_n[0] = (int)take; s_n[1] = n_items;
                }
                __syncthreads();
                const int take = s_n[0];
                if (take < 0) break;
                h_items = s_n[1];
                h_left = 1;
                h_chunk = take;                                   // first row of the claimed tile(s)
                ++h_units;
            }
            {   // rows h_chunk .. h_chunk + ROWS - 1 of the queue (as mv_eval_rows builds them for a min-sdf segment: ray_tracing.py:287-297);
                // list entry and range were written by another workgroup during this kernel: cache-bypassing loads
                const long long total = (long long)h_items * tp.n_steps, q0 = (long long)h_chunk;
                h_nr = (int)min((long long)ROWS, total - q0);
                if (tid < ROWS) {
                    float* p = lds.pts + tid * 3;
                    if (tid < h_nr) {
                        const long long q = q0 + tid;
                        const int it = (int)(q / tp.n_steps), i = (int)(q - (long long)it * tp.n_steps);
                        const int g2 = __hip_atomic_load(&w_list_min[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x0fffffff;
                        const float* cc = cam_loc + 3 * (g2 / P);
                        const float* dd = dirs + 3 * (size_t)g2;
                        const float zmin = __hip_atomic_load(&w_zmin[g2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float zmax = __hip_atomic_load(&w_zmax[g2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float z = tail.steps[i] * (zmax - zmin) + zmin;          // ray_tracing.py:290
                        p[0] = cc[0] + z * dd[0]; p[1] = cc[1] + z * dd[1]; p[2] = cc[2] + z * dd[2];
                        h_svi = (long long)it * tp.n_steps + i;
                    } else { p[0] = 0.f; p[1] = 0.f; p[2] = 0.f; }
                }
                n = h_nr;
                --h_left;
            }
            __syncthreads();
        }
        mv_eval_dispatch<MT, NTW, NW, NET, (MT == 1 || (MT <= 2 && !std::is_same<NET, MvNet>::value))>(net, (n + 15) >> 4, lds.act, lds.pe, lds.pts, lds.sdfv, tid);
        if (helping) {
            if (tid < h_nr) tail.sv[h_svi] = lds.sdfv[tid];
            __syncthreads();                                      // values read before the next chunk's points overwrite the tile
            continue;
        }
        if (w == 0) {
            int used = 0;
            if (phase != 3) {
                const float vs = req_s ? mv_clamp(lds.sdfv[row_s], -tp.dist_clip, tp.dist_clip) : 0.f;
                const float ve = req_e ? mv_clamp(lds.sdfv[row_e], -tp.dist_clip, tp.dist_clip) : 0.f;
                const bool had_s = req_s, had_e = req_e;          // sides whose speculative rows (if any) belong to this round
                bool end_iter = false;
                if (phase == 0) { next_s = vs; next_e = ve; }
                else {
                    if (phase == 1) { next_s = vs; next_e = ve; k = 0; bo_s = false; bo_e = false; }
                    else { if (req_s) next_s = vs; if (req_e) next_e = ve; k++; }
                    for (;;) {                                                        // ray_tracing.py:173-191
                        const bool np_s = next_s < 0.f, np_e = next_e < 0.f;
                        if (!(k < tp.line_step_iters && (np_s || np_e))) { end_iter = true; break; }
                        const float coef = lss1 / (float)(1 << k);
                        if (np_s) { acc_s -= coef * curr_s; ts = acc_s; bo_s = true; }
                        if (np_e) { acc_e += coef * curr_e; te = acc_e; bo_e = true; }
                        const int dd = k - lvl0;                                      // level k + 1 = speculative level index dd of this round
                        const int rs = (np_s && had_s && dd >= 0 && dd < 3) ? (dd == 0 ? sr_s[0] : (dd == 1 ? sr_s[1] : sr_s[2])) : -1;
                        const int re = (np_e && had_e && dd >= 0 && dd < 3) ? (dd == 0 ? sr_e[0] : (dd == 1 ? sr_e[1] : sr_e[2])) : -1;
                        if (rs >= 0) { next_s = mv_clamp(lds.sdfv[rs], -tp.dist_clip, tp.dist_clip); used++; }
                        if (re >= 0) { next_e = mv_clamp(lds.sdfv[re], -tp.dist_clip, tp.dist_clip); used++; }
                        req_s = np_s && rs < 0; req_e = np_e && re < 0;
                        if (req_s || req_e) { phase = 2; break; }                     // evaluate the missing side(s) at level k + 1 next round
                        k++;
                    }
                    if (end_iter) {
                        unf_s = unf_s && (acc_s < acc_e);                             // ray_tracing.py:193-194
                        unf_e = unf_e && (acc_s < acc_e);
                        hard_s = bo_s; hard_e = bo_e;
                    }
                }
                if (phase == 0 || end_iter) {                                         // top of the while loop, ray_tracing.py:139-171
                    curr_s = unf_s ? next_s : 0.f;
                    curr_e = unf_e ? next_e : 0.f;
                    if (curr_s <= tp.thr) curr_s = 0.f;
                    if (curr_e <= tp.thr) curr_e = 0.f;
                    unf_s = unf_s && (curr_s > tp.thr);
                    unf_e = unf_e && (curr_e > tp.thr);
                    if ((!unf_s && !unf_e) || iters == tp.st_iters) {
                        phase = 3; req_s = false; req_e = false;
                    } else {
                        iters++;
                        acc_s = acc_s + curr_s;
                        acc_e = acc_e - curr_e;
                        req_s = unf_s; req_e = unf_e; ts = acc_s; te = acc_e;
                        phase = 1;
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) used += __shfl_xor(used, o);             // speculative rows the reference would have evaluated
            if (lane == 0) nrows_total += (unsigned long long)used;
        }
        // (mv_sdf_eval_col0 ended with a barrier; wave 0 rewrites pts/s_n only after its own reads above)
    }

    if (tail.enable) {
        if (tid == 0 && h_units) atomicAdd(&counters[MV_CNT_TAIL_ROWS], (unsigned long long)h_units * (unsigned long long)ROWS);
        if (tail.probe && tid == 0) {
            unsigned* pr = tail.probe + 4 * blockIdx.x;
            pr[0] = n_rounds; pr[1] = h_units; pr[2] = (unsigned)(clk1 - clk0); pr[3] = (unsigned)((long long)wall_clock64() - clk1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct SampleCtx {
    const float* cam_loc; const float* dirs; int R, P, training, RPW;
    const float* intervals; const float* steps;
    float* o_points; uint8_t* o_mask; float* o_dists;
    const float* w_zmin; const float* w_zmax;
    float* sec_state;                 // [4][R]: z_low, z_high, sdf_low, sdf_high of secant rays
    int* sec_list;                    // gids of rays that need the secant
    float* sv;                        // [n_items * n_steps] sample values, row = item index on its ORIGINAL list
    int* list_rest; int* src_rest;    // sampler rays the first sample window left open: list entry, sv row
    int n_first;                      // size of the first sampler window (n_steps: single pass)
    unsigned long long* counters;
};

// one segment of sample rows: the samples [i0, i0 + ni) of every item of a device-side list
struct RowSeg { const int* list; const int* src; int cnt_index, i0, ni, blocks; int unit_rows; };   // unit_rows > 0: the rows are a queue, each workgroup claims its 16*MT rows from counters[MV_CNT_TAIL_NEXT]

// Sample rows of a work list, FLATTENED over rays: global row q = item * ni + j (sample i0 + j).  A workgroup evaluates one chunk
// of 16*MT consecutive rows (always full tiles, evenly spread over the chip) and stores the SDF values; the per-ray logic runs
// in k_reduce_items.  Sampler rays: ray_tracing.py:206-219; min-sdf rays: ray_tracing.py:287-301.
template <int MT, int NTW, int NW, class NET>
__device__ void mv_eval_rows(const NET& net, const MvTraceParams& tp, const SampleCtx& c, const RowSeg& sg, long long q0, float* smem, int n_list) {
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x;
    const int n_steps = tp.n_steps, ni = sg.ni;
    const long long total = (long long)n_list * ni;
    if (q0 >= total) return;
    const int nr = (int)min((long long)ROWS, total - q0);
    TraceLds lds = mv_carve<NET>(smem, ROWS, net.S, 3 + 6 * net.multires, 0);
    long long svi = 0;
    if (tid < ROWS) {
        float* p = lds.pts + tid * 3;
        if (tid < nr) {
            const long long q = q0 + tid;
            const int it = (int)(q / ni), i = sg.i0 + (int)(q - (long long)it * ni);
            const int e = sg.list[it];
            const int gid = e & 0x0fffffff;
            const bool samp = (e >> 28) & MV_ITEM_SAMPLER;
            const float* cc = c.cam_loc + 3 * (gid / c.P);
            const float* d = c.dirs + 3 * (size_t)gid;
            const float zmin = c.w_zmin[gid], zmax = c.w_zmax[gid];
            const float z = samp ? (zmin + c.intervals[i] * (zmax - zmin))        // ray_tracing.py:208
                                 : (c.steps[i] * (zmax - zmin) + zmin);            // ray_tracing.py:290
            p[0] = cc[0] + z * d[0]; p[1] = cc[1] + z * d[1]; p[2] = cc[2] + z * d[2];
            svi = (long long)(sg.src ? sg.src[it] : it) * n_steps + i;
        } else { p[0] = 0.f; p[1] = 0.f; p[2] = 0.f; }
    }
    __syncthreads();
    mv_eval_dispatch<MT, NTW, NW>(net, (nr + 15) >> 4, lds.act, lds.pe, lds.pts, lds.sdfv, tid);
    if (tid < nr) c.sv[svi] = lds.sdfv[tid];
}

// Per-ray reduction of the stored sample values (one thread per listed ray).
//   mode 0: sampler rays after their FIRST window of n_first samples.  A ray inside the object mask whose first negative sample
//           lies in the window at index >= 1 is settled: ray_tracing.py:221-256 reads only sdf_val[ind - 1] and sdf_val[ind] of it
//           (mask, secant hand-off), whatever the later samples are.  Every other ray goes on the rest list.
//   mode 1: sampler rays of the rest list, all n_steps values present: first sign change / P_out argmin / secant hand-off.
//   mode 2: min-sdf rays: argmin (ray_tracing.py:303-307).
// One WAVE per listed ray: lane l holds the samples l, l + 64, ...; ballots find the first negative / zero sample, a shuffle
// reduction the first minimum (same results as the serial scans of the reference: argmin returns the first minimal index).
__device__ __forceinline__ int mv_first_where(const float* __restrict__ sv, int n, int lane, int what) {   // what 0: v < 0, 1: sign(v) == 0
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const float v = i < n ? sv[i] : 1.0f;
        const unsigned long long m = __ballot(what == 0 ? (v < 0.f) : (!(v > 0.f) && !(v < 0.f)));   // sign() of a NaN counts as 0
        if (m) return base + __builtin_ctzll(m);
    }
    return -1;
}
__device__ __forceinline__ int mv_first_argmin(const float* __restrict__ sv, int n, int lane) {
    float bv = INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) { const float v = sv[i]; if (v < bv) { bv = v; bi = i; } }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    return bi == 0x7fffffff ? 0 : bi;                           // nothing below +inf: the serial scan keeps index 0
}

// Sixteen waves (= sixteen listed rays) per workgroup: the appends to the rest / secant lists are ONE atomic per list and workgroup, the waves take
// consecutive slots behind it (a wave per workgroup and an atomic per ray serialised ~1000 same-address atomics at the c5 share: 21 us of a launch
// that otherwise takes 5).  The order of a list's entries was arrival order before and is arbitrary still: every consumer works per ray.
#define MV_RED_WAVES 16
__global__ __launch_bounds__(64 * MV_RED_WAVES) void k_reduce_items(MvTraceParams tp, SampleCtx c, const int* __restrict__ list, const int* __restrict__ src,
                                                                  int cnt_index, int mode) {
    __shared__ int s_cat[MV_RED_WAVES];
    __shared__ unsigned long long s_base[2];
    const int n_list = (int)c.counters[cnt_index];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int it = blockIdx.x * MV_RED_WAVES + w;
    if (blockIdx.x * MV_RED_WAVES >= n_list) return;               // workgroup-uniform
    const bool active = it < n_list, lead = lane == 0;
    const int n_steps = tp.n_steps;
    int cat = 0;                                                  // 1: append (e, it) to the rest list; 2: append gid to the secant list
    int e = 0, gid = 0;
    bool have_dist = false;
    float dist = 0.f;
    const float* cc = nullptr; const float* d = nullptr;
    if (active) {
        e = list[it];
        gid = e & 0x0fffffff;
        const int kind = e >> 28;
        const bool om = kind & MV_ITEM_OM;
        cc = c.cam_loc + 3 * (gid / c.P);
        d = c.dirs + 3 * (size_t)gid;
        const float zmin = c.w_zmin[gid], zmax = c.w_zmax[gid];
        const int row = src ? src[it] : it;
        const float* sv = c.sv + (size_t)row * n_steps;
        if (mode == 0 && c.n_first < n_steps) {
            if (it == 0 && lead) {
                atomicAdd(&c.counters[MV_CNT_ROWS_SAMPLER], (unsigned long long)n_list * (unsigned long long)n_steps);
                atomicAdd(&c.counters[MV_CNT_ROWS_SAMPLER_EVAL], (unsigned long long)n_list * (unsigned long long)c.n_first);
            }
            const int ind = mv_first_where(sv, c.n_first, lane, 0);
            if (!(om && ind >= 1)) {                                                   // open: needs the other samples (ind == 0 wraps to the last one)
                cat = 1;
            } else if (lead) {
                dist = zmin + c.intervals[ind] * (zmax - zmin);
                have_dist = true;
                c.o_mask[gid] = 1;
                cat = 2;
                c.sec_state[gid] = zmin + c.intervals[ind - 1] * (zmax - zmin);
                c.sec_state[(size_t)c.R + gid] = dist;
                c.sec_state[2 * (size_t)c.R + gid] = sv[ind - 1];
                c.sec_state[3 * (size_t)c.R + gid] = sv[ind];
            }
        } else if (mode <= 1) {
            if (it == 0 && lead) {
                if (mode == 0) {
                    atomicAdd(&c.counters[MV_CNT_ROWS_SAMPLER], (unsigned long long)n_list * (unsigned long long)n_steps);
                    atomicAdd(&c.counters[MV_CNT_ROWS_SAMPLER_EVAL], (unsigned long long)n_list * (unsigned long long)n_steps);
                } else
                    atomicAdd(&c.counters[MV_CNT_ROWS_SAMPLER_EVAL], (unsigned long long)n_list * (unsigned long long)(n_steps - c.n_first));
            }
            // argmin(sign(sdf) * [n..1]) (ray_tracing.py:221-222), first minimum: the first negative sample, else the first exact zero, else the last
            int ind = mv_first_where(sv, n_steps, lane, 0);
            if (ind < 0) ind = mv_first_where(sv, n_steps, lane, 1);
            if (ind < 0) ind = n_steps - 1;
            const bool net_surf = sv[ind] < 0.f;
            const int i2 = (om && net_surf) ? 0 : mv_first_argmin(sv, n_steps, lane);   // P_out: argmin sdf, ray_tracing.py:229-235
            if (lead) {
                dist = zmin + c.intervals[(om && net_surf) ? ind : i2] * (zmax - zmin);
                have_dist = true;
                c.o_mask[gid] = net_surf ? 1 : 0;                                      // ray_tracing.py:237-239, 61
                const bool do_secant = c.training ? (net_surf && om) : net_surf;       // ray_tracing.py:242
                if (do_secant) {
                    int lo = ind - 1; if (lo < 0) lo += n_steps;                      // negative index wraps
                    cat = 2;
                    c.sec_state[gid] = zmin + c.intervals[lo] * (zmax - zmin);
                    c.sec_state[(size_t)c.R + gid] = zmin + c.intervals[ind] * (zmax - zmin);
                    c.sec_state[2 * (size_t)c.R + gid] = sv[lo];
                    c.sec_state[3 * (size_t)c.R + gid] = sv[ind];
                }
            }
        } else {
            if (it == 0 && lead) atomicAdd(&c.counters[MV_CNT_ROWS_MINSDF], (unsigned long long)n_list * (unsigned long long)n_steps);
            const int bi = mv_first_argmin(sv, n_steps, lane);                         // min over the shared random steps
            if (lead) { dist = c.steps[bi] * (zmax - zmin) + zmin; have_dist = true; }
        }
        if (lead && have_dist) {
            c.o_dists[gid] = dist;                                                     // secant rays are overwritten by the secant stage
            c.o_points[3 * (size_t)gid + 0] = cc[0] + dist * d[0];
            c.o_points[3 * (size_t)gid + 1] = cc[1] + dist * d[1];
            c.o_points[3 * (size_t)gid + 2] = cc[2] + dist * d[2];
        }
    }
    if (mode == 2) return;                                        // no lists behind the min-sdf rays
    // ---- the appends: one atomic per list and workgroup
    if (lead) s_cat[w] = cat;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned n1 = 0, n2 = 0;
        for (int k = 0; k < MV_RED_WAVES; ++k) { n1 += s_cat[k] == 1; n2 += s_cat[k] == 2; }
        if (n1) s_base[0] = atomicAdd(&c.counters[MV_CNT_N_SAMPLER_REST], (unsigned long long)n1);
        if (n2) s_base[1] = atomicAdd(&c.counters[MV_CNT_N_SECANT], (unsigned long long)n2);
    }
    __syncthreads();
    if (lead && cat) {
        unsigned rank = 0;
        for (int k = 0; k < w; ++k) rank += s_cat[k] == cat;
        const unsigned long long k = s_base[cat - 1] + rank;
        if (cat == 1) { c.list_rest[k] = e; c.src_rest[k] = it; }
        else c.sec_list[k] = gid;
    }
}

// secant (ray_tracing.py:260-278) for 16*MT listed rays per workgroup: n_secant dependent rounds, every round one evaluation of
// all the workgroup's rays (rows are full tiles instead of one or two rows per workgroup).
template <int MT, int NTW, int NW, class NET>
__device__ void mv_secant_rays(const NET& net, const MvTraceParams& tp, const SampleCtx& c, int n_list, int block, float* smem) {
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x;
    const int r0 = block * ROWS;
    if (r0 >= n_list) return;
    const int n = min(ROWS, n_list - r0);
    TraceLds lds = mv_carve<NET>(smem, ROWS, net.S, 3 + 6 * net.multires, 0);
    const bool mine = tid < n;
    int gid = 0;
    float cc[3] = {0, 0, 0}, d[3] = {0, 0, 0}, z_low = 0, z_high = 0, sdf_low = 0, sdf_high = 1, z_pred = 0;
    if (mine) {
        gid = c.sec_list[r0 + tid];
        const int b = gid / c.P;
        for (int i = 0; i < 3; ++i) { cc[i] = c.cam_loc[3 * b + i]; d[i] = c.dirs[3 * (size_t)gid + i]; }
        z_low = c.sec_state[gid]; z_high = c.sec_state[(size_t)c.R + gid];
        sdf_low = c.sec_state[2 * (size_t)c.R + gid]; sdf_high = c.sec_state[3 * (size_t)c.R + gid];
        z_pred = -sdf_low * (z_high - z_low) / (sdf_high - sdf_low) + z_low;
    }
    for (int it = 0; it < tp.n_secant; ++it) {
        if (tid < ROWS) { float* p = lds.pts + tid * 3; p[0] = cc[0] + z_pred * d[0]; p[1] = cc[1] + z_pred * d[1]; p[2] = cc[2] + z_pred * d[2]; }
        __syncthreads();
        mv_eval_dispatch<MT, NTW, NW>(net, (n + 15) >> 4, lds.act, lds.pe, lds.pts, lds.sdfv, tid);
        if (mine) {
            const float sm = lds.sdfv[tid];
            if (sm > 0.f) { z_low = z_pred; sdf_low = sm; }
            if (sm < 0.f) { z_high = z_pred; sdf_high = sm; }
            z_pred = -sdf_low * (z_high - z_low) / (sdf_high - sdf_low) + z_low;
        }
    }
    if (mine) {
        c.o_dists[gid] = z_pred;
        c.o_points[3 * (size_t)gid + 0] = cc[0] + z_pred * d[0];
        c.o_points[3 * (size_t)gid + 1] = cc[1] + z_pred * d[1];
        c.o_points[3 * (size_t)gid + 2] = cc[2] + z_pred * d[2];
    }
    if (tid == 0) atomicAdd(&c.counters[MV_CNT_ROWS_SECANT], (unsigned long long)n * (unsigned long long)tp.n_secant);
}

// ---- tail filling of k_sphere_trace ----
// The min-sdf rows are a queue of units of `unit_rows` consecutive rows of the flattened (list item, sample) space, claimed through
// counters[MV_CNT_TAIL_NEXT]: first by sphere-tracing workgroups whose rays are done (here), then by the min-sdf workgroups of the last
// k_ray_samples launch, which take whatever is left.  A unit is handed out only when every list item it touches is completely written:
// items are appended by atomics on counters[MV_CNT_N_MINSDF] (reserved) and counted again in counters[MV_CNT_TAIL_READY] once entry and
// range are visible; ready == reserved (read in that order) means no append is in flight, i.e. all `reserved` items are complete.
// Only FULL units are taken while other workgroups still trace (the list may still grow); helpers stop as soon as the last workgroup is
// done -- the launch that follows evaluates the rest on the whole chip.  Values are written to the min-sdf sample buffer exactly as that
// launch would (same engine arithmetic: bit-identical).
// The first sec_blocks workgroups run secant chains, the others evaluate sample rows of up to two row segments: the dependent
// secant chains of a few dozen workgroups overlap with the throughput-shaped sampling.
// (three weight terms: the 16 / 32-row shapes are held to 128 registers = two workgroups per CU like every other engine of the family)
template <class NET> struct mv_net_wt { static constexpr int v = 1; };
template <int NS, int WT> struct mv_net_wt<MvNetBs<NS, WT>> { static constexpr int v = WT; };
template <int MT, int NTW, int NW, class NET>
__global__ __launch_bounds__(64 * NW, (mv_net_wt<NET>::v == 3 && NTW == 2 && MT <= 2 && NW == 8) ? 4 : NW / 4) void k_ray_samples(NET net, MvTraceParams tp, SampleCtx c, RowSeg s0, RowSeg s1, int sec_blocks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int b = blockIdx.x;
    if (b < sec_blocks) {
        // (s_setprio 3 for these chain waves: no change with the fp32 engine (round 2: one datapath) nor with the bf16 family (round 4: c5share bf16x2
        // k_ray_samples 0.436-0.441 ms either way))
        mv_secant_rays<(MT > 1 ? 1 : MT), NTW, NW, NET>(net, tp, c, (int)c.counters[MV_CNT_N_SECANT], b, smem);
        return;
    }
    b -= sec_blocks;
    // one call site for the row evaluation (a second one doubles the kernel's code: the 32-row fp32 instantiation would no longer fit the
    // instruction cache): pick segment and chunk range first
    const bool first = b < s0.blocks;
    if (!first && b - s0.blocks >= s1.blocks) return;
    const RowSeg sg = first ? s0 : s1;
    long long q0 = (long long)(first ? b : b - s0.blocks) * (16 * MT);
    const int n_list = (int)c.counters[sg.cnt_index];
    if (sg.unit_rows > 0) {                                      // the row queue k_sphere_trace's finished workgroups already served (tail filling)
        if (q0 >= (long long)n_list * sg.ni + 16 * MT) return;     // more workgroups than tiles in the whole queue: no claim (one atomic per
                                                                   // workgroup of the worst-case grid costs ~50 us on one address)
        __shared__ long long s_q0;
        if (threadIdx.x == 0) s_q0 = (long long)atomicAdd(&c.counters[MV_CNT_TAIL_NEXT], (unsigned long long)(16 * MT));
        __syncthreads();
        q0 = s_q0;
    }
    mv_eval_rows<MT, NTW, NW, NET>(net, tp, c, sg, q0, smem, n_list);
}
// ---------------------------------------------------------------------------------------------------------------
// tail filling switches: MVSDF_TAIL=0 turns it off, MVSDF_TAIL=2 also enables it for the bf16 engine (measured slower there: DESIGN.md); MVSDF_TAIL_PROBE=1 makes mvsdf_trace_tail_probe() return per-workgroup records
static int mv_tail_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MVSDF_TAIL"); v = e ? atoi(e) : 1; }
    return v;
}
static unsigned* g_tail_probe = nullptr;
static unsigned* mv_tail_probe() { return g_tail_probe; }
// dev hook (tools/tail_probe.py): device buffer [workgroups][4] that k_sphere_trace fills with {rounds, units helped, ticks tracing, ticks helping}
extern "C" void mv_trace_set_tail_probe(unsigned* buf) { g_tail_probe = buf; }
// the min-sdf rows' own sample-value buffer: the second [R][n_steps] buffer of the workspace (behind the sampler's buffer and the rest lists)
static float* mv_minsdf_sv(float* ws, int R, int n_steps) {
    float* sec_state = ws + 2 * (size_t)R;
    int* w_list = (int*)(sec_state + 4 * (size_t)R);
    float* sv = (float*)(w_list + 3 * (size_t)R);
    int* list_rest = (int*)(sv + (size_t)R * n_steps);
    return (float*)(list_rest + 2 * (size_t)R);
}

// is the tail filling on for this call?  (mt1: row tiles per sphere-tracing workgroup.)  Only for grids of <= 256 workgroups (one per CU): with more,
// a finished workgroup's slot is wanted by a tracing workgroup that has not started yet -- helping would delay it.
static int mv_cu_count() {                                         // compute units of the current device (256 on an unpartitioned MI355X)
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 1; }
        n = v;
    }
    return n;
}
// The bound is also what makes the helpers' one wait safe (a helper that over-claimed a tile sleeps until the rows it owns are published, see the claim
// loop in k_sphere_trace): with at most one workgroup per compute unit OF THIS DEVICE every workgroup of the grid can be resident at once, so the tracing
// workgroups a waiting helper depends on never wait for its slot.  (A partitioned device reports fewer compute units and gets no tail filling at c2.)
template <class NET>
static bool mv_tail_on(int training, const float* steps, int R, int mt1) {
    constexpr bool is_bf = !std::is_same<NET, MvNet>::value;
    const int grid1 = (R + 8 * mt1 - 1) / (8 * mt1);
    // on by default for the fmaf-chain engine and -- above 2048 rays -- for the three-weight-term engine 'f32x3' (round 6, three alternating runs each: c3 4.012 -> 3.961 ms,
    // c5 share 2.270 -> 2.238; c2 1.470 vs 1.473: no difference, left off); the bf16-weight engines lose (c5 share bf16x2 1.506 -> 1.545): MVSDF_TAIL=2 only
    const int need = !is_bf ? 1 : ((mv_net_wt<NET>::v == 3 && R > 2048) ? 1 : 2);
    return training && steps && mv_tail_mode() >= need && grid1 <= mv_cu_count();
}

template <class NET>
static size_t trace_lds_bytes(const NET& net, int MT, int sv_floats, int rpw, bool sphere = false) {
    const int rows = 16 * MT, d0 = 3 + 6 * net.multires;
    size_t f = (size_t)mv_act_rows<NET>(rows, sphere) * net.S + ((rows * d0 + 3) & ~3) + rows * 4 + rows + sv_floats;
    return f * 4 + 16 + (size_t)rpw * (8 * 4 + 4) + 16;
}

template <int MT, int NTW, int NW, class NET>
static hipError_t launch_stage1(const NET& net, const MvTraceParams& tp, const float* cam_loc, const float* dirs, const uint8_t* om, int B, int P,
                                int training, const float* steps, int mt2, float* points, uint8_t* mask, float* dists, float* ws,
                                unsigned long long* counters, hipStream_t stream) {
    const int R = B * P, NR = 8 * MT;
    float* w_zmin = ws;
    float* w_zmax = w_zmin + R;
    int* w_list = (int*)(w_zmax + 5 * (size_t)R);
    int* w_list_min = w_list + R;
    TailCtx tail;
    memset(&tail, 0, sizeof(tail));
    const int grid1 = (R + NR - 1) / NR;
    // fp32 engine only: with the bf16 engine a tile takes half the time, the launch that follows is bound by its secant chains and the helpers
    // cost the sphere kernel more than they save (measured: c2 +10 us, c5share +20 us per step)
    if (mv_tail_on<NET>(training, steps, R, MT)) {
        tail.enable = 1;
        tail.steps = steps;
        tail.sv = mv_minsdf_sv(ws, R, tp.n_steps);
        tail.unit_rows = 16 * MT;
        tail.spin = 1;
        tail.probe = mv_tail_probe();
        static int stop_env = -1;
        if (stop_env < 0) { const char* e = mv_dev_env("MVSDF_TAIL_STOP"); stop_env = e ? atoi(e) : -1; }
        tail.stop_left = stop_env >= 0 ? stop_env : grid1 / 4;      // (swept at c2: 0 / 16 / 32 / 48 / 64 of 256 -> tracer 1575 / 1527 / 1521 / 1507 / 1507 us, off: 1549)
    }
    const size_t lds1 = trace_lds_bytes(net, MT, 0, 0, NTW < 4);
    static size_t set1 = 0;                                     // raise the dynamic-LDS cap once per size (per instantiation)
    if (lds1 > set1) {
        hipError_t e = hipFuncSetAttribute((const void*)k_sphere_trace<MT, NTW, NW, NET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return e;
        set1 = lds1;
    }
    hipLaunchKernelGGL((k_sphere_trace<MT, NTW, NW, NET>), dim3(grid1), dim3(64 * NW), lds1, stream, net, tp, cam_loc, dirs, om, R, P,
                       training, points, mask, dists, w_zmin, w_zmax, w_list, w_list_min, counters, tail);
    return hipGetLastError();
}

template <int MT, int NTW, int NW, class NET>
static hipError_t launch_stage2(const NET& net, const MvTraceParams& tp, const float* cam_loc, const float* dirs, int B, int P, int training,
                                const float* intervals, const float* steps, float* points, uint8_t* mask, float* dists, float* ws,
                                unsigned long long* counters, int parts, int mt1, hipStream_t stream) {
    // parts bit 0: sampler rows + their reduction (the hit mask is FINAL after it); bit 1: secant + min-sdf rows in one launch;
    // bit 2: min-sdf rows + their reduction alone (own sample-value buffer: may run concurrently with bit 0 on another stream); bit 3: secant alone
    const int R = B * P, ROWS = 16 * MT;
    float* w_zmin = ws;
    float* w_zmax = w_zmin + R;
    float* sec_state = w_zmax + R;                               // [4][R]
    int* w_list = (int*)(sec_state + 4 * (size_t)R);
    int* w_list_min = w_list + R;
    int* sec_list = w_list_min + R;
    float* sv = (float*)(sec_list + R);                          // [R * n_steps]
    const size_t lds2 = trace_lds_bytes(net, MT, 0, 0);
    static size_t set2 = 0;
    if (lds2 > set2) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ray_samples<MT, NTW, NW, NET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
        set2 = lds2;
    }
    static int nf_env = -1;
    if (nf_env < 0) { const char* e = mv_dev_env("MVSDF_NFIRST"); nf_env = e ? atoi(e) : 12; if (nf_env < 2) nf_env = 2; }
    const int n = tp.n_steps, nf = nf_env < n ? nf_env : n;
    SampleCtx c;
    c.cam_loc = cam_loc; c.dirs = dirs; c.R = R; c.P = P; c.training = training; c.RPW = 0; c.intervals = intervals; c.steps = steps;
    c.o_points = points; c.o_mask = mask; c.o_dists = dists; c.w_zmin = w_zmin; c.w_zmax = w_zmax; c.sec_state = sec_state;
    c.sec_list = sec_list; c.sv = sv; c.counters = counters;
    c.list_rest = (int*)(sv + (size_t)R * n); c.src_rest = c.list_rest + R; c.n_first = nf;
    // worst-case grids (every ray listed); blocks beyond the device-side counts exit at once
    auto blocks_for = [&](int per_item) { return (int)(((long long)R * per_item + ROWS - 1) / ROWS); };
    const int sec_blocks = (R + 15) / 16, red_blocks = (R + MV_RED_WAVES - 1) / MV_RED_WAVES;   // one wave per listed ray, 16 per workgroup
    const RowSeg none = {nullptr, nullptr, 0, 0, 1, 0, 0};
    if (parts & 1) {
        // sampler rays: first window of nf samples, then the other samples of the rays the window left open
        const RowSeg first = {w_list, nullptr, (int)MV_CNT_N_SAMPLER, 0, nf, blocks_for(nf), 0};
        hipLaunchKernelGGL((k_ray_samples<MT, NTW, NW, NET>), dim3(first.blocks), dim3(64 * NW), lds2, stream, net, tp, c, first, none, 0);
        hipLaunchKernelGGL(k_reduce_items, dim3(red_blocks), dim3(64 * MV_RED_WAVES), 0, stream, tp, c, w_list, (const int*)nullptr, (int)MV_CNT_N_SAMPLER, 0);
        if (nf < n) {
            const RowSeg rest = {c.list_rest, c.src_rest, (int)MV_CNT_N_SAMPLER_REST, nf, n - nf, blocks_for(n - nf), 0};
            hipLaunchKernelGGL((k_ray_samples<MT, NTW, NW, NET>), dim3(rest.blocks), dim3(64 * NW), lds2, stream, net, tp, c, rest, none, 0);
            hipLaunchKernelGGL(k_reduce_items, dim3(red_blocks), dim3(64 * MV_RED_WAVES), 0, stream, tp, c, c.list_rest, c.src_rest, (int)MV_CNT_N_SAMPLER_REST, 1);
        }
    }
    if (parts & 2) {
        // with tail filling the min-sdf rows are a queue that k_sphere_trace's finished workgroups have already served: own value buffer,
        // units claimed dynamically (the grid stays worst-case: workgroups without a unit exit at once)
        const bool tail = mv_tail_on<NET>(training, steps, R, mt1);
        SampleCtx cm = c;
        if (tail) cm.sv = mv_minsdf_sv(ws, R, n);
        const RowSeg minsdf = {w_list_min, nullptr, (int)MV_CNT_N_MINSDF, 0, n, training ? blocks_for(n) + (tail ? 1 : 0) : 0, tail ? 1 : 0};
        hipLaunchKernelGGL((k_ray_samples<MT, NTW, NW, NET>), dim3(sec_blocks + minsdf.blocks), dim3(64 * NW), lds2, stream, net, tp, cm, minsdf, none,
                           sec_blocks);
        if (training) hipLaunchKernelGGL(k_reduce_items, dim3(red_blocks), dim3(64 * MV_RED_WAVES), 0, stream, tp, cm, w_list_min, (const int*)nullptr, (int)MV_CNT_N_MINSDF, 2);
    }
    if ((parts & 4) && training) {
        SampleCtx c2 = c;
        c2.sv = (float*)(c.src_rest + R);                        // second sample-value buffer [R * n_steps]
        const bool tail = mv_tail_on<NET>(training, steps, R, mt1);
        const RowSeg minsdf = {w_list_min, nullptr, (int)MV_CNT_N_MINSDF, 0, n, blocks_for(n) + (tail ? 1 : 0), tail ? 1 : 0};
        hipLaunchKernelGGL((k_ray_samples<MT, NTW, NW, NET>), dim3(minsdf.blocks), dim3(64 * NW), lds2, stream, net, tp, c2, minsdf, none, 0);
        hipLaunchKernelGGL(k_reduce_items, dim3(red_blocks), dim3(64 * MV_RED_WAVES), 0, stream, tp, c2, w_list_min, (const int*)nullptr, (int)MV_CNT_N_MINSDF, 2);
    }
    if (parts & 8)
        hipLaunchKernelGGL((k_ray_samples<MT, NTW, NW, NET>), dim3(sec_blocks), dim3(64 * NW), lds2, stream, net, tp, c, none, none, sec_blocks);
    return hipGetLastError();
}

// mt1: row tiles per workgroup of the sphere-tracing kernel (8*mt1 rays); mt2: row tiles per chunk of the sample-row kernels.
#ifndef MV_SPHERE_NW16
#define MV_SPHERE_NW16 1                                            // (-DMV_SPHERE_NW16=0: the 8-wave x 2-tile form, for A/B builds -- tools/pp_ab.sh)
#endif
template <class NET>
hipError_t mv_trace_launch(int stages, const NET& net, const MvTraceParams& tp, int mt1, int mt2, const float* cam_loc, const float* dirs,
                           const uint8_t* om, int B, int P, int training, const float* intervals, const float* steps, float* points,
                           uint8_t* mask, float* dists, float* ws, unsigned long long* counters, hipStream_t stream) {
    int maxnt = 0;
    for (int l = 0; l < net.n_layers - 1; ++l) maxnt = net.L[l].NT > maxnt ? net.L[l].NT : maxnt;
    if (maxnt > 32) return hipErrorInvalidValue;
    // waves per workgroup: 8 (two per SIMD) once there are >= 2 column tiles per wave to share (width 256 up; forcing 4 there measured 0.78 -> 0.99 ms, round 3)
    constexpr bool is_bf = !std::is_same<NET, MvNet>::value;                    // the bf16-MFMA engines are built for 8-wave workgroups only
    const bool eight = is_bf || maxnt >= 16;
    const bool wide = maxnt > 16;
    hipError_t e = hipSuccess;
    auto eff = [&](int mt) { return wide ? (mt >= 2 ? 2 : 1) : (mt >= 4 ? 4 : (mt >= 2 ? 2 : 1)); };   // the instantiation a requested tile count maps to
    const int mt1_eff = eff(mt1), mt2_eff = eff(mt2);
#define MV_S1(MT_, NTW_, NW_) e = launch_stage1<MT_, NTW_, NW_>(net, tp, cam_loc, dirs, om, B, P, training, steps, mt2_eff, points, mask, dists, ws, counters, stream)
#define MV_S2(MT_, NTW_, NW_) e = launch_stage2<MT_, NTW_, NW_>(net, tp, cam_loc, dirs, B, P, training, intervals, steps, points, mask, dists, ws, counters, (stages >> 1) & 15, mt1_eff, stream)
    if (stages & 1) {
        if (eight) {
            if (wide) { if (mt1 >= 2) MV_S1(2, 4, 8); else MV_S1(1, 4, 8); }
            else if (mt1 >= 4) MV_S1(4, 2, 8); else if (mt1 >= 2) MV_S1(2, 2, 8);
            else {
                // one 16-row tile per workgroup, three weight terms: SIXTEEN waves x one column tile (late round 6).  The evaluations of this kernel wait for each
                // other, a wave's share of a layer is a latency chain (k-blocks of 3 x 2 dependent instructions, then the softplus epilogue): half the chain per
                // wave.  tools/micro/x3_engine_rounds.hip: 38.4 (8 waves x 2 tiles) -> 33.7 us per evaluation with the ping-pong tiles.  Same instruction sequence
                // per output column: same bits.
                if constexpr (MV_SPHERE_NW16 != 0 && mv_net_wt<NET>::v == 3) MV_S1(1, 1, 16); else MV_S1(1, 2, 8);
            }
        } else if constexpr (!is_bf) {
            if (wide) { if (mt1 >= 2) MV_S1(2, 8, 4); else MV_S1(1, 8, 4); }
            else if (mt1 >= 4) MV_S1(4, 4, 4); else if (mt1 >= 2) MV_S1(2, 4, 4); else MV_S1(1, 4, 4);
        }
        if (e != hipSuccess) return e;
    }
    // the two sampler launches are small (about one wave of workgroups at a few thousand rays): 16-row chunks spread them over more
    // CUs (measured 285 -> 241 us at 2048 rays).  MVSDF_MT_FIRST overrides (dev).
    static int mtf_env = -1;
    if (mtf_env < 0) { const char* e2 = mv_dev_env("MVSDF_MT_FIRST"); mtf_env = e2 ? atoi(e2) : 0; }
    for (int part = 1; part <= 8; part <<= 1) {
        if (!((stages >> 1) & part)) continue;
        // (three weight terms: a 16-row evaluation is bound by its 3.1 MB weight stream, two tiles share it: c2 1.687 -> 1.664 ms, c5 share 2.539 -> 2.483)
        const int mtp = part == 1 ? (mtf_env > 0 ? mtf_env : (mv_net_wt<NET>::v == 3 ? (mt2 > 2 ? 2 : mt2) : ((long long)B * P <= 4096 ? 1 : mt2))) : mt2;
        const int st_ = stages;
        stages = (stages & 1) | (part << 1);                    // MV_S2 reads `stages` for the parts to launch
        if (eight) {
            if (wide) { if (mtp >= 2) MV_S2(2, 4, 8); else MV_S2(1, 4, 8); }
            else if (mtp >= 4) MV_S2(4, 2, 8); else if (mtp >= 2) MV_S2(2, 2, 8); else MV_S2(1, 2, 8);
        } else if constexpr (!is_bf) {
            if (wide) { if (mtp >= 2) MV_S2(2, 8, 4); else MV_S2(1, 8, 4); }
            else if (mtp >= 4) MV_S2(4, 4, 4); else if (mtp >= 2) MV_S2(2, 4, 4); else MV_S2(1, 4, 4);
        }
        stages = st_;
        if (e != hipSuccess) return e;
    }
#undef MV_S1
#undef MV_S2
    return e;
}

// =============================================================================================================
// Generic tracer for an OPAQUE `sdf` callable (ray_tracing.py:27-32; the reference calls it as a lambda, idr.py:194).  The per-ray state
// machine is the one of k_sphere_trace, split at its evaluation points: `k_gen_init` / `k_gen_step` EMIT the points each ray wants
// evaluated (dense [R][2][3] + request flags), the host runs the Python callable on the requested rows, `k_gen_step` CONSUMES the values
// and emits the next requests.  `k_gen_finish` is the tail of k_sphere_trace (masks, work lists); sampler / min-sdf rows are emitted by
// `k_gen_rows`, reduced by k_reduce_items, the secant by `k_gen_secant`.  Slow (one host round trip per evaluation) but the decisions stay in
// HIP, bit-identical to the fused path -- this is what pins the state machine to the reference's analytic-SDF goldens on the GPU.
struct GenRay {                      // 16 dwords per ray
    float acc_s, acc_e, next_s, next_e, curr_s, curr_e, t0, t1, ts, te;
    int iters, k, phase, flags, pad0, pad1;      // flags: 1 unf_s, 2 unf_e, 4 req_s, 8 req_e, 16 isect, 32 om
};

__device__ __forceinline__ void mv_gen_emit(const GenRay& g, const float* c, const float* d, uint8_t* req, float* pts) {
    const bool rs = g.flags & 4, re = g.flags & 8;
    req[0] = rs ? 1 : 0; req[1] = re ? 1 : 0;
    pts[0] = c[0] + g.ts * d[0]; pts[1] = c[1] + g.ts * d[1]; pts[2] = c[2] + g.ts * d[2];
    pts[3] = c[0] + g.te * d[0]; pts[4] = c[1] + g.te * d[1]; pts[5] = c[2] + g.te * d[2];
}

__global__ void k_gen_init(MvTraceParams tp, const float* __restrict__ cam_loc, const float* __restrict__ dirs, const uint8_t* __restrict__ om,
                           int R, int P, GenRay* __restrict__ state, uint8_t* __restrict__ req, float* __restrict__ pts) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= R) return;
    float c[3], d[3];
    for (int i = 0; i < 3; ++i) { c[i] = cam_loc[3 * (gid / P) + i]; d[i] = dirs[3 * (size_t)gid + i]; }
    GenRay g;
    const bool isect = mv_sphere_isect(c, d, tp.r, g.t0, g.t1);
    g.acc_s = isect ? g.t0 : 0.f; g.acc_e = isect ? g.t1 : 0.f;
    g.next_s = g.next_e = g.curr_s = g.curr_e = 0.f;
    g.ts = g.acc_s; g.te = g.acc_e;
    g.iters = 0; g.k = 0; g.phase = isect ? 0 : 3;
    g.flags = (isect ? (1 | 2 | 4 | 8 | 16) : 0) | (om[gid] ? 32 : 0);
    g.pad0 = g.pad1 = 0;
    state[gid] = g;
    mv_gen_emit(g, c, d, req + 2 * (size_t)gid, pts + 6 * (size_t)gid);
}

// consume vals[R][2] (the callable's outputs scattered back to the dense layout), advance every ray by one round, emit the next requests
__global__ void k_gen_step(MvTraceParams tp, const float* __restrict__ cam_loc, const float* __restrict__ dirs, int R, int P,
                           GenRay* __restrict__ state, const float* __restrict__ vals, uint8_t* __restrict__ req, float* __restrict__ pts,
                           unsigned long long* __restrict__ counters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int nreq = 0;
    if (gid < R) {
        GenRay g = state[gid];
        if (g.phase != 3) {
            bool unf_s = g.flags & 1, unf_e = g.flags & 2;
            const bool req_s = g.flags & 4, req_e = g.flags & 8;
            nreq = (req_s ? 1 : 0) + (req_e ? 1 : 0);
            const float vs = req_s ? mv_clamp(vals[2 * (size_t)gid], -tp.dist_clip, tp.dist_clip) : 0.f;
            const float ve = req_e ? mv_clamp(vals[2 * (size_t)gid + 1], -tp.dist_clip, tp.dist_clip) : 0.f;
            bool nreq_s = false, nreq_e = false, end_iter = false;
            if (g.phase == 0) { g.next_s = vs; g.next_e = ve; }
            else if (g.phase == 1) { g.next_s = vs; g.next_e = ve; g.k = 0; }
            else { if (req_s) g.next_s = vs; if (req_e) g.next_e = ve; g.k++; }
            if (g.phase != 0) {
                const bool np_s = g.next_s < 0.f, np_e = g.next_e < 0.f;
                if (g.k < tp.line_step_iters && (np_s || np_e)) {                     // ray_tracing.py:173-191
                    const float coef = (1.0f - tp.line_search_step) / (float)(1 << g.k);
                    nreq_s = np_s; nreq_e = np_e;
                    if (np_s) { g.acc_s -= coef * g.curr_s; g.ts = g.acc_s; }
                    if (np_e) { g.acc_e += coef * g.curr_e; g.te = g.acc_e; }
                    g.phase = 2;
                } else {
                    end_iter = true;
                    unf_s = unf_s && (g.acc_s < g.acc_e);                             // ray_tracing.py:193-194
                    unf_e = unf_e && (g.acc_s < g.acc_e);
                }
            }
            if (g.phase == 0 || end_iter) {                                           // top of the while loop, ray_tracing.py:139-171
                g.curr_s = unf_s ? g.next_s : 0.f;
                g.curr_e = unf_e ? g.next_e : 0.f;
                if (g.curr_s <= tp.thr) g.curr_s = 0.f;
                if (g.curr_e <= tp.thr) g.curr_e = 0.f;
                unf_s = unf_s && (g.curr_s > tp.thr);
                unf_e = unf_e && (g.curr_e > tp.thr);
                if ((!unf_s && !unf_e) || g.iters == tp.st_iters) {
                    g.phase = 3; nreq_s = false; nreq_e = false;
                } else {
                    g.iters++;
                    g.acc_s = g.acc_s + g.curr_s;
                    g.acc_e = g.acc_e - g.curr_e;
                    nreq_s = unf_s; nreq_e = unf_e; g.ts = g.acc_s; g.te = g.acc_e;
                    g.phase = 1;
                }
            }
            g.flags = (g.flags & (16 | 32)) | (unf_s ? 1 : 0) | (unf_e ? 2 : 0) | (nreq_s ? 4 : 0) | (nreq_e ? 8 : 0);
            state[gid] = g;
            float c[3], d[3];
            for (int i = 0; i < 3; ++i) { c[i] = cam_loc[3 * (gid / P) + i]; d[i] = dirs[3 * (size_t)gid + i]; }
            mv_gen_emit(g, c, d, req + 2 * (size_t)gid, pts + 6 * (size_t)gid);
        } else {
            req[2 * (size_t)gid] = 0; req[2 * (size_t)gid + 1] = 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) nreq += __shfl_xor(nreq, o);
    if ((threadIdx.x & 63) == 0 && nreq) atomicAdd(&counters[MV_CNT_ROWS_SPHERE], (unsigned long long)nreq);
}

// tail of k_sphere_trace: ray_tracing.py:41-44, 73-94
__global__ void k_gen_finish(MvTraceParams tp, const float* __restrict__ cam_loc, const float* __restrict__ dirs, int R, int P, int training,
                             const GenRay* __restrict__ state, float* __restrict__ o_points, uint8_t* __restrict__ o_mask,
                             float* __restrict__ o_dists, float* __restrict__ w_zmin, float* __restrict__ w_zmax, int* __restrict__ w_list,
                             int* __restrict__ w_list_min, unsigned long long* __restrict__ counters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= R) return;
    const GenRay g = state[gid];
    float c[3], d[3];
    for (int i = 0; i < 3; ++i) { c[i] = cam_loc[3 * (gid / P) + i]; d[i] = dirs[3 * (size_t)gid + i]; }
    const bool om = g.flags & 32, isect = g.flags & 16;
    const bool net_mask = g.acc_s < g.acc_e;
    const bool sampler = g.flags & 1;
    float dist = g.acc_s, zmin = g.acc_s, zmax = g.acc_e;
    bool listed = false;
    int kind = 0;
    if (sampler) { listed = true; kind = MV_ITEM_SAMPLER | (om ? MV_ITEM_OM : 0); }
    else if (training) {
        const bool in_mask = !net_mask && om, out_mask = !om;
        if (in_mask || out_mask) {
            if (!isect) {
                const float dot = (d[0] * c[0] + d[1] * c[1]) + d[2] * c[2];
                dist = -dot;
            } else {
                listed = true; kind = MV_ITEM_MINSDF;
                zmin = (net_mask && out_mask) ? g.acc_s : g.t0;
                zmax = g.t1;
            }
        }
    }
    o_mask[gid] = net_mask ? 1 : 0;
    o_dists[gid] = dist;
    o_points[3 * (size_t)gid + 0] = c[0] + dist * d[0];
    o_points[3 * (size_t)gid + 1] = c[1] + dist * d[1];
    o_points[3 * (size_t)gid + 2] = c[2] + dist * d[2];
    w_zmin[gid] = zmin; w_zmax[gid] = zmax;
    (void)listed; (void)kind; (void)w_list; (void)w_list_min; (void)counters;
    // the work lists are filled in RAY ORDER by k_gen_lists (one thread): the opaque callable may depend on the order of its rows
    // (the reference evaluates them in ray order, ray_tracing.py:215-219), so no atomics here
    w_list[gid] = listed ? (gid | (kind << 28)) : -1;                              // scratch: per-ray entry, compacted below
}

// stable compaction of the per-ray entries into the sampler / min-sdf lists (single workgroup, ballot scan over chunks of 1024 rays)
__global__ __launch_bounds__(1024) void k_gen_lists(int R, int* __restrict__ w_list, int* __restrict__ w_list_min, int* __restrict__ scratch,
                                                    unsigned long long* __restrict__ counters) {
    __shared__ int wsum[2][16];
    __shared__ int base[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += 1024) {
        const int gid = r0 + tid;
        const int e = gid < R ? scratch[gid] : -1;
        const bool smp = e >= 0 && ((e >> 28) & MV_ITEM_SAMPLER), mn = e >= 0 && ((e >> 28) & MV_ITEM_MINSDF);
        const unsigned long long bs = __ballot(smp), bm = __ballot(mn), lt = (1ull << lane) - 1ull;
        if (lane == 0) { wsum[0][w] = __popcll(bs); wsum[1][w] = __popcll(bm); }
        __syncthreads();
        int os = base[0], omn = base[1];
        for (int j = 0; j < w; ++j) { os += wsum[0][j]; omn += wsum[1][j]; }
        if (smp) w_list[os + __popcll(bs & lt)] = e;
        if (mn) w_list_min[omn + __popcll(bm & lt)] = e;
        __syncthreads();
        if (tid == 0) { for (int j = 0; j < 16; ++j) { base[0] += wsum[0][j]; base[1] += wsum[1][j]; } }
        __syncthreads();
    }
    if (tid == 0) { counters[MV_CNT_N_SAMPLER] = (unsigned long long)base[0]; counters[MV_CNT_N_MINSDF] = (unsigned long long)base[1]; }
}

// sample points of a work list: row q = item * n_steps + i  (ray_tracing.py:206-213 sampler, 287-297 min-sdf)
__global__ void k_gen_rows(int n_list, int n_steps, const int* __restrict__ list, const float* __restrict__ cam_loc, const float* __restrict__ dirs,
                           int P, const float* __restrict__ zs /* intervals or steps */, int is_sampler, const float* __restrict__ w_zmin,
                           const float* __restrict__ w_zmax, float* __restrict__ out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long long)n_list * n_steps) return;
    const int it = (int)(q / n_steps), i = (int)(q - (long long)it * n_steps);
    const int gid = list[it] & 0x0fffffff;
    const float* cc = cam_loc + 3 * (gid / P);
    const float* d = dirs + 3 * (size_t)gid;
    const float zmin = w_zmin[gid], zmax = w_zmax[gid];
    const float z = is_sampler ? (zmin + zs[i] * (zmax - zmin)) : (zs[i] * (zmax - zmin) + zmin);
    out[3 * q + 0] = cc[0] + z * d[0]; out[3 * q + 1] = cc[1] + z * d[1]; out[3 * q + 2] = cc[2] + z * d[2];
}

// secant (ray_tracing.py:260-278), one thread per listed ray.  op 0: emit z_pred points; op 1: consume sdf_mid, update the bracket;
// op 2: write the final z_pred to dists / points.  z_pred is a pure function of the bracket, recomputed where needed.
__global__ void k_gen_secant(int op, int n_sec, int R, int P, const int* __restrict__ sec_list, float* __restrict__ sec_state,
                             const float* __restrict__ cam_loc, const float* __restrict__ dirs, const float* __restrict__ vals,
                             float* __restrict__ pts_out, float* __restrict__ o_points, float* __restrict__ o_dists) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sec) return;
    const int gid = sec_list[i];
    float z_low = sec_state[gid], z_high = sec_state[(size_t)R + gid], sdf_low = sec_state[2 * (size_t)R + gid], sdf_high = sec_state[3 * (size_t)R + gid];
    float z_pred = -sdf_low * (z_high - z_low) / (sdf_high - sdf_low) + z_low;
    const float* cc = cam_loc + 3 * (gid / P);
    const float* d = dirs + 3 * (size_t)gid;
    if (op == 1) {
        const float sm = vals[i];
        if (sm > 0.f) { sec_state[gid] = z_pred; sec_state[2 * (size_t)R + gid] = sm; }
        if (sm < 0.f) { sec_state[(size_t)R + gid] = z_pred; sec_state[3 * (size_t)R + gid] = sm; }
        return;
    }
    float* o = op == 0 ? pts_out + 3 * (size_t)i : o_points + 3 * (size_t)gid;
    o[0] = cc[0] + z_pred * d[0]; o[1] = cc[1] + z_pred * d[1]; o[2] = cc[2] + z_pred * d[2];
    if (op == 2) o_dists[gid] = z_pred;
}

// sort the secant list by ray id (k_reduce_items appends with atomics; the callable sees its rows in ray order like the reference's masks)
__global__ __launch_bounds__(1024) void k_gen_sort_list(int* __restrict__ list, const unsigned long long* __restrict__ counters, int cnt_index,
                                                        uint8_t* __restrict__ marks, int R) {
    // marks[R] must be zero on entry; single workgroup: mark, then stable compaction
    __shared__ int wsum[16];
    __shared__ int base;
    const int n = (int)counters[cnt_index], tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < n; i += 1024) marks[list[i]] = 1;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += 1024) {
        const int gid = r0 + tid;
        const bool m = gid < R && marks[gid];
        const unsigned long long b = __ballot(m), lt = (1ull << lane) - 1ull;
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int o = base;
        for (int j = 0; j < w; ++j) o += wsum[j];
        if (m) list[o + __popcll(b & lt)] = gid;
        __syncthreads();
        if (tid == 0) for (int j = 0; j < 16; ++j) base += wsum[j];
        __syncthreads();
    }
}

struct GenWs { float* w_zmin; float* w_zmax; float* sec_state; int* w_list; int* w_list_min; int* sec_list; float* sv; int* list_rest; int* src_rest; };
static GenWs mv_gen_ws(void* ws, int R, int n) {
    GenWs g;
    g.w_zmin = (float*)ws; g.w_zmax = g.w_zmin + R; g.sec_state = g.w_zmax + R;
    g.w_list = (int*)(g.sec_state + 4 * (size_t)R); g.w_list_min = g.w_list + R; g.sec_list = g.w_list_min + R;
    g.sv = (float*)(g.sec_list + R); g.list_rest = (int*)(g.sv + (size_t)R * n); g.src_rest = g.list_rest + R;
    return g;
}

// =============================================================================================================
extern "C" {

size_t mvsdf_trace_workspace_bytes_n(int R, int n_steps) { return (size_t)(R > 0 ? R : 0) * (44 + 8 * (size_t)(n_steps > 0 ? n_steps : 0)) + 256; }
size_t mvsdf_trace_workspace_bytes(int R) { return mvsdf_trace_workspace_bytes_n(R, 128); }

static int trace_impl(int stages, const MvsdfNetDesc* desc, const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs,
                      const uint8_t* object_mask, int B, int P, int training, const float* intervals, const float* minsdf_steps,
                      float* points, uint8_t* mask, float* dists, unsigned long long* counters, void* workspace,
                      size_t workspace_bytes, int mt, int rpw, void* stream) {
    MvNet net;
    MvNetBs<2> net2;
    MvNetBs<3> net3;
    MvNetBs<3, 3> net33;
    const int td = desc ? desc->trace_dtype : 0;
    if (td == 1) return mv_fail(-2, "mvsdf_trace: trace_dtype 1 (bf16 weights AND 8-bit activations) was removed in round 5: use 3 (bf16x2: same speed, parity-checked)");
    int rc = (td == 3 ? mv_make_net_bs(desc, &net2, 2) : (td == 4 ? mv_make_net_bs(desc, &net3, 3) :
             (td == 5 ? mv_make_net_bs(desc, &net33, 3) : mv_make_net_trace(desc, &net))));
    if (rc) return rc;
    if (!tp || !cam_loc || !ray_dirs || !object_mask || !intervals || !points || !mask || !dists || !counters || !workspace)
        return mv_fail(-1, "mvsdf_trace: null argument");
    if (B <= 0 || P <= 0 || (long long)B * P >= (1 << 28)) return mv_fail(-1, "mvsdf_trace: B*P out of range");
    if (training && !minsdf_steps) return mv_fail(-1, "mvsdf_trace: training needs minsdf_steps");
    if (tp->n_steps < 2 || tp->n_steps > 1024 || tp->line_step_iters < 0 || tp->line_step_iters > 30)
        return mv_fail(-1, "mvsdf_trace: tracer parameters out of range");
    const int R = B * P;
    if (workspace_bytes < mvsdf_trace_workspace_bytes_n(R, tp->n_steps)) return mv_fail(-1, "mvsdf_trace: workspace too small");
    if (rpw < 1) rpw = 1;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if ((stages & 1) && !(stages & 0x100)) e = hipMemsetAsync(counters, 0, 16 * sizeof(unsigned long long), s);   // 0x100: the caller's previous launch zeroed them
    if (e != hipSuccess) return mv_check(e, "mvsdf_trace: memset");
    if (td == 3)
        e = mv_trace_launch(stages, net2, *tp, mt, rpw, cam_loc, ray_dirs, object_mask, B, P, training, intervals,
                            minsdf_steps ? minsdf_steps : intervals, points, mask, dists, (float*)workspace, counters, s);
    else if (td == 4)
        e = mv_trace_launch(stages, net3, *tp, mt, rpw, cam_loc, ray_dirs, object_mask, B, P, training, intervals,
                            minsdf_steps ? minsdf_steps : intervals, points, mask, dists, (float*)workspace, counters, s);
    else if (td == 5)
        e = mv_trace_launch(stages, net33, *tp, mt, rpw, cam_loc, ray_dirs, object_mask, B, P, training, intervals,
                            minsdf_steps ? minsdf_steps : intervals, points, mask, dists, (float*)workspace, counters, s);
    else
        e = mv_trace_launch(stages, net, *tp, mt, rpw, cam_loc, ray_dirs, object_mask, B, P, training, intervals,
                            minsdf_steps ? minsdf_steps : intervals, points, mask, dists, (float*)workspace, counters, s);
    return mv_check(e, "mvsdf_trace");
}

int mvsdf_trace(const MvsdfNetDesc* desc, const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs,
                const uint8_t* object_mask, int B, int P, int training, const float* intervals, const float* minsdf_steps,
                float* points, uint8_t* mask, float* dists, unsigned long long* counters, void* workspace,
                size_t workspace_bytes, int mt, int rpw, void* stream) {
    return trace_impl(7, desc, tp, cam_loc, ray_dirs, object_mask, B, P, training, intervals, minsdf_steps, points, mask, dists, counters,
                      workspace, workspace_bytes, mt, rpw, stream);
}

/* The launches of mvsdf_trace separately (same arguments, same workspace): stage 1 = sphere tracing (zeroes the counters),
 * stage 2 = ray sampler + secant + min-sdf; or stage 3 = ray sampler rows only (`mask` is final after it) followed by
 * stage 4 = secant + min-sdf (only points / dists still change).  Lets a caller bracket each kernel with events and read the
 * hit count while the last stage still runs. */
int mvsdf_trace_stage(int stage, const MvsdfNetDesc* desc, const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs,
                      const uint8_t* object_mask, int B, int P, int training, const float* intervals, const float* minsdf_steps,
                      float* points, uint8_t* mask, float* dists, unsigned long long* counters, void* workspace,
                      size_t workspace_bytes, int mt, int rpw, void* stream) {
    if (stage < 1 || stage > 6) return mv_fail(-1, "mvsdf_trace_stage: stage must be 1..6");
    static const int bits[7] = {0, 1, 6, 2, 4, 8, 16};
    return trace_impl(bits[stage], desc, tp, cam_loc, ray_dirs, object_mask, B, P, training, intervals, minsdf_steps, points, mask, dists, counters,
                      workspace, workspace_bytes, mt, rpw, stream);
}

// the step driver's form of stage 1: the counters were zeroed by k_step_prologue (one fill node less per step)
int mv_trace_stage1_prezeroed(const MvsdfNetDesc* desc, const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, const uint8_t* object_mask,
                              int B, int P, int training, const float* intervals, const float* minsdf_steps, float* points, uint8_t* mask, float* dists,
                              unsigned long long* counters, void* workspace, size_t workspace_bytes, int mt, int rpw, void* stream) {
    return trace_impl(1 | 0x100, desc, tp, cam_loc, ray_dirs, object_mask, B, P, training, intervals, minsdf_steps, points, mask, dists, counters,
                      workspace, workspace_bytes, mt, rpw, stream);
}

/* ---- generic tracer for an opaque SDF callable (see the kernel comments above) ---- */
size_t mvsdf_tracegen_state_bytes(int R) { return (size_t)(R > 0 ? R : 0) * sizeof(GenRay); }

static int tracegen_check(const MvsdfTraceParams* tp, int B, int P) {
    if (!tp || B <= 0 || P <= 0 || (long long)B * P >= (1 << 28)) return mv_fail(-1, "mvsdf_tracegen: bad sizes");
    if (tp->n_steps < 2 || tp->n_steps > 1024 || tp->line_step_iters < 0 || tp->line_step_iters > 30) return mv_fail(-1, "mvsdf_tracegen: tracer parameters out of range");
    return 0;
}

int mvsdf_tracegen_init(const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, const uint8_t* object_mask, int B, int P,
                        void* state, uint8_t* req, float* pts, unsigned long long* counters, void* stream) {
    int rc = tracegen_check(tp, B, P);
    if (rc) return rc;
    if (!cam_loc || !ray_dirs || !object_mask || !state || !req || !pts || !counters) return mv_fail(-1, "mvsdf_tracegen_init: null argument");
    const int R = B * P;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counters, 0, 16 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return mv_check(e, "mvsdf_tracegen_init: memset");
    hipLaunchKernelGGL(k_gen_init, dim3((R + 255) / 256), dim3(256), 0, s, *tp, cam_loc, ray_dirs, object_mask, R, P, (GenRay*)state, req, pts);
    return mv_check(hipGetLastError(), "mvsdf_tracegen_init");
}

int mvsdf_tracegen_step(const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, int B, int P, void* state, const float* vals,
                        uint8_t* req, float* pts, unsigned long long* counters, void* stream) {
    int rc = tracegen_check(tp, B, P);
    if (rc) return rc;
    if (!cam_loc || !ray_dirs || !state || !vals || !req || !pts || !counters) return mv_fail(-1, "mvsdf_tracegen_step: null argument");
    const int R = B * P;
    hipLaunchKernelGGL(k_gen_step, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, *tp, cam_loc, ray_dirs, R, P, (GenRay*)state, vals, req, pts, counters);
    return mv_check(hipGetLastError(), "mvsdf_tracegen_step");
}

int mvsdf_tracegen_finish(const MvsdfTraceParams* tp, const float* cam_loc, const float* ray_dirs, int B, int P, int training, const void* state,
                          float* points, uint8_t* mask, float* dists, unsigned long long* counters, void* workspace, size_t workspace_bytes,
                          void* stream) {
    int rc = tracegen_check(tp, B, P);
    if (rc) return rc;
    const int R = B * P;
    if (!cam_loc || !ray_dirs || !state || !points || !mask || !dists || !counters || !workspace) return mv_fail(-1, "mvsdf_tracegen_finish: null argument");
    if (workspace_bytes < mvsdf_trace_workspace_bytes_n(R, tp->n_steps)) return mv_fail(-1, "mvsdf_tracegen_finish: workspace too small");
    GenWs w = mv_gen_ws(workspace, R, tp->n_steps);
    hipStream_t s = (hipStream_t)stream;
    int* scratch = w.sec_list;                                  // per-ray entries before the stable compaction (sec_list is filled later)
    hipLaunchKernelGGL(k_gen_finish, dim3((R + 255) / 256), dim3(256), 0, s, *tp, cam_loc, ray_dirs, R, P, training, (const GenRay*)state, points, mask,
                       dists, w.w_zmin, w.w_zmax, scratch, w.w_list_min, counters);
    hipLaunchKernelGGL(k_gen_lists, dim3(1), dim3(1024), 0, s, R, w.w_list, w.w_list_min, scratch, counters);
    return mv_check(hipGetLastError(), "mvsdf_tracegen_finish");
}

/* kind 0: ray-sampler rows (intervals), 1: min-sdf rows (steps) of the n_list listed rays -> out_pts[n_list * n_steps][3] */
int mvsdf_tracegen_rows(const MvsdfTraceParams* tp, int kind, const float* cam_loc, const float* ray_dirs, int B, int P, const float* zs, int n_list,
                        void* workspace, float* out_pts, void* stream) {
    int rc = tracegen_check(tp, B, P);
    if (rc) return rc;
    if (!cam_loc || !ray_dirs || !zs || !workspace || !out_pts || n_list <= 0 || n_list > B * P || kind < 0 || kind > 1)
        return mv_fail(-1, "mvsdf_tracegen_rows: bad arguments");
    GenWs w = mv_gen_ws(workspace, B * P, tp->n_steps);
    const long long total = (long long)n_list * tp->n_steps;
    hipLaunchKernelGGL(k_gen_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_list, tp->n_steps,
                       kind == 0 ? w.w_list : w.w_list_min, cam_loc, ray_dirs, P, zs, kind == 0 ? 1 : 0, w.w_zmin, w.w_zmax, out_pts);
    return mv_check(hipGetLastError(), "mvsdf_tracegen_rows");
}

/* per-ray reduction of the callable's values sv[n_list][n_steps] (kind 0: first sign change / P_out argmin / secant hand-off,
 * ray_tracing.py:221-256; kind 1: min-sdf argmin, 303-307).  After kind 0 the secant list is sorted by ray and `mask` is final. */
int mvsdf_tracegen_reduce(const MvsdfTraceParams* tp, int kind, const float* cam_loc, const float* ray_dirs, int B, int P, int training,
                          const float* intervals, const float* minsdf_steps, const float* sv, float* points, uint8_t* mask, float* dists,
                          unsigned long long* counters, void* workspace, uint8_t* marks, void* stream) {
    int rc = tracegen_check(tp, B, P);
    if (rc) return rc;
    const int R = B * P;
    if (!cam_loc || !ray_dirs || !intervals || !sv || !points || !mask || !dists || !counters || !workspace || kind < 0 || kind > 1 || (kind == 0 && !marks))
        return mv_fail(-1, "mvsdf_tracegen_reduce: bad arguments");
    GenWs w = mv_gen_ws(workspace, R, tp->n_steps);
    SampleCtx c;
    c.cam_loc = cam_loc; c.dirs = ray_dirs; c.R = R; c.P = P; c.training = training; c.RPW = 0; c.intervals = intervals;
    c.steps = minsdf_steps ? minsdf_steps : intervals;
    c.o_points = points; c.o_mask = mask; c.o_dists = dists; c.w_zmin = w.w_zmin; c.w_zmax = w.w_zmax; c.sec_state = w.sec_state;
    c.sec_list = w.sec_list; c.sv = (float*)sv; c.counters = counters; c.list_rest = w.list_rest; c.src_rest = w.src_rest; c.n_first = tp->n_steps;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) {
        hipLaunchKernelGGL(k_reduce_items, dim3((R + MV_RED_WAVES - 1) / MV_RED_WAVES), dim3(64 * MV_RED_WAVES), 0, s, *tp, c, w.w_list, (const int*)nullptr, (int)MV_CNT_N_SAMPLER, 0);
        hipError_t e = hipMemsetAsync(marks, 0, (size_t)R, s);
        if (e != hipSuccess) return mv_check(e, "mvsdf_tracegen_reduce: memset");
        hipLaunchKernelGGL(k_gen_sort_list, dim3(1), dim3(1024), 0, s, w.sec_list, counters, (int)MV_CNT_N_SECANT, marks, R);
    } else {
        hipLaunchKernelGGL(k_reduce_items, dim3((R + MV_RED_WAVES - 1) / MV_RED_WAVES), dim3(64 * MV_RED_WAVES), 0, s, *tp, c, w.w_list_min, (const int*)nullptr, (int)MV_CNT_N_MINSDF, 2);
    }
    return mv_check(hipGetLastError(), "mvsdf_tracegen_reduce");
}

/* op 0: emit the n_sec secant points -> pts_out[n_sec][3]; op 1: consume vals[n_sec]; op 2: write the final dists / points */
int mvsdf_tracegen_secant(const MvsdfTraceParams* tp, int op, const float* cam_loc, const float* ray_dirs, int B, int P, int n_sec, const float* vals,
                          float* pts_out, float* points, float* dists, unsigned long long* counters, void* workspace, void* stream) {
    int rc = tracegen_check(tp, B, P);
    if (rc) return rc;
    const int R = B * P;
    if (!cam_loc || !ray_dirs || !workspace || n_sec <= 0 || n_sec > R || op < 0 || op > 2 || (op == 0 && !pts_out) || (op == 1 && !vals) ||
        (op == 2 && (!points || !dists)))
        return mv_fail(-1, "mvsdf_tracegen_secant: bad arguments");
    GenWs w = mv_gen_ws(workspace, R, tp->n_steps);
    hipLaunchKernelGGL(k_gen_secant, dim3((n_sec + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, n_sec, R, P, w.sec_list, w.sec_state, cam_loc,
                       ray_dirs, vals, pts_out, points, dists);
    (void)counters;
    return mv_check(hipGetLastError(), "mvsdf_tracegen_secant");
}

}  // extern "C"
