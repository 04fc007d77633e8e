// tile_engine_rows.h -- "row-owner" evaluation of the tracing MLP for the throughput regime (many sample rows).
//
// tile_engine.h splits a layer's COLUMNS over the waves of a workgroup: every wave needs every activation, so each layer is
// barrier -> GEMM -> barrier -> softplus and the matrix pipe idles during every epilogue; weights stream from L2 once per 32-64 rows.
// Here a wave OWNS 16 rows for the whole network (workgroup = 4 waves = 64 rows):
//   * its activations never leave the wave: the 16x16 output tiles are transposed into the next layer's A fragments through a 1.3 KB
//     wave-private LDS tile -- no workgroup barrier between layers, the waves drift apart and one wave's softplus runs under the
//     other waves' MFMAs;
//   * what the waves share is the WEIGHT stream: each 16-wide k-block of a layer ([NT tiles][64 lanes][float4], 16 KB at width 256) is
//     staged ONCE per workgroup in a 3-stage LDS ring (every wave copies a quarter of it) and read by all four waves -- "weight tiles
//     staged in LDS and reused across a wavefront's rays".  Producer / consumer hand-over through per-stage LDS counters, no barriers;
//     the ring runs through the whole network (the next layer's first k-blocks arrive during the epilogue).
// Same arithmetic in the same order as mv_sdf_eval_col0 (v_mfma_f32_16x16x4_f32 in ascending k, det_math epilogue): bit-identical.
#pragma once
#include "tile_engine.h"

#define MV_ROWS_RING 3
#define MV_ROWS_TS 20            // row stride (floats) of the wave-private transpose tile: 16 columns + 4 pad (b128 reads stay 16-B aligned)

struct MvRowsLds {
    float4* ring;                // [MV_ROWS_RING][NTL][64]
    float* tile;                 // [4 waves][2][16][MV_ROWS_TS]  (two transpose tiles per wave, alternating)
    int* full;                   // [MV_ROWS_RING] waves that have written their part of the stage (monotonic)
    int* empty;                  // [MV_ROWS_RING] waves that have finished reading the stage (monotonic)
};
template <int NTL>
__host__ __device__ static inline size_t mv_rows_lds_floats() { return (size_t)MV_ROWS_RING * NTL * 64 * 4 + 4 * 2 * 16 * MV_ROWS_TS + 16; }
template <int NTL>
__device__ __forceinline__ MvRowsLds mv_rows_carve(float* base) {
    MvRowsLds l;
    l.ring = (float4*)base;
    l.tile = base + (size_t)MV_ROWS_RING * NTL * 64 * 4;
    l.full = (int*)(l.tile + 4 * 2 * 16 * MV_ROWS_TS);
    l.empty = l.full + MV_ROWS_RING;
    return l;
}

// position of column c (0..15) inside a transposed tile row: the four k-steps of one lane-quarter (k = 4s + q) are contiguous
__device__ __forceinline__ int mv_rows_pos(int c) { return ((c & 3) << 2) | (c >> 2); }

// ImplicitNetwork.forward(...)[:, 0] for the 64 rows of a 256-thread workgroup: wave w owns rows [16w, 16w + 16).  pts: LDS [64][3];
// out: LDS [64].  The caller zeroes lds.full / lds.empty (6 ints) AND the ring (unused tile slots must be finite) and issues a workgroup
// barrier before the call; ends with a barrier.
template <int NTL>
__device__ __forceinline__ void mv_sdf_eval_col0_rows(const MvNet& net, const MvRowsLds& lds, const float* pts, float* out, int tid) {
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    const int nl = net.n_layers, d0 = 3 + 6 * net.multires;
    const unsigned skm = net.skip_mask;
    float* tile = lds.tile + w * 2 * 16 * MV_ROWS_TS;

    // ---- the weight stream: global k-block index G runs over (layer, kb); loader cursor (ll, lkb), data of ONE block in flight in registers
    int ll = 0, lkb = 0, G_load = 0;                             // next block to load from global memory
    float4 ld[NTL / 4];
    auto load_block = [&]() {                                    // issue the global loads of block (ll, lkb) -- this wave's tiles w, w+4, ...
        if (ll < nl) {
            const MvLayer& L = net.L[ll];
            const int NT = (ll == nl - 1) ? 1 : L.NT;
            const float4* wp = L.wp + (size_t)lkb * 64 + lane;
#pragma unroll
            for (int j = 0; j < NTL / 4; ++j) {
                const int ct = w + 4 * j;
                if (ct < NT) ld[j] = wp[(size_t)ct * L.KB * 64];
            }
        }
    };
    auto store_block = [&]() {                                   // write the block in `ld` (index G_load) into its ring stage and publish it
        if (ll < nl) {
            const int st = G_load % MV_ROWS_RING, gen = G_load / MV_ROWS_RING;
            const int NT = (ll == nl - 1) ? 1 : net.L[ll].NT;
            mv_ks_wait(lds.empty + st, 4 * gen);                 // everyone has finished reading the stage's previous tenant
            float4* dst = lds.ring + (size_t)st * NTL * 64 + lane;
#pragma unroll
            for (int j = 0; j < NTL / 4; ++j) {
                const int ct = w + 4 * j;
                if (ct < NT) dst[ct * 64] = ld[j];
            }
            mv_ks_signal(lds.full + st, lane);
            ++G_load;
            if (++lkb == net.L[ll].KB) { lkb = 0; ++ll; }
        }
    };

    // ---- layer-0 input: positional encoding of this wave's rows straight into A-fragment layout (pe_a[kb][s] = pe[row r][16 kb + 4 s + q])
    float4 pe_a[4];
    {
        const float* x = pts + (16 * w + r) * 3;
        const float x0 = x[0], x1 = x[1], x2 = x[2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int p = 16 * kb + 4 * s + q;
                float val = 0.0f;
                if (p < 3) val = p == 0 ? x0 : (p == 1 ? x1 : x2);
                else if (p < d0) {
                    const int jj = p - 3, m = jj / 6, rem = jj - 6 * m, c = rem % 3;
                    float sn, co;
                    dm_sincos((c == 0 ? x0 : (c == 1 ? x1 : x2)) * (float)(1 << m), &sn, &co);
                    val = rem < 3 ? sn : co;
                }
                v[s] = val;
            }
            pe_a[kb] = float4{v[0], v[1], v[2], v[3]};
        }
    }

    load_block();                                                // block 0
    store_block();
    load_block();                                                // block 1 stays in registers until the first compute step

    // A fragment of k-block t of layer l (l >= 1) = softplus of the PREVIOUS layer's tile t, transposed through the wave-private LDS tile.
    // Produced just before the k-block that consumes it (k ascends, so tile t is needed exactly at step t): the epilogue is spread over
    // the next layer's k-loop, one tile per k-block, and the other wave of the SIMD issues its MFMAs meanwhile.
    f32x4 prev[NTL];                                             // z (pre-bias) of the previous layer, tile t = columns [16t, 16t + 16)
    int G = 0;                                                   // block being consumed
    for (int l = 0; l < nl; ++l) {
        const MvLayer& L = net.L[l];
        const bool last = (l == nl - 1);
        const int KB = L.KB;
        const MvLayer& Lp = net.L[l > 0 ? l - 1 : 0];            // producer of this layer's input
        const int Np = Lp.N, NTp = Lp.NT;
        const bool from_skip = mv_skip_at(skm, l);                        // this layer's input is cat([h, PE]) / sqrt(2)
        f32x4 acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The GEMM computes ALL NTL column tiles (no per-tile branches between the MFMAs): slots of the ring that a layer does not fill hold
        // zeros or an earlier layer's (finite) weights, their accumulators are never read.  The last layer needs column 0 only: one tile.
#define MV_ROWS_KLOOP(NTC_)                                                                                                              \
        _Pragma("unroll") for (int kb = 0; kb < NTL; ++kb) {                                                                             \
            if (kb < KB) {                                                                                                               \
                float4 av;                                                                                                               \
                if (l == 0) av = pe_a[kb < 4 ? kb : 3];                                                                                  \
                else {                                                                                                                   \
                    const int col = 16 * kb + r;                                                                                         \
                    float h[4] = {0.f, 0.f, 0.f, 0.f};                                                                                   \
                    if (kb < NTp && col < Np) {                                                                                          \
                        const float bvt = Lp.bias[col];                                                                                  \
                        const dm_f2 h01 = mv_act2(dm_f2{prev[kb][0] + bvt, prev[kb][1] + bvt});                                          \
                        const dm_f2 h23 = mv_act2(dm_f2{prev[kb][2] + bvt, prev[kb][3] + bvt});                                          \
                        h[0] = h01.x; h[1] = h01.y; h[2] = h23.x; h[3] = h23.y;                                                          \
                        if (from_skip) { h[0] *= 0.7071067690849304f; h[1] *= 0.7071067690849304f; h[2] *= 0.7071067690849304f; h[3] *= 0.7071067690849304f; } \
                    }                                                                                                                    \
                    float* tl = tile + (kb & 1) * 16 * MV_ROWS_TS;                                                                       \
                    const int pos = mv_rows_pos(r);                                                                                      \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) tl[(4 * q + i) * MV_ROWS_TS + pos] = h[i];                             \
                    if (from_skip && 16 * kb + 16 > Np) {        /* columns [Np, Np + d0) of the skip input: PE / sqrt(2), idr.py:86-87 */ \
                        _Pragma("unroll") for (int sl = 0; sl < 12; ++sl) {                                                              \
                            const int p = 16 * (sl >> 2) + 4 * (sl & 3) + q, c = Np + p;                                                 \
                            if (p < d0 && (c >> 4) == kb) tl[r * MV_ROWS_TS + mv_rows_pos(c & 15)] = dm_div_sqrt2(((const float*)&pe_a[sl >> 2])[sl & 3]); \
                        }                                                                                                                \
                    }                                                                                                                    \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
                    av = *(const float4*)(tl + r * MV_ROWS_TS + 4 * q);                                                                  \
                }                                                                                                                        \
                store_block();                                   /* block G + 1 -> its stage (waits for the readers of block G - 2) */  \
                load_block();                                    /* block G + 2 -> registers */                                         \
                const int st = G % MV_ROWS_RING, gen = G / MV_ROWS_RING;                                                                 \
                mv_ks_wait(lds.full + st, 4 * (gen + 1));                                                                                \
                const float4* bs = lds.ring + (size_t)st * NTL * 64 + lane;                                                              \
                _Pragma("unroll") for (int g4 = 0; g4 < ((NTC_) + 3) / 4; ++g4) {                                                        \
                    constexpr int NJ = (NTC_) >= 4 ? 4 : (NTC_);                                                                         \
                    float4 b[NJ];                                                                                                        \
                    _Pragma("unroll") for (int j = 0; j < NJ; ++j) b[j] = bs[(4 * g4 + j) * 64];                                         \
                    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                        \
                        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                   \
                            acc[4 * g4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&av)[s], ((const float*)&b[j])[s], acc[4 * g4 + j], 0, 0, 0); \
                }                                                                                                                        \
                mv_ks_signal(lds.empty + st, lane);              /* (waits for this wave's LDS reads first) */                          \
                ++G;                                                                                                                     \
            }                                                                                                                            \
        }
        if (last) { MV_ROWS_KLOOP(1) } else { MV_ROWS_KLOOP(NTL) }
#undef MV_ROWS_KLOOP
        if (last) {
            if (r == 0) {
                const float b_last = L.bias[0];
#pragma unroll
                for (int i = 0; i < 4; ++i) out[16 * w + 4 * q + i] = acc[0][i] + b_last;
            }
        } else {
#pragma unroll
            for (int t = 0; t < NTL; ++t) prev[t] = acc[t];
        }
    }
    __syncthreads();
}
