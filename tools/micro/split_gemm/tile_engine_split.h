// tile_engine_split.h -- the GEMMs of the fused DIFFERENTIABLE chain kernels (layer_kernels.h: value / normal forward, first- and second-order backward,
// rendering net) on gfx950's bf16 matrix cores with fp32 accuracy: both operands as three bf16 terms, six products.
//
// Those kernels are chains of dependent layer phases on ONE 16-row tile per CU (3 k rows at the bench shape): a phase's GEMM is 3.4 us of pure
// v_mfma_f32_16x16x4_f32 issue (4.4 us measured, tools/chain_probe.py) out of ~7 us -- the fp32 matrix rate itself is the bound, 17 times per kernel.
// v_mfma_f32_16x16x32_bf16 multiplies 8x the k per instruction in half the cycles.  An fp32 value is EXACTLY t0 + t1 + t2 with t0 = bf16(a),
// t1 = bf16(a - t0), t2 = bf16(a - t0 - t1) (tile_engine_bf16s.h), a bf16 x bf16 product is exact in fp32, so
//     a w = sum over i + j <= 2 of a_i w_j  +  terms below 2^-24 |a w|          (a0 w0, a0 w1, a1 w0, a0 w2, a1 w1, a2 w0)
// is a float-accurate product from six matrix instructions of 16 cycles instead of eight of 32 per 32-wide k-block: 2.7x the matrix rate.  The sums are
// formed in the matrix core's own order, not as the k-ascending fmaf chain of the fp32 engine: the differentiable half is tolerance-checked against the
// reference's autograd (1e-4 / 5e-4), never bit for bit -- the no-grad TRACING engine, whose decisions are bit-checked, keeps the fp32 instruction.
//   * weights: split once per step into three bf16 packs (mvsdf_pack_split_net / the step driver), layout of tile_engine_bf16.h per term:
//         ws[((ct * KB32 + kb) * 3 + term) * 64 + lane][j] = term(W[16 ct + (lane & 15)][32 kb + 8 (lane >> 4) + j])
//   * activations: the chain kernels keep their fp32 LDS tile (4x4-transposed 16-blocks; prologues and epilogues are untouched); before a GEMM the
//     workgroup splits it ONCE into three bf16 tiles [rows][S] in natural k order (mv_split_act: one task per (row, 16-block)) -- every wave needs the
//     whole tile, splitting per wave would repeat the VALU work 8-16 times.
#pragma once
#include "tile_engine.h"
#include "tile_engine_bf16s.h"

// rows x Kp fp32 (stride S floats, 4x4-transposed 16-blocks) -> three bf16 tiles of rows x S elements each (natural k order)
template <int ROWS, int NTH>
__device__ __forceinline__ void mv_split_act(const float* __restrict__ act, int S, int Kp, uint16_t* __restrict__ terms, int tid) {
    const int nb = Kp >> 4;
    for (int task = tid; task < ROWS * nb; task += NTH) {
        const int row = task / nb, b = task - row * nb;
        const float* src = act + row * S + 16 * b;
        const f32x4 v0 = *(const f32x4*)(src), v1 = *(const f32x4*)(src + 4), v2 = *(const f32x4*)(src + 8), v3 = *(const f32x4*)(src + 12);
        // position 4 i + j holds k = 4 j + i (mv_perm): k = 0..15 in order
        const float k_[16] = {v0[0], v1[0], v2[0], v3[0], v0[1], v1[1], v2[1], v3[1], v0[2], v1[2], v2[2], v3[2], v0[3], v1[3], v2[3], v3[3]};
        uint32_t p[8][3];
#pragma unroll
        for (int e = 0; e < 8; ++e) mv_split_pk<3>(dm_f2{k_[2 * e], k_[2 * e + 1]}, p[e]);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            uint16_t* dst = terms + (size_t)s * ROWS * S + row * S + 16 * b;
            *(uint4*)(dst) = uint4{p[0][s], p[1][s], p[2][s], p[3][s]};
            *(uint4*)(dst + 8) = uint4{p[4][s], p[5][s], p[6][s], p[7][s]};
        }
    }
}

// acc[rt][t] += act[rt*16.., :K] * W[(ct0+t)*16.., :K]^T   for t < ntw (<= NTW), from the term tiles and the split pack; k-block kb + 1's operands are
// requested before the 6 * MTc * ntw matrix instructions of k-block kb issue
template <int MTc, int NTW>
__device__ __forceinline__ void mv_gemm_split(const MvLayer& L, const uint16_t* __restrict__ terms, int rows, int S, int ct0, int ntw, f32x4 (&acc)[MTc][NTW], int lane) {
    const int KB32 = L.KB >> 1;
    const uint4* __restrict__ wp = L.ws + (size_t)ct0 * KB32 * 3 * 64 + lane;
    const uint16_t* arow = terms + (lane & 15) * S + 8 * (lane >> 4);
    const int TT = rows * S;
    uint4 a[2][MTc][3], b[2][NTW][3];
    auto load = [&](int kb, int buf) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tc = t < ntw ? t : ntw - 1;                                           // clamped: no branch around a load
#pragma unroll
            for (int s = 0; s < 3; ++s) b[buf][t][s] = wp[(((size_t)tc * KB32 + kb) * 3 + s) * 64];
        }
#pragma unroll
        for (int r = 0; r < MTc; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) a[buf][r][s] = *(const uint4*)(arow + s * TT + r * 16 * S + kb * 32);
    };
    load(0, 0);
    for (int kb0 = 0; kb0 < KB32; kb0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = kb0 + u;
            if (kb < KB32) {
                load(kb + 1 < KB32 ? kb + 1 : kb, u ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < MTc; ++r)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        if (t < ntw) {
                            // smallest products first
#define MV_SP(i, j) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, a[u][r][i]), __builtin_bit_cast(mv_bf8, b[u][t][j]), acc[r][t], 0, 0, 0)
                            MV_SP(2, 0); MV_SP(0, 2); MV_SP(1, 1); MV_SP(1, 0); MV_SP(0, 1); MV_SP(0, 0);
#undef MV_SP
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// The GEMM of one chain phase.  ALL threads of the workgroup call it (the split pass and its barrier are collective); `terms` == nullptr or a layer
// without a split pack: the fp32 engine (mv_gemm_dispatch).  The caller's barriers around the GEMM stay as they are: the one before it completes `act`,
// the one after it keeps the next phase's split pass away from waves that still read the term tiles.
// do_split = false: a further group of column tiles of the SAME layer (the tile has not changed since the split of the first group).
template <int MTc, int NTW, int NTH>
__device__ __forceinline__ void mv_gemm_chain_g(const MvLayer& L, const float* __restrict__ act, int S, int ct0, int ntw, f32x4 (&acc)[MTc][NTW], int lane, int tid,
                                                uint16_t* terms, bool do_split) {
    if (terms != nullptr && L.ws != nullptr) {
        if (do_split) {
            mv_split_act<16 * MTc, NTH>(act, S, L.KB * 16, terms, tid);
            mv_barrier_lds();
        }
        if (ntw > 0) mv_gemm_split<MTc, NTW>(L, terms, 16 * MTc, S, ct0, ntw, acc, lane);
    } else if (ntw > 0) {
        mv_gemm_dispatch<MTc, NTW>(L, act, S, ct0, ntw, acc, lane);
    }
}
template <int MTc, int NTW, int NTH>
__device__ __forceinline__ void mv_gemm_chain(const MvLayer& L, const float* __restrict__ act, int S, int ct0, int ntw, f32x4 (&acc)[MTc][NTW], int lane, int tid,
                                              uint16_t* terms) {
    mv_gemm_chain_g<MTc, NTW, NTH>(L, act, S, ct0, ntw, acc, lane, tid, terms, true);
}
