// wgrad_x3.h -- the weight gradients of every layer (k_wgrad_net's work, layer_kernels.h) on gfx950's bf16 matrix cores in the three-term fp32 arithmetic of
// chain_x3.h / tile_engine_bf16s.h: dW[o][i] = sum over rows of P[row][o] Q[row][i] with every fp32 element of P and Q split into three bf16 terms (exact),
// the six products p_s q_j, s + j <= 2, of v_mfma_f32_16x16x32_bf16, fp32 accumulation.
//
// The contraction runs over ROWS, so both matrix operands are needed "k-major" (a lane holds eight consecutive rows of one column) while the tensors are
// row-major.  gfx950's LDS transpose read does that for free: a 64-row x 64-column stage of an operand is written to LDS as three planes (one per term) of
// 2 x 4 subtiles [32 rows][16 columns] of bf16 (a thread's float4 -- four columns of a row -- becomes one 8-byte write per plane), and
// ds_read_b64_tr_b16 with lane-linear addresses (lane l reads the 8 bytes at 8 l of a 512-byte half subtile) hands lane (q, r) the four rows 4 q .. 4 q + 3 of
// column r.  Two such reads (rows 0-15 and 16-31 of the subtile) are a lane's eight k-values of one matrix instruction; P and Q use the same row permutation
// (k = 16 h + 4 q + j), so every row meets itself.  Same workgroup decomposition, slabs, bias sums, column-sum blocks and reduction (k_reduce_net) as k_wgrad_net.
#pragma once
#include "layer_kernels.h"
#include "tile_engine_bf16s.h"

#define MV_WX3_SUB 528                     // bf16 elements per [32][16] subtile + 16 of padding: the four column tiles of a row land in different banks when staged
#define MV_WX3_PLANE (8 * MV_WX3_SUB)      // one term of one operand: 2 row halves x 4 column tiles
typedef short mv_v4s __attribute__((ext_vector_type(4)));

// eight k-values (rows 16 h + 4 q + j, h = 0, 1) of column r of subtile `sub` of plane `pl`: the operand fragment of one matrix instruction
__device__ __forceinline__ uint4 mv_wx3_frag(const uint16_t* pl, int sub, int lane) {
    const uint16_t* p = pl + sub * MV_WX3_SUB + 4 * lane;
    const mv_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mv_v4s*)p);
    const mv_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mv_v4s*)(p + 256));
    uint4 v;
    v.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16); v.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    v.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16); v.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return v;
}

__global__ __launch_bounds__(MV_THREADS) void k_wgrad_net_x3(WgradNetArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t Pl[3 * MV_WX3_PLANE];
    __shared__ __attribute__((aligned(16))) uint16_t Ql[3 * MV_WX3_PLANE];
    int bid = blockIdx.x;
    if (a.xcd_runs) {                                            // (k_wgrad_net's XCD-aware block order)
        const int x = bid & 7, i = bid >> 3;
        bid = (((i >> 4) << 3) + x) * 16 + (i & 15);
        if (bid >= a.nblocks) return;
    }
    if (a.colX && bid >= a.col_blk0) { mv_colsum_block(a, bid - a.col_blk0, (float*)Pl); return; }
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
    int l = 0;
    while (l + 1 < a.n_layers && bid >= a.L[l + 1].blk0) ++l;
    const WgradLayer& L = a.L[l];
    const int local = bid - L.blk0, nb = L.nbx * L.nby;
    const int chl = local / nb, rem = local - chl * nb, by = rem / L.nbx, bx = rem - by * L.nbx;
    const int ch = L.ch0 + chl;
    const int i0 = bx * 64, o0 = by * 64, No = L.No, Ki = L.Ki;
    const int rbeg = ch * a.chunk, rend = min(L.M, rbeg + a.chunk);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const bool do_bias = bx == 0;
    const int npairs = L.P2 ? 2 : 1;
    const int nrb = (rend - rbeg + 63) / 64, nst = npairs * nrb;
    // full 64-column tiles of 16-byte-aligned rows: unconditional 16-byte loads (rows clamped, masked when staged); else guarded element loads
    const bool fast = ((L.ldp1 & 3) == 0) && ((L.ldq1 & 3) == 0) && ((((size_t)L.P1) & 15) == 0) && ((((size_t)L.Q1) & 15) == 0) && o0 + 64 <= No && i0 + 64 <= Ki &&
                      (!L.P2 || (((L.ldp2 & 3) == 0) && ((L.ldq2 & 3) == 0) && ((((size_t)L.P2) & 15) == 0) && ((((size_t)L.Q2) & 15) == 0)));
    float4 pv[4], qv[4];
    auto issue = [&](int st) {
        const int pair = st >= nrb ? 1 : 0, rb = rbeg + (st - pair * nrb) * 64;
        const float* P = pair ? L.P2 : L.P1;
        const float* Q = pair ? L.Q2 : L.Q1;
        const int ldp = pair ? L.ldp2 : L.ldp1, ldq = pair ? L.ldq2 : L.ldq1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = u * MV_THREADS + tid, rr = idx >> 4, c4 = (idx & 15) * 4;
            const int row = min(rb + rr, rend - 1);
            if (fast) {
                pv[u] = *(const float4*)(P + (size_t)row * ldp + o0 + c4);
                qv[u] = *(const float4*)(Q + (size_t)row * ldq + i0 + c4);
            } else {
                float tp[4], tq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    tp[e] = o0 + c4 + e < No ? P[(size_t)row * ldp + o0 + c4 + e] : 0.0f;
                    tq[e] = i0 + c4 + e < Ki ? Q[(size_t)row * ldq + i0 + c4 + e] : 0.0f;
                }
                pv[u] = make_float4(tp[0], tp[1], tp[2], tp[3]);
                qv[u] = make_float4(tq[0], tq[1], tq[2], tq[3]);
            }
        }
    };
    auto store = [&](int st) {                                   // split into the three term planes, subtile layout
        const int pair = st >= nrb ? 1 : 0, rb = rbeg + (st - pair * nrb) * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = u * MV_THREADS + tid, rr = idx >> 4, c4 = (idx & 15) * 4;
            const bool in = rb + rr < rend;
            const float4 p4 = in ? pv[u] : make_float4(0.f, 0.f, 0.f, 0.f), q4 = in ? qv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int off = ((rr >> 5) * 4 + (c4 >> 4)) * MV_WX3_SUB + (rr & 31) * 16 + (c4 & 15);
            uint32_t a0[3], a1[3], b0[3], b1[3];
            mv_split_pk<3>(dm_f2{p4.x, p4.y}, a0); mv_split_pk<3>(dm_f2{p4.z, p4.w}, a1);
            mv_split_pk<3>(dm_f2{q4.x, q4.y}, b0); mv_split_pk<3>(dm_f2{q4.z, q4.w}, b1);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                *(uint2*)(Pl + s * MV_WX3_PLANE + off) = uint2{a0[s], a1[s]};
                *(uint2*)(Ql + s * MV_WX3_PLANE + off) = uint2{b0[s], b1[s]};
            }
        }
    };
    auto compute = [&](int st) {
        if (do_bias && st < nrb) {                               // column sums of P (the bias gradient): column tid & 63, the 16 rows of quarter w
            const int col = tid & 63;
            const uint16_t* pc = Pl + ((w >> 1) * 4 + (col >> 4)) * MV_WX3_SUB + (w & 1) * 256 + (col & 15);
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr)
                bsum += (mv_bf2f(pc[rr * 16]) + mv_bf2f(pc[MV_WX3_PLANE + rr * 16])) + mv_bf2f(pc[2 * MV_WX3_PLANE + rr * 16]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {                            // two k-blocks of 32 rows
            uint4 af[3], bf[4][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) af[s] = mv_wx3_frag(Pl + s * MV_WX3_PLANE, h * 4 + w, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 3; ++s) bf[t][s] = mv_wx3_frag(Ql + s * MV_WX3_PLANE, h * 4 + t, lane);
#pragma unroll
            for (int o = 2; o >= 0; --o)                          // smallest products first
#pragma unroll
                for (int s = 0; s <= o; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, af[s]), __builtin_bit_cast(mv_bf8, bf[t][o - s]), acc[t], 0, 0, 0);
        }
    };
    if (nst > 0) {
        issue(0);
        for (int st = 0; st + 1 < nst; ++st) {
            __syncthreads();
            store(st);
            __syncthreads();
            issue(st + 1);
            compute(st);
        }
        __syncthreads();
        store(nst - 1);
        __syncthreads();
        compute(nst - 1);
    }
    float* slab = L.slab + (size_t)ch * No * Ki;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = i0 + 16 * t + r;
        if (i < Ki) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = o0 + 16 * w + 4 * q + e;
                if (o < No) slab[(size_t)o * Ki + i] = acc[t][e];
            }
        }
    }
    if (do_bias) {                                               // the four row quarters of every column, in quarter order
        __syncthreads();
        float* red = (float*)Ql;
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64 && o0 + tid < No) L.bslab[(size_t)ch * No + o0 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
    }
}
