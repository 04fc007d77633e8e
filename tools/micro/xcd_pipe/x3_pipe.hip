// Prototype (dev, gfx950): the f32x3 tracing MLP as a WEIGHT-STATIONARY pipeline per XCD instead of a weight stream per evaluation.
//
// Today (tile_engine_bf16s.h inside k_sphere_trace): every CU owns 16 rows and streams the whole three-term weight set (3.1 MB) through its L2 port for each of the
// ~22 dependent evaluations of a training step: 37 us per evaluation, of which the port is busy 45 % (profiles/r05_x3_engine_phases.txt).
// Here: the 32 CUs of an XCD form one pipeline.  One CU runs positional encoding + layer 0, four CUs per hidden layer hold that layer's weights IN REGISTERS (one
// 16-column tile per wave: 96 VGPRs of B fragments, two waves per tile taking alternate row tiles), one CU the last layer.  A 16-row tile travels from layer to layer
// through the XCD's L2 as MFMA-ready A fragments (3 bf16 terms x k-blocks x 1 KiB); a producer wave stores its 16 x 16 outputs, waits for the stores, then stores a
// round number into the tile's flag line; consumers poll the flags with cache-bypassing loads.  No LDS, no workgroup barrier, no fence: workgroup -> XCD is blockIdx % 8
// and same-XCD stores are visible through the shared L2 (profiles/r05_xcd_hop_probe.txt).  Every output element sees exactly the instruction sequence of the engine
// (same k order, same term order, same epilogue), so the results must be BIT-IDENTICAL to mv_sdf_eval_col0 -- checked below against that engine on the same inputs.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I../../../mvsdf_amd/csrc x3_pipe.hip -o x3_pipe && ./x3_pipe [rounds] [tiles per XCD]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "tile_engine_bf16s.h"

typedef int pv4i __attribute__((ext_vector_type(4)));
#define PIPE_KBMAX 8
#define PIPE_SPIN_MAX 3000000

struct PipeBufs {
    unsigned long long* ph;  // [8] phase clock ticks of one workgroup's wave 0 (dev)
    uint4* abuf;            // [xcd][T][2][3][PIPE_KBMAX][64]   A fragments of the tile's current layer boundary (ping-pong by boundary parity)
    float* pe_side;         // [xcd][T][40][16]                 PE / sqrt(2) of the tile's points (the skip layer's extra input columns)
    float* pts;             // [xcd][T][16][4]                  the tile's points (owned by one wave of the PE stage)
    float* out;             // [xcd][T][16]
    unsigned* flags;        // [xcd][T][10][16]                 round number per (boundary, producing column tile); boundary 9 = out
    unsigned* abort_flag;
    int T, rounds;
};

__device__ __forceinline__ size_t pb_frag(const PipeBufs& p, int x, int t, int buf, int s, int kb) { return (((((size_t)x * p.T + t) * 2 + buf) * 3 + s) * PIPE_KBMAX + kb) * 64; }
__device__ __forceinline__ unsigned* pb_flag(const PipeBufs& p, int x, int t, int b) { return p.flags + (((size_t)x * p.T + t) * 10 + b) * 16; }

__device__ __forceinline__ unsigned ld_flag(const unsigned* f) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(f) : "memory");
    return v;
}
// wait until the first n entries of a flag line are >= r (every lane polls one entry).  false: gave up (abort flag set)
__device__ __forceinline__ bool wait_flags(const PipeBufs& p, const unsigned* line, int n, unsigned r, int lane) {
    int spin = 0;
    for (;;) {
        const unsigned v = lane < n ? ld_flag(line + lane) : r;
        if (__all((int)(v >= r))) return true;
        if (++spin > PIPE_SPIN_MAX || ((spin & 255) == 0 && ld_flag(p.abort_flag))) {
            if (lane == 0) atomicAdd(p.abort_flag, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
__device__ __forceinline__ void set_flag(unsigned* f, unsigned r) {
    asm volatile("s_waitcnt vmcnt(0)\n\tglobal_store_dword %0, %1, off" :: "v"(f), "v"(r) : "memory");
}
__device__ __forceinline__ uint4 ld_frag(__amdgpu_buffer_rsrc_t rs, size_t elem, int lane) {
    const pv4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((elem + lane) * 16), 0, 0x11);      // sc0 sc1: from the XCD's L2, never a stale L1 line
    return uint4{(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
}

// the six matrix instructions of one k-block on one column tile, in the engine's order (mv_gemm_rolling_bw: a_s w_j with s + j = 2, then 1, then 0)
__device__ __forceinline__ f32x4 kblock_mfma(f32x4 acc, const uint4 (&b)[3], const uint4 (&a)[3]) {
#pragma unroll
    for (int o = 2; o >= 0; --o)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int j = o - s;
            if (j >= 0 && j < 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[j]), __builtin_bit_cast(mv_bf8, a[s]), acc, 0, 0, 0);
        }
    return acc;
}

// epilogue values of one accumulator -> the three term fragments of the next boundary (cols 16 ct + 4 q .. + 3 of row r)
__device__ __forceinline__ void store_cols(const PipeBufs& p, int x, int t, int buf, int ct, int r, int q, dm_f2 h0, dm_f2 h1) {
    uint32_t p0[3], p1[3];
    mv_split_pk<3>(h0, p0);
    mv_split_pk<3>(h1, p1);
    const int kb = ct >> 1, lane2 = (2 * (ct & 1) + (q >> 1)) * 16 + r;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        uint16_t* dst = (uint16_t*)(p.abuf + pb_frag(p, x, t, buf, s, kb) + lane2) + 4 * (q & 1);
        *(uint2*)dst = uint2{p0[s], p1[s]};
    }
}

__device__ __forceinline__ float pe_value(const float* xyz, int c) {                 // embedder.py:10-36 column c of [x, sin(2^0 x), cos(2^0 x), ...] (dm_sincos like the engine)
    if (c < 3) return xyz[c];
    const int m = (c - 3) / 6, rem = (c - 3) % 6;
    float s, co;
    dm_sincos(xyz[rem % 3] * (float)(1 << m), &s, &co);
    return rem < 3 ? s : co;
}

// ---- PE stage: the tile's points of this round (stands in for the ray state machine), positional encoding -> boundary 0 fragments + pe_side.  Wave w: tiles t = w (mod 4) ----
__device__ void pe_stage(const MvNetBs<3, 3>& net, const PipeBufs& p, int x, int w, int nw, int lane) {
    const int r = lane & 15, q = lane >> 4, d0 = 3 + 6 * net.multires;
    for (int rd = 1; rd <= p.rounds; ++rd) {
        for (int t = w; t < p.T; t += nw) {
            float o = 0.0f;
            if (rd > 1) {
                if (!wait_flags(p, pb_flag(p, x, t, 9), 1, (unsigned)(rd - 1), lane)) return;
                o = __uint_as_float(ld_flag((const unsigned*)(p.out + ((size_t)x * p.T + t) * 16 + r)));      // written by the last layer's CU: from L2
            }
            float* pp = p.pts + (((size_t)x * p.T + t) * 16 + r) * 4;                   // (only this wave touches the tile's points)
            float xyz[3] = {pp[0], pp[1], pp[2]};
            if (rd > 1) xyz[0] = xyz[0] + 1e-3f * o;                                  // the next round depends on this one
            if (q == 0) pp[0] = xyz[0];
            float* ps = p.pe_side + (((size_t)x * p.T + t) * 40) * 16;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                uint32_t pk[4][3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = 32 * kb + 8 * q + 2 * i;
                    const float v0 = c0 < d0 ? pe_value(xyz, c0) : 0.0f, v1 = c0 + 1 < d0 ? pe_value(xyz, c0 + 1) : 0.0f;
                    if (c0 < d0) ps[c0 * 16 + r] = dm_div_sqrt2(v0);
                    if (c0 + 1 < d0) ps[(c0 + 1) * 16 + r] = dm_div_sqrt2(v1);
                    mv_split_pk<3>(dm_f2{mv_x3_flush(v0), mv_x3_flush(v1)}, pk[i]);
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) p.abuf[pb_frag(p, x, t, 0, s, kb) + lane] = uint4{pk[0][s], pk[1][s], pk[2][s], pk[3][s]};
            }
            set_flag(pb_flag(p, x, t, 0), (unsigned)rd);
        }
    }
}

// ---- a layer's workgroup: 4 waves = 4 column tiles (layer 0: 4 x 4); ITEMS of TPI row tiles (TPI independent accumulators per wave: the matrix instructions of
// one tile are a dependent chain), items of one parity.  The item's A fragments (TPI x 24 KiB) are fetched ONCE per workgroup into LDS (requested while the
// previous item is multiplied), then every wave multiplies its column tile(s) from LDS. ----
#ifndef PIPE_TPI
#define PIPE_TPI 2
#endif
template <int NTW, int KBM>
__device__ void layer_wg(const MvNetBs<3, 3>& net, const PipeBufs& p, int x, int l, int c, int par, int w, int lane, uint4* lds) {
    constexpr int TPI = PIPE_TPI, NPRE = 6 * TPI;
    const int r = lane & 15, q = lane >> 4, tid = w * 64 + lane;
    const MvLayerBf& L = net.L[l];
    const int nl = net.n_layers, NT = L.NT, N = L.N, KB = L.KB, d0 = 3 + 6 * net.multires;
    const bool last = l == nl - 1;
    const bool to_skip = !last && mv_skip_at(net.skip_mask, l + 1);
    const int ntot = last ? 1 : (to_skip ? (N + d0 + 15) / 16 : NT);                 // column tiles of the next boundary this layer produces
    const int ct0 = (4 * c + w) * NTW;
    const int nprod = l == 0 ? 1 : (L.K + 15) / 16;
    uint4 b[KBM][NTW][3];
    f32x4 bias4[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
        const int ct = ct0 + tt, cc = ct < NT ? ct : NT - 1;
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int j = 0; j < 3; ++j) b[kb][tt][j] = L.wp[(((size_t)cc * KB + (kb < KB ? kb : KB - 1)) * 3 + j) * 64 + lane];
        bias4[tt] = *(const f32x4*)(L.bias + cc * 16 + 4 * q);
    }
    const float sc = to_skip ? 0.7071067690849304f : 1.0f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.abuf, 0, 0x7fffffff, 0x00020000);
    const int nitem_all = (p.T + TPI - 1) / TPI, nt = (nitem_all - par + 1) / 2, items = p.rounds * nt;
    uint4 pre[NPRE];
    auto tile_of = [&](int it, int u) { return (par + 2 * (it % nt)) * TPI + u; };
    auto issue = [&](int it) {                                                       // the item's fragments -> registers (6 per thread and tile)
#pragma unroll
        for (int u = 0; u < TPI; ++u) {
            const int t = tile_of(it, u) < p.T ? tile_of(it, u) : p.T - 1;
            const size_t base = pb_frag(p, x, t, l & 1, 0, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) pre[6 * u + i] = ld_frag(rs, base + 256 * i, tid);
        }
    };
    auto ready = [&](int it, bool block) -> int {                                    // 1 ready, 0 not yet, -1 abort
        const int rd = 1 + it / nt;
        for (int u = 0; u < TPI; ++u) {
            const int t = tile_of(it, u);
            if (t >= p.T) continue;
            if (block) { if (!wait_flags(p, pb_flag(p, x, t, l), nprod, (unsigned)rd, lane)) return -1; }
            else {
                const unsigned v = lane < nprod ? ld_flag(pb_flag(p, x, t, l) + lane) : (unsigned)rd;
                if (!__all((int)(v >= (unsigned)rd))) return 0;
            }
        }
        return 1;
    };
    if (items <= 0) return;
    if (ready(0, true) < 0) return;
    issue(0);
    const bool probe = p.ph && x == 0 && l == 2 && c == 0 && par == 0 && tid == 0;
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = wall_clock64();
#define PH(i) { const unsigned long long t_ = wall_clock64(); ph_[i] += t_ - tp_; tp_ = t_; }
    for (int it = 0; it < items; ++it) {
        const int rd = 1 + it / nt;
        __syncthreads();                                                             // everyone is done reading the previous item's fragments
#pragma unroll
        for (int i = 0; i < NPRE; ++i) lds[(i / 6) * (3 * PIPE_KBMAX * 64) + tid + 256 * (i % 6)] = pre[i];
        PH(0)
        __syncthreads();                                                             // the item's fragments are in LDS
        PH(1)
        bool fetched = false;
        if (it + 1 < items && ready(it + 1, false) == 1) { issue(it + 1); fetched = true; }
        PH(2)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
            const int ct = ct0 + tt;
            if (ct >= ntot) continue;
            f32x4 acc[TPI];
#pragma unroll
            for (int u = 0; u < TPI; ++u) acc[u] = bias4[tt];
            if (ct < NT) {
#pragma unroll
                for (int kb = 0; kb < KBM; ++kb) {
                    if (kb < KB) {
                        uint4 a[TPI][3];
#pragma unroll
                        for (int u = 0; u < TPI; ++u)
#pragma unroll
                            for (int s = 0; s < 3; ++s) a[u][s] = lds[u * (3 * PIPE_KBMAX * 64) + (s * PIPE_KBMAX + kb) * 64 + lane];
                        // the tiles' chains interleaved instruction by instruction (each accumulator still sees its six products in the engine's order)
#pragma unroll
                        for (int o = 2; o >= 0; --o)
#pragma unroll
                            for (int s = 0; s < 3; ++s) {
                                const int j = o - s;
                                if (j >= 0 && j < 3) {
#pragma unroll
                                    for (int u = 0; u < TPI; ++u)
                                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mv_bf8, b[kb][tt][j]), __builtin_bit_cast(mv_bf8, a[u][s]), acc[u], 0, 0, 0);
                                }
                            }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < TPI; ++u) {
                const int t = tile_of(it, u);
                if (t >= p.T) continue;
                if (last) {
                    if (q == 0) p.out[((size_t)x * p.T + t) * 16 + r] = acc[u][0];
                } else {
                    dm_f2 h0 = dm2_softplus100_lean(dm_f2{acc[u][0], acc[u][1]}) * dm2_s(sc), h1 = dm2_softplus100_lean(dm_f2{acc[u][2], acc[u][3]}) * dm2_s(sc);
                    float hv[4] = {h0.x, h0.y, h1.x, h1.y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                                    // columns past N: the PE part of the skip input (idr.py:86-87), else zero padding
                        const int col = ct * 16 + 4 * q + e;
                        if (col >= N) {
                            const int j = col - N;
                            float v = 0.0f;
                            if (to_skip && j < d0) v = mv_x3_flush(__uint_as_float(ld_flag((const unsigned*)(p.pe_side + (((size_t)x * p.T + t) * 40 + j) * 16 + r))));
                            hv[e] = v;
                        }
                    }
                    store_cols(p, x, t, (l + 1) & 1, ct, r, q, dm_f2{hv[0], hv[1]}, dm_f2{hv[2], hv[3]});
                }
            }
        }
        PH(3)
#pragma unroll
        for (int u = 0; u < TPI; ++u) {
            const int t = tile_of(it, u);
            if (t >= p.T) continue;
            if (last) { if (ct0 < ntot) set_flag(pb_flag(p, x, t, 9), (unsigned)rd); }
            else {
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) if (ct0 + tt < ntot) set_flag(pb_flag(p, x, t, l + 1) + ct0 + tt, (unsigned)rd);
            }
        }
        PH(4)
        if (it + 1 < items && !fetched) {
            if (ready(it + 1, true) < 0) return;
            issue(it + 1);
            ph_[7] += 1;
        }
        PH(5)
    }
    if (probe) for (int i = 0; i < 8; ++i) p.ph[i] = ph_[i];
}

// ---- last layer (column 0 only): wave-level, tiles t = w (mod nw); all 24 fragments of a tile requested at once ----
__device__ void last_stage(const MvNetBs<3, 3>& net, const PipeBufs& p, int x, int w, int nw, int lane) {
    const int r = lane & 15, q = lane >> 4, l = net.n_layers - 1;
    const MvLayerBf& L = net.L[l];
    const int KB = L.KB, nprod = (L.K + 15) / 16;
    uint4 b[PIPE_KBMAX][3];
#pragma unroll
    for (int kb = 0; kb < PIPE_KBMAX; ++kb)
#pragma unroll
        for (int j = 0; j < 3; ++j) b[kb][j] = L.wp[(((size_t)(kb < KB ? kb : KB - 1)) * 3 + j) * 64 + lane];
    const f32x4 bias4 = *(const f32x4*)(L.bias + 4 * q);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.abuf, 0, 0x7fffffff, 0x00020000);
    for (int rd = 1; rd <= p.rounds; ++rd) {
        for (int t = w; t < p.T; t += nw) {
            if (!wait_flags(p, pb_flag(p, x, t, l), nprod, (unsigned)rd, lane)) return;
            uint4 a[PIPE_KBMAX][3];
#pragma unroll
            for (int kb = 0; kb < PIPE_KBMAX; ++kb)
#pragma unroll
                for (int s = 0; s < 3; ++s) a[kb][s] = ld_frag(rs, pb_frag(p, x, t, l & 1, s, kb < KB ? kb : KB - 1), lane);
            f32x4 acc = bias4;
#pragma unroll
            for (int kb = 0; kb < PIPE_KBMAX; ++kb)
                if (kb < KB) acc = kblock_mfma(acc, b[kb], a[kb]);
            if (q == 0) p.out[((size_t)x * p.T + t) * 16 + r] = acc[0];
            set_flag(pb_flag(p, x, t, 9), (unsigned)rd);
        }
    }
}

// roles per XCD (64 workgroups of 256 threads = two per CU): 0-3: PE stage (16 waves); 4, 5: layer 0 (parity 0 / 1; 4 column tiles per wave); then 8 per hidden
// layer (4 column groups x 2 parities); the last two: the last layer (8 waves)
__global__ __launch_bounds__(256, 2) void k_pipe(MvNetBs<3, 3> net, PipeBufs p) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds_a[];
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = net.n_layers, nh = nl - 2;
    if (j < 4) pe_stage(net, p, x, 4 * j + w, 16, lane);                           // 16 waves: two row tiles each at 32 tiles per XCD
    else if (j < 6) layer_wg<4, 2>(net, p, x, 0, 0, j - 4, w, lane, lds_a);
    else if (j < 6 + 8 * nh) { const int k = j - 6; layer_wg<1, PIPE_KBMAX>(net, p, x, 1 + k / 8, (k % 8) >> 1, k & 1, w, lane, lds_a); }
    else if (j < 8 + 8 * nh) last_stage(net, p, x, 4 * (j - 6 - 8 * nh) + w, 8, lane);
}

// ---- reference: the product engine, one workgroup per tile (the rounds probe) ----
__global__ __launch_bounds__(512) void k_rounds(MvNetBs<3, 3> net, const float* __restrict__ x, int rounds, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, d0 = 3 + 6 * net.multires;
    float* act = smem;
    float* pe = act + 16 * net.S;
    float* pts = pe + ((16 * d0 + 3) & ~3);
    float* out = pts + 16 * 4;
    for (int i = tid; i < 16 * 3; i += 512) pts[i] = x[(size_t)blockIdx.x * 16 * 3 + i];
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        mv_sdf_eval_col0<1, 2, 8, true, 3, 3>(net, act, pe, pts, out, tid);
        if (r + 1 < rounds && tid < 16) pts[3 * tid] = pts[3 * tid] + 1e-3f * out[tid];
        __syncthreads();
    }
    if (tid < 16) y[(size_t)blockIdx.x * 16 + tid] = out[tid];
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 14, T = argc > 2 ? atoi(argv[2]) : 32;
    const int W = 256, d0 = 39, nl = 9;
    int K[9], N[9];
    for (int l = 0; l < nl; ++l) { K[l] = W; N[l] = W; }
    K[0] = d0; N[3] = W - d0; N[8] = 1;
    MvNetBs<3, 3> net = {};
    net.n_layers = nl; net.skip_mask = 1u << 4; net.multires = 6;
    int maxk = 0;
    srand(1);
    for (int l = 0; l < nl; ++l) {
        MvLayerBf& L = net.L[l];
        L.K = K[l]; L.N = N[l]; L.nsplit = 0; L.KB = mv_bf_kb(K[l], 0); L.NT = mv_ceil16(N[l]) / 16;
        maxk = L.KB * 32 > maxk ? L.KB * 32 : maxk;
        // three-term packs of random fp32 weights (w = t0 + t1 + t2 exactly), padding rows / columns zero: wp[((ct * KB + kb) * 3 + j) * 64 + lane][8]
        const size_t el = 3 * mv_packed_bf16_elems(N[l], K[l], 0);
        std::vector<uint16_t> h(el, 0);
        const float scale = 1.6f / sqrtf((float)K[l]);
        for (int ct = 0; ct < L.NT; ++ct)
            for (int kb = 0; kb < L.KB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int n = ct * 16 + (lane & 15), k = kb * 32 + 8 * (lane >> 4) + i;
                        float wv = (n < N[l] && k < K[l]) ? ((rand() & 0xffff) / 65536.0f - 0.5f) * scale : 0.0f;
                        for (int j = 0; j < 3; ++j) {
                            const uint16_t tb = mv_f2bf(wv);
                            h[((((size_t)ct * L.KB + kb) * 3 + j) * 64 + lane) * 8 + i] = tb;
                            uint32_t u = (uint32_t)tb << 16; float tf; memcpy(&tf, &u, 4);
                            wv -= tf;
                        }
                    }
        std::vector<float> bv(mv_ceil16(N[l]) + 16, 0.0f);
        for (int n = 0; n < N[l]; ++n) bv[n] = ((rand() & 0xffff) / 65536.0f - 0.5f) * 0.02f;
        void *dw, *db;
        (void)hipMalloc(&dw, el * 2); (void)hipMemcpy(dw, h.data(), el * 2, hipMemcpyHostToDevice);
        (void)hipMalloc(&db, bv.size() * 4); (void)hipMemcpy(db, bv.data(), bv.size() * 4, hipMemcpyHostToDevice);
        L.wp = (const uint4*)dw; L.bias = (const float*)db;
    }
    net.S = 3 * ((maxk + 8) / 2);
    const int tiles = 8 * T;
    std::vector<float> hx((size_t)tiles * 16 * 3);
    for (auto& v : hx) v = (rand() & 0xffff) / 65536.0f - 0.5f;
    // ---- reference
    float *dx, *dy;
    (void)hipMalloc(&dx, hx.size() * 4); (void)hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&dy, (size_t)tiles * 16 * 4);
    const size_t lds = ((size_t)16 * net.S + ((16 * d0 + 3) & ~3) + 16 * 4 + 16) * 4;
    (void)hipFuncSetAttribute((const void*)k_rounds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms_ref = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rounds, dim3(tiles), dim3(512), lds, 0, net, dx, rounds, dy);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms_ref, e0, e1);
    }
    std::vector<float> yref((size_t)tiles * 16);
    (void)hipMemcpy(yref.data(), dy, yref.size() * 4, hipMemcpyDeviceToHost);
    // ---- pipeline
    PipeBufs p;
    p.T = T; p.rounds = rounds;
    const size_t abuf_el = (size_t)8 * T * 2 * 3 * PIPE_KBMAX * 64;
    (void)hipMalloc(&p.abuf, abuf_el * 16); (void)hipMemset(p.abuf, 0, abuf_el * 16);
    (void)hipMalloc(&p.pe_side, (size_t)8 * T * 40 * 16 * 4); (void)hipMemset(p.pe_side, 0, (size_t)8 * T * 40 * 16 * 4);
    (void)hipMalloc(&p.pts, (size_t)8 * T * 16 * 4 * 4);
    (void)hipMalloc(&p.out, (size_t)8 * T * 16 * 4);
    (void)hipMalloc(&p.flags, (size_t)8 * T * 10 * 16 * 4);
    (void)hipMalloc(&p.abort_flag, 4);
    (void)hipMalloc(&p.ph, 64); (void)hipMemset(p.ph, 0, 64);
    std::vector<float> hp((size_t)8 * T * 16 * 4, 0.0f);
    for (int x = 0; x < 8; ++x)
        for (int t = 0; t < T; ++t) {
            const int b = t * 8 + x;                                                  // reference workgroup b <-> (xcd x, tile t)
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 3; ++c) hp[(((size_t)x * T + t) * 16 + r) * 4 + c] = hx[((size_t)b * 16 + r) * 3 + c];
        }
    float ms = 0;
    const size_t plds = (size_t)PIPE_TPI * 3 * PIPE_KBMAX * 64 * 16;
    (void)hipFuncSetAttribute((const void*)k_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
    std::vector<float> y((size_t)tiles * 16);
    unsigned ab = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemcpy(p.pts, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemset(p.flags, 0, (size_t)8 * T * 10 * 16 * 4);
        (void)hipMemset(p.abort_flag, 0, 4);
        (void)hipMemset(p.out, 0, (size_t)8 * T * 16 * 4);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_pipe, dim3(8 * (8 + 8 * (nl - 2))), dim3(256), plds, 0, net, p);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&ab, p.abort_flag, 4, hipMemcpyDeviceToHost);
    }
    hipError_t err = hipGetLastError();
    std::vector<float> yo((size_t)8 * T * 16);
    (void)hipMemcpy(yo.data(), p.out, yo.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; double maxd = 0;
    for (int x = 0; x < 8; ++x)
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 16; ++r) {
                const float a = yo[((size_t)x * T + t) * 16 + r], b = yref[((size_t)(t * 8 + x)) * 16 + r];
                uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
                if (ua != ub) { ++bad; const double d = fabs((double)a - b); if (d > maxd) maxd = d; }
            }
    printf("%s; abort flag %u\n", hipGetErrorString(err), ab);
    printf("%d tiles of 16 rows (%d per XCD), %d dependent rounds: engine (one workgroup per tile, weights streamed) %.1f us per round; pipeline (weights in registers, %d workgroups of 256) %.1f us per round\n",
           tiles, T, rounds, 1e3 * ms_ref / rounds, 8 * (8 + 8 * (nl - 2)), 1e3 * ms / rounds);
    unsigned long long hph[8]; (void)hipMemcpy(hph, p.ph, 64, hipMemcpyDeviceToHost);
    const double ni = (double)rounds * ((((T + PIPE_TPI - 1) / PIPE_TPI) + 1) / 2);
    printf("layer-2 workgroup, us per item: regs->LDS %.2f | barrier %.2f | try-poll + issue next %.2f | multiply + epilogue + stores %.2f | store ack + flag %.2f | blocking wait + issue %.2f  (blocking waits: %.0f of %.0f items)\n",
           hph[0] * 0.01 / ni, hph[1] * 0.01 / ni, hph[2] * 0.01 / ni, hph[3] * 0.01 / ni, hph[4] * 0.01 / ni, hph[5] * 0.01 / ni, (double)hph[7], ni);
    printf("outputs differing from the engine bit for bit: %zu of %d (max |d| %.3g); sample %g %g\n", bad, tiles * 16, maxd, yo[0], yref[0]);
    return 0;
}
