// rows_engine_bf16.h -- the THROUGHPUT form of the split-activation tracing MLP (tile_engine_bf16s.h) for the row-rich launches: sampler windows,
// min-sdf rows, SDF grids.
//
// tile_engine_bf16s.h splits a layer's COLUMNS over the eight waves of a workgroup that owns 16-64 rows: every wave re-reads the whole activation
// tile from LDS, every workgroup streams the whole weight set (1.1 MB) from L2 per 32 rows, and each layer is two workgroup barriers with the
// epilogue between them -- the right shape for the dependent 16-row evaluations of the sphere tracer, not for 10^5 independent rows.  Here
// (north_star's literal design: "weight tiles staged in LDS and reused across a wavefront's rays"):
//   * a workgroup = 4 waves = 128 rows; every wave OWNS 32 rows for the whole network.  Its activations never leave its registers: with the
//     WEIGHTS as the first operand of v_mfma_f32_32x32x16_bf16 (M = 32 output features, N = the wave's 32 rows) a lane ends up with 16 outputs
//     of ONE row -- and with the output features of a tile assigned to the matrix rows in the order
//         feature(m) = 16 (m >> 4) + 8 ((m >> 2) & 1) + 4 ((m >> 3) & 1) + (m & 3)
//     those 16 values are exactly the lane's halves of the next layer's k-blocks 2T and 2T + 1 (k = 16 c + 8 (lane >> 5) + j): softplus, term
//     split, pack -- no LDS, no shuffle, no barrier between layers;
//   * the weights stream global -> LDS ONCE per workgroup in chunks of two 32-feature tiles x all k (32 KB), three-slot ring, by
//     global_load_lds_dwordx4 (no staging registers).  The LDS image is assembled from the EXISTING bf16 pack (tile_engine_bf16.h) by per-lane
//     source addresses, so no second pack exists; one s_barrier per chunk is the only synchronisation;
//   * the epilogue of tile pair p runs interleaved with the matrix instructions of pair p + 1 (one wave per SIMD: nobody else hides it).
// Arithmetic = tile_engine_bf16s.h's (same term split, same softplus, bias as the accumulator's start value); the matrix core's internal summation
// order differs with the instruction shape, so results agree to accumulation noise, not bit for bit (tests/test_gpu_rows.py).
// Covers the networks whose layer inputs are 16 k-blocks (240 < K <= 256) or at most 4 (K <= 64): the 8 x 256 net of BASELINE.json and the 64-wide test
// nets; everything else keeps the column-split engine (mv_ro_fits).  Every layer runs the same number of k-blocks (a shorter input is zero-padded in
// the registers and meets clamped, finite weight k-blocks), the pair index is a run-time value and the two activation register sets swap roles from
// layer to layer: one layer body per role, ~45 KB of code -- the kernel has to stay inside the 64 KB instruction cache.
#pragma once
#include <type_traits>
#include "tile_engine_bf16s.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// dev-only ablation switches (tools/build_ro_ablations.sh; never set in the shipped build): 1 = activation -> plain conversion, 2 = no matrix instructions,
// 4 = no weight loads, 8 = no LDS reads of the weight fragments, 16 = no chunk barriers
#ifndef MV_RO_ABLATE
#define MV_RO_ABLATE 0
#endif
// VALU instructions placed behind each matrix instruction of a k-block (sched_group_barrier)
#ifndef MV_RO_VPER
#define MV_RO_VPER(NS) 7
#endif
#define MV_RO_THREADS 256
#define MV_RO_ROWS 128
#define MV_RO_RING 3
#define MV_RO_KBM 16                                         // k-blocks of 16 inputs per layer
#define MV_RO_NLD 8                                          // global_load_lds per wave and chunk (2 * KBM blocks of 1 KiB over 4 waves)
#define MV_RO_SLOT (2 * MV_RO_KBM * 1024)                    // bytes of one ring slot

__host__ __device__ constexpr size_t mv_ro_lds_bytes(int n_layers) {
    return (size_t)MV_RO_RING * MV_RO_SLOT + (size_t)n_layers * 1024 + 4 * 32 * 40 * 4 + MV_RO_ROWS * 3 * 4 + MV_RO_ROWS * 4;
}

struct MvRoLds {
    char* ring; float* bias; float* pe; float* pts; float* out;
};
__device__ __forceinline__ MvRoLds mv_ro_carve(char* base, int n_layers) {
    MvRoLds l;
    l.ring = base;
    l.bias = (float*)(base + MV_RO_RING * MV_RO_SLOT);
    l.pe = l.bias + n_layers * 256;
    l.pts = l.pe + 4 * 32 * 40;
    l.out = l.pts + MV_RO_ROWS * 3;
    return l;
}

// can this network run on the row-owner engine?
__host__ inline bool mv_ro_fits(const MvNetBf& net) {
    if (net.n_layers < 2 || net.n_layers > MV_MAXL) return false;
    if (3 + 6 * net.multires > 40) return false;
    for (int l = 0; l < net.n_layers; ++l) {
        const int kb16 = 2 * net.L[l].KB;
        if (!(kb16 <= 4 || kb16 == 16) || net.L[l].nsplit != 0) return false;
        if (l < net.n_layers - 1 && net.L[l].N > 256) return false;
    }
    return net.L[0].KB <= 2;
}
// k-blocks of 16 every layer runs: 16 (some layer input is 241..256 wide) or 4
__host__ inline int mv_ro_kbc(const MvNetBf& net) {
    for (int l = 0; l < net.n_layers; ++l) if (2 * net.L[l].KB == 16) return 16;
    return 4;
}

// ---- chunk loader: tiles 2p, 2p + 1 of layer l -> ring slot.  Block b = tloc * 16 + c of the slot is the A operand (32 features x 16 k) of tile
// T = 2p + tloc, k-block c: lane (m = lane & 31, hh = lane >> 5) holds W[feature(m)][16 c + 8 hh .. + 8], fetched from the 16-column-tile pack:
// column tile ct = 2 T + (m >> 4), k-block kb = c >> 1, pack lane (r = 8 ((m >> 2) & 1) + 4 ((m >> 3) & 1) + (m & 3), q = 2 (c & 1) + hh).
// Wave w loads the k-blocks c = w, w + 4, w + 8, w + 12 of both tiles: 8 pieces per chunk, always (a short layer re-loads its k-block: the
// s_waitcnt vmcnt() counts stay compile-time constants).  A piece = wave-uniform address (SGPR pair: one base per chunk and tile, a constant stride
// per piece) + a per-lane byte offset that only changes with the layer: no vector address arithmetic per load.
// The LDS-DMA instruction is issued from inline asm (M0 = the wave-uniform LDS address, saved and restored in the same statement): with the builtin
// the compiler treats every later ds_read of the ring as a possible reader of the DMA's destination and puts s_waitcnt vmcnt(0) in front of it
// (seen in the ISA: one per k-block) -- the prefetch distance of two chunks was gone.  The counted waits are ours (mv_ro_eval::chunk_sync).
// One such instruction costs ~130 cycles of the wave's issue time (ablation: no loads = -18 us of a 78 us pass), so the pieces of a chunk are
// issued one every other k-block of the chunk being computed, between its matrix instructions, not as a batch at its start.
__device__ __forceinline__ void mv_ro_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned mv_lds_addr(const void* p) {
    return (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p);
}
struct MvRoLoader {
    const MvNetBf* net;
    unsigned ring_lds;
    int w, lane, nl;
    int l, p;                         // the chunk the next piece() calls belong to (l >= nl: past the end, the last chunk again)
    // per layer
    int KB, ntl, stride;              // k-blocks of 32; index of the layer's last 32-feature tile; bytes between this wave's k-blocks (0: a short layer)
    bool odd;                         // the last tile's upper 16 features do not exist in the pack (odd number of 16-feature column tiles)
    const char* wl;                   // the pack + this wave's first k-block
    unsigned voff, voff_last;         // per-lane byte offsets (voff_last: for that last, half tile)
    // per chunk
    const char* base[2];              // tile 0 / tile 1
    unsigned vo[2];
    unsigned lds_dst;                 // LDS address of this wave's first block in the chunk's slot
    int slot;
    __device__ __forceinline__ int npairs(int ll) const { const int nt32 = (ll == nl - 1) ? 1 : (net->L[ll].N + 31) >> 5; return (nt32 + 1) >> 1; }
    __device__ __forceinline__ void set_layer() {
        const MvLayerBf& L = net->L[l < nl ? l : nl - 1];
        const int m = lane & 31, hh = lane >> 5, r = 8 * ((m >> 2) & 1) + 4 * ((m >> 3) & 1) + (m & 3);
        KB = L.KB;
        const int kb16 = 2 * KB, cw = w < kb16 ? w : kb16 - 1;
        stride = kb16 == 16 ? 2 * 64 * 16 * 2 : 0;                  // k-block c + 4: two k-blocks of 32 further
        wl = (const char*)L.wp + ((size_t)(cw >> 1) * 64 + 32 * (cw & 1)) * 16;
        const int NT = (l >= nl - 1) ? 1 : L.NT;                    // the last layer: column tile 0 only
        ntl = (NT - 1) >> 1; odd = (NT & 1) != 0;
        voff_last = (unsigned)(16 * hh + r) * 16u;
        voff = NT == 1 ? voff_last : voff_last + (unsigned)((m >> 4) * KB * 64) * 16u;
    }
    __device__ __forceinline__ void set_chunk() {
        const int pp = l < nl ? p : 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int T = 2 * pp + t;
            T = T < ntl ? T : ntl;
            base[t] = wl + (size_t)(2 * T) * KB * 64 * 16;
            vo[t] = (odd && T == ntl) ? voff_last : voff;
        }
        lds_dst = ring_lds + slot * MV_RO_SLOT + w * 1024;
    }
    __device__ __forceinline__ void init(const MvNetBf& n_, unsigned ring, int w_, int lane_) {
        net = &n_; ring_lds = ring; w = w_; lane = lane_; nl = n_.n_layers; l = 0; p = 0; slot = 0;
        set_layer();
        set_chunk();
    }
    // piece j (0 .. 7) of the current chunk: tile j >> 2, this wave's k-block number j & 3
    __device__ __forceinline__ void piece(int j) const {
        if (MV_RO_ABLATE & 4) return;
        const int t = j >> 2, i = j & 3;
        mv_ro_glds16(base[t] + i * stride, vo[t], __builtin_amdgcn_readfirstlane(lds_dst + (t * MV_RO_KBM + 4 * i) * 1024));
    }
    __device__ __forceinline__ void advance() {
        slot = slot == MV_RO_RING - 1 ? 0 : slot + 1;
        if (l < nl) { if (++p == npairs(l)) { p = 0; ++l; set_layer(); } }
        set_chunk();
    }
    __device__ __forceinline__ void issue_all() {
#pragma unroll
        for (int j = 0; j < MV_RO_NLD; ++j) piece(j);
        advance();
    }
};

// the activation function + conversion to the B-operand terms of two neighbouring outputs (one 32-bit word per term)
template <int NS>
__device__ __forceinline__ void mv_ro_act_pair(float z0, float z1, float sc, uint32_t (&p)[NS]) {
    if constexpr ((MV_RO_ABLATE & 1) != 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) p[s] = mv_f2bf_pk(z0 * sc, z1);
    } else {
        const dm_f2 h = dm_f2{mv_softplus100_acc1(z0) * sc, mv_softplus100_acc1(z1) * sc};
        mv_split_pk<NS>(h, p);
    }
}

// One tile pair of a layer: KBC k-blocks of matrix instructions into acc[A][0..1] (which hold the biases), with -- PREV -- the epilogue of the
// PREVIOUS pair (acc[1 - A] -> xo, the four next-layer k-blocks that pair produces) and the LDS-DMA pieces of the chunk two ahead spread over the
// k-blocks.  Everything is statically indexed: straight-line code.  One wave per SIMD: nobody else fills the matrix pipe's shadow, so the order
// inside a k-block is pinned --
//     ds_read (A fragments of k-block c + 1) | MFMA | a share of the epilogue slice | MFMA | ... | LDS-DMA piece
// (sched_group_barrier), and a full scheduling barrier per k-block keeps the compiler from hoisting the whole pair's LDS reads to the top
// (seen: 512 registers and spills).  KBC == 0: no matrix work, only the epilogue (the layer's last pair).
// The pair index is a RUN-TIME value: only the accumulator parity A is static, so a layer is three instances of this body, not four per input
// width -- the kernel has to stay inside the 64 KB instruction cache.
template <int NS, int KBC, int A, bool PREV, bool TWOACC>
__device__ __forceinline__ void mv_ro_pair(const char* slot, const uint4 (&xin)[NS][MV_RO_KBM], uint4 (&xo)[NS][4], f32x16 (&acc)[TWOACC ? 2 : 1][2], float sc,
                                           const MvRoLoader& ld) {
    constexpr int KBM = MV_RO_KBM, NLD = MV_RO_NLD, AI = TWOACC ? A : 0, AP = TWOACC ? 1 - A : 0;
    constexpr int NIT = PREV ? 16 : KBC;                            // k-block iterations (the epilogue slices of a short layer outlast its matrix work)
    constexpr int LDSTEP = NIT >= 2 * NLD ? 2 : 1;                  // a DMA piece every LDSTEP-th iteration
    uint4 a0 = uint4{0u, 0u, 0u, 0u}, a1 = a0;
    if (KBC > 0) { a0 = *(const uint4*)(slot); a1 = *(const uint4*)(slot + KBM * 1024); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NIT; ++c) {
        uint4 n0 = a0, n1 = a1;
        if (c + 1 < KBC && !(MV_RO_ABLATE & 8)) { n0 = *(const uint4*)(slot + (c + 1) * 1024); n1 = *(const uint4*)(slot + (KBM + c + 1) * 1024); }
        if (c < KBC && !(MV_RO_ABLATE & 2)) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                acc[AI][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mv_bf8, a0), __builtin_bit_cast(mv_bf8, xin[s][c]), acc[AI][0], 0, 0, 0);
                acc[AI][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mv_bf8, a1), __builtin_bit_cast(mv_bf8, xin[s][c]), acc[AI][1], 0, 0, 0);
            }
        }
        if (PREV) {                                                 // outputs 2 pi, 2 pi + 1 of tile c >> 3 of the previous pair
            const int t = c >> 3, pi = c & 7;
            uint32_t pw_[NS];
            mv_ro_act_pair<NS>(acc[AP][t][2 * pi], acc[AP][t][2 * pi + 1], sc, pw_);
#pragma unroll
            for (int s = 0; s < NS; ++s) ((uint32_t*)&xo[s][2 * t + (pi >> 2)])[pi & 3] = pw_[s];
        }
        if (KBC > 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (c < KBC) {
#pragma unroll
                for (int i = 0; i < 2 * NS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, MV_RO_VPER(NS), 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (c % LDSTEP == 0 && c / LDSTEP < NLD) ld.piece(c / LDSTEP);
        }
        a0 = n0; a1 = n1;
    }
    if (KBC > 0) {
#pragma unroll
        for (int j = (NIT + LDSTEP - 1) / LDSTEP; j < NLD; ++j) ld.piece(j);    // (a short layer: the pieces that found no k-block)
    }
}

// the four k-blocks pair q produced -> their place in the next layer's input (q is a run-time value: a switch over register moves)
template <int NS>
__device__ __forceinline__ void mv_ro_scatter(const uint4 (&xo)[NS][4], uint4 (&xout)[NS][MV_RO_KBM], int q) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        if (q == qq) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) xout[s][4 * qq + i] = xo[s][i];
            // (an immediate operand that differs per case: without it the optimiser merges the four cases' stores into ONE store through a selected
            // pointer, and an array addressed through a run-time pointer lives in scratch memory, not in registers -- seen: 1 KB of scratch)
            asm volatile("; scatter case %0" ::"n"(qq));
        }
    }
}

// One Linear + softplus of the wave's 32 rows: xin -> xout (both in registers), KBC k-blocks of input (16: a 256-wide input, the PE input of layer 0
// zero-padded to it -- the clamped weight k-blocks then meet zero activations; 4: the 64-wide test nets).
template <int NS, int KBC>
__device__ __forceinline__ void mv_ro_layer(const MvNetBf& net, int l, const MvRoLds& lds, const uint4 (&xin)[NS][MV_RO_KBM], uint4 (&xout)[NS][MV_RO_KBM],
                                            MvRoLoader& ld, int& cs_slot, const float* per, int lane, int h) {
    constexpr bool TWOACC = NS < 3;                                 // NS = 3: one accumulator set (two sets + 2 x 192 activation registers do not fit 512)
    const MvLayerBf& L = net.L[l];
    const int np = ld.npairs(l);
    const bool to_skip = mv_skip_at(net.skip_mask, l + 1);
    const float sc = to_skip ? 0.7071067690849304f : 1.0f;          // cat([x, input]) / sqrt(2), idr.py:86-87
    const float* bl = lds.bias + l * 256 + 8 * h;
    f32x16 acc[TWOACC ? 2 : 1][2];                                  // [pair parity][tile of the pair]
    uint4 xo[NS][4];
    auto begin = [&](int p, auto AC) {                              // chunk p ready; accumulators <- biases
        constexpr int A = TWOACC ? decltype(AC)::value : 0;
        if constexpr ((MV_RO_ABLATE & 16) != 0) asm volatile("" ::: "memory");
        else if constexpr ((MV_RO_ABLATE & 4) != 0) asm volatile("s_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");   // this wave's loads of the chunk have landed (the 8 of the next may fly), then everybody's
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* bt = bl + 32 * (2 * p + t);
            const f32x4 b0 = *(const f32x4*)(bt), b1 = *(const f32x4*)(bt + 4), b2 = *(const f32x4*)(bt + 16), b3 = *(const f32x4*)(bt + 20);
            acc[A][t] = f32x16{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
        }
    };
    auto slot_ptr = [&]() {
        const char* sp = lds.ring + cs_slot * MV_RO_SLOT + lane * 16;
        cs_slot = cs_slot == MV_RO_RING - 1 ? 0 : cs_slot + 1;
        return sp;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if constexpr (TWOACC) {
        begin(0, I0{});
        mv_ro_pair<NS, KBC, 0, false, true>(slot_ptr(), xin, xo, acc, sc, ld);
        ld.advance();
        int p = 1;
        for (; p < np; p += 2) {
            begin(p, I1{});
            mv_ro_pair<NS, KBC, 1, true, true>(slot_ptr(), xin, xo, acc, sc, ld);
            ld.advance();
            mv_ro_scatter<NS>(xo, xout, p - 1);
            if (p + 1 < np) {
                begin(p + 1, I0{});
                mv_ro_pair<NS, KBC, 0, true, true>(slot_ptr(), xin, xo, acc, sc, ld);
                ld.advance();
                mv_ro_scatter<NS>(xo, xout, p);
            }
        }
        // the layer's last pair: nothing left to hide its epilogue under
        if ((np - 1) & 1) mv_ro_pair<NS, 0, 0, true, true>(nullptr, xin, xo, acc, sc, ld);
        else mv_ro_pair<NS, 0, 1, true, true>(nullptr, xin, xo, acc, sc, ld);
        mv_ro_scatter<NS>(xo, xout, np - 1);
    } else {
        for (int p = 0; p < np; ++p) {
            begin(p, I0{});
            mv_ro_pair<NS, KBC, 0, false, false>(slot_ptr(), xin, xo, acc, sc, ld);
            ld.advance();
            mv_ro_pair<NS, 0, 0, true, false>(nullptr, xin, xo, acc, sc, ld);
            mv_ro_scatter<NS>(xo, xout, p);
        }
    }
    if (to_skip) {                                                  // the PE part behind this layer's outputs (slots N ..; everything past it zero)
        const int N = L.N;
#pragma unroll
        for (int c = 0; c < MV_RO_KBM; ++c) {
            if (16 * c + 16 > N && 16 * c < N + 40) {               // (wave-uniform) the k-blocks the PE part touches
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = 16 * c + 8 * h + 2 * jj, j0 = k - N, j1 = j0 + 1;
                    // dm_div_sqrt2 of the PE value BEFORE the split (the column-split engine's order); pe[] is zero beyond d0
                    const float v0 = dm_div_sqrt2(per[j0 < 0 ? 0 : (j0 > 39 ? 39 : j0)]), v1 = dm_div_sqrt2(per[j1 < 0 ? 0 : (j1 > 39 ? 39 : j1)]);
                    uint32_t pw_[NS];
                    mv_split_pk<NS>(dm_f2{j0 > 39 ? 0.0f : v0, j1 > 39 ? 0.0f : v1}, pw_);
                    const uint32_t keep = (j0 >= 0 ? 0u : 0x0000ffffu) | (j1 >= 0 ? 0u : 0xffff0000u);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        uint32_t& dst = ((uint32_t*)&xout[s][c])[jj];
                        dst = (dst & keep) | (pw_[s] & ~keep);
                    }
                }
            }
        }
    }
}

// the last Linear: output column 0 = matrix row 0 of tile 0 = accumulator element 0 of the lanes with h == 0
template <int NS, int KBC>
__device__ __forceinline__ void mv_ro_last(const MvNetBf& net, const MvRoLds& lds, const uint4 (&xin)[NS][MV_RO_KBM], MvRoLoader& ld, int cs_slot, int w, int lane) {
    const int nl = net.n_layers;
    if constexpr ((MV_RO_ABLATE & 16) != 0) asm volatile("" ::: "memory");
    else if constexpr ((MV_RO_ABLATE & 4) != 0) asm volatile("s_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    ld.issue_all();
    const char* slot = lds.ring + cs_slot * MV_RO_SLOT + lane * 16;
    const float b0 = lds.bias[(nl - 1) * 256];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = b0;
#pragma unroll
    for (int c = 0; c < KBC; ++c) {
        const uint4 a0 = *(const uint4*)(slot + c * 1024);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mv_bf8, a0), __builtin_bit_cast(mv_bf8, xin[s][c]), acc, 0, 0, 0);
    }
    if ((lane >> 5) == 0) lds.out[w * 32 + (lane & 31)] = acc[0];
}

// ImplicitNetwork.forward(...)[:, 0] for the workgroup's 128 rows (points in LDS pts[128][3]); results -> LDS out[128].  All 256 threads call.
// KBC: k-blocks of 16 every layer runs (16 or 4: mv_ro_fits / mv_ro_kbc).
template <int NS, int KBC>
__device__ void mv_ro_eval(const MvNetBf& net, const MvRoLds& lds, int tid) {
    constexpr int KBM = MV_RO_KBM;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, h = lane >> 5;
    const int nl = net.n_layers, d0 = 3 + 6 * net.multires;
    // ---- biases -> LDS (zero padded to 256 per layer), positional encoding of the wave's rows -> LDS pe[w][row][40] (zero padded)
    for (int i = tid; i < nl * 256; i += MV_RO_THREADS) {
        const int l = i >> 8, c = i & 255;
        lds.bias[i] = c < net.L[l].N ? net.L[l].bias[c] : 0.0f;
    }
    float* pew = lds.pe + w * 32 * 40;
    {
        const float* pw = lds.pts + w * 32 * 3;
        const int T = 3 * net.multires + 1;
        for (int task = lane; task < 32 * T; task += 64) {
            const int row = task / T, j = task - row * T;
            const float* x = pw + row * 3;
            float* pr = pew + row * 40;
            if (j < 3 * net.multires) {
                const int mm = j / 3, c = j - 3 * mm;
                float s, co;
                dm_sincos(x[c] * (float)(1 << mm), &s, &co);
                pr[3 + 6 * mm + c] = s;
                pr[3 + 6 * mm + 3 + c] = co;
            } else {
                pr[0] = x[0]; pr[1] = x[1]; pr[2] = x[2];
                for (int c = d0; c < 40; ++c) pr[c] = 0.0f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // the bias loads are the last ordinary global loads before the chunk pipeline
    // ---- chunk pipeline: the loader runs two chunks ahead of the consumer
    MvRoLoader ld;
    ld.init(net, mv_lds_addr(lds.ring), w, lane);
    ld.issue_all();
    ld.issue_all();
    int cs_slot = 0;
    // ---- layer-0 input: the PE slots (k-blocks 0 .. 2; pe[] is zero from d0 to 40), everything else zero.  The two activation sets swap roles per layer.
    uint4 xa[NS][KBM], xb[NS][KBM];
    const float* per = pew + n * 40;
#pragma unroll
    for (int c = 0; c < KBM; ++c) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { xa[s][c] = uint4{0u, 0u, 0u, 0u}; xb[s][c] = uint4{0u, 0u, 0u, 0u}; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int k0 = 16 * c + 8 * h;                              // this lane's eight slots of the k-block: k0 .. k0 + 7
        const bool in0 = k0 < 40, in1 = k0 + 4 < 40;
        const f32x4 v0 = *(const f32x4*)(per + (in0 ? k0 : 32)), v1 = *(const f32x4*)(per + (in1 ? k0 + 4 : 32));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float a = jj < 2 ? (in0 ? v0[(2 * jj) & 3] : 0.0f) : (in1 ? v1[(2 * jj) & 3] : 0.0f);
            const float b = jj < 2 ? (in0 ? v0[(2 * jj + 1) & 3] : 0.0f) : (in1 ? v1[(2 * jj + 1) & 3] : 0.0f);
            uint32_t pw_[NS];
            mv_split_pk<NS>(dm_f2{a, b}, pw_);
#pragma unroll
            for (int s = 0; s < NS; ++s) ((uint32_t*)&xa[s][c])[jj] = pw_[s];
        }
    }
    int l = 0;
    for (; l + 1 < nl - 1; l += 2) {
        mv_ro_layer<NS, KBC>(net, l, lds, xa, xb, ld, cs_slot, per, lane, h);
        mv_ro_layer<NS, KBC>(net, l + 1, lds, xb, xa, ld, cs_slot, per, lane, h);
    }
    if (l < nl - 1) {
        mv_ro_layer<NS, KBC>(net, l, lds, xa, xb, ld, cs_slot, per, lane, h);
        mv_ro_last<NS, KBC>(net, lds, xb, ld, cs_slot, w, lane);
    } else {
        mv_ro_last<NS, KBC>(net, lds, xa, ld, cs_slot, w, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the two trailing chunk loads (dummies) land before the ring is reused
    __syncthreads();
}
