// mlp_common.h -- device-side description of a folded MLP and the MFMA-packed weight layout.
//
// Packed layout of one Linear (out = N, in = K), consumed by v_mfma_f32_16x16x4_f32 without any
// LDS staging of weights: Kp = ceil32(K), Np = ceil16(N),
//     Wp[ct][kb][lane][s] = W[ct*16 + (lane & 15)][kb*16 + 4*s + (lane >> 4)]      (0 outside N x K)
// i.e. one 16-byte load per lane (1 KiB per wave, fully coalesced) yields the B operands of the four
// MFMA k-steps of a 16-wide k-block, and the k-steps are issued in ascending k: the accumulation is a
// k-ascending fmaf chain from 0, bit-identical to a scalar fmaf loop.
// The transposed pack (for v = s W, i.e. contraction over the OUT dimension) uses the same formula on W^T.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MV_MAXL 12

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MvLayer {
    const float4* wp;   // packed [NT][KB][64] float4
    const float* bias;  // [N]
    int K, N;           // true in / out
    int KB, NT;         // k-blocks of 16, column tiles of 16
};

struct MvNet {
    MvLayer L[MV_MAXL];
    int n_layers;
    unsigned skip_mask; // bit l set: the input of layer l is cat([x, PE]) / sqrt(2)  (idr.py:86-87; the shipped conf: layer 4 only)
    int multires;       // PE frequencies; d_pe = 3 + 6*multires
    int S;              // LDS activation row stride in floats (== 8 mod 64: conflict-free ds_read_b128 A fragments)
};

__host__ __device__ static inline bool mv_skip_at(unsigned mask, int l) { return l >= 0 && ((mask >> l) & 1u) != 0; }
__host__ __device__ static inline int mv_ceil16(int x) { return (x + 15) & ~15; }
// K is padded to a multiple of 32 (an EVEN number of 16-wide k-blocks) so the 2x-unrolled pipelined loop has no tail
__host__ __device__ static inline int mv_kpad(int x) { return (x + 31) & ~31; }
__host__ __device__ static inline size_t mv_packed_floats(int N, int K) {
    return (size_t)mv_ceil16(N) * mv_kpad(K);
}
// position of logical column c inside an LDS activation row: 4x4 transpose within each 16-block so that a
// lane's four consecutive k-steps (k = 4s + q) are one contiguous float4 at 4q..4q+3.
__host__ __device__ static inline int mv_perm(int c) { return (c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3); }
