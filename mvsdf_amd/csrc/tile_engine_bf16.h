// tile_engine_bf16.h -- shared pieces of the tracing engines on gfx950's bf16 matrix cores (v_mfma_f32_16x16x32_bf16): the bf16 pack layout
//       Wp16[ct][kb][lane][i] = bf16(W[ct*16 + (lane & 15)][kb*32 + 8*(lane >> 4) + i])          (one 16-byte load per lane and k-block)
// the network descriptor, the conversions and the ring-depth constants.  The engines themselves are in tile_engine_bf16s.h (activations carried as 2 / 3 bf16
// terms: trace_dtype 3 / 4 / 5).  The round 2-4 engine that rounded the hidden ACTIVATIONS to bf16 as well (trace_dtype 1: 3 mask flips per 4096 rays, depth p99
// 1.5e-3) lived here; it was removed in round 5 -- `bf16x2` runs at its speed and is parity-checked.
#pragma once
#include "mlp_common.h"
#include "det_math.h"
#include "det_math_pk.h"

typedef short mv_bf8 __attribute__((ext_vector_type(8)));

// phase stamps of tools/micro/bf16_engine_rounds.hip (dev probe); nothing in the product build
#ifndef MV_PH
#define MV_PH_DECL
#define MV_PH(p)
#define MV_PH_END
#endif

struct MvLayerBf {
    const uint4* wp;    // packed [NT][KB][64] x 8 bf16
    const float* bias;  // [N] fp32, READ UP TO THE NEXT MULTIPLE OF 16 ENTRIES (16-byte loads of four columns; entries past N are loaded and never used).  A
                        // requirement of the C ABI, stated at MvsdfNetDesc.bias in include/mvsdf_hip.h: hipMalloc'd and torch-allocated buffers are readable
                        // that far; a caller that sub-allocates biases from its own arena pads each to a multiple of 16 floats
    int K, N;           // true in / out
    int nsplit;         // trailing input columns that enter as hi + lo pairs (layer 0: all of them; skip layer: the PE part)
    int KB, NT;         // k-blocks of 32 over K + nsplit, column tiles of 16
};

struct MvNetBf {
    MvLayerBf L[MV_MAXL];
    int n_layers;
    unsigned skip_mask;
    int multires;
    int S;              // LDS activation row stride in FLOAT units (the bf16 row holds 2*S elements), so LDS carving matches the fp32 engine
};

__host__ __device__ static inline uint16_t mv_f2bf(float f) {     // round to nearest even (finite inputs)
    uint32_t u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float mv_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
// two fp32 -> two bf16 (low half = a) with the gfx950 conversion instruction; same rounding as mv_f2bf for finite inputs
typedef float mv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mv_f2bf_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(mv_f32x2{a, b}, mv_bf16x2));
}
__host__ __device__ static inline int mv_bf_kb(int K, int nsplit) { return (K + nsplit + 31) / 32; }
__host__ __device__ static inline size_t mv_packed_bf16_elems(int N, int K, int nsplit) { return (size_t)mv_ceil16(N) * mv_bf_kb(K, nsplit) * 32; }

// weight ring depths of the engines in tile_engine_bf16s.h (CARRIED: the next layer's first k-blocks are fetched under this layer's work; tools/micro/bf16_engine_rounds.hip)
__host__ __device__ constexpr int mv_bf_pd(int NTW, bool carry) { return (carry && NTW < 4) ? 8 : 4; }   // weight k-blocks in registers per column tile
__host__ __device__ constexpr int mv_bf_pdr(int NTW, bool carry) { return carry ? mv_bf_pd(NTW, carry) / 2 : 0; }   // ... of which re-loaded inside the ring
#define MV_BF_PDA 4                                                 // activation k-blocks in flight (LDS)
