// Where does one dependent evaluation ("round") of the f32x3 tracing engine (tile_engine_bf16s.h, MvNetBs<3, 3>: fp32 weights and activations as three bf16 terms each)
// spend its time?  (dev probe, gfx950; the round-3 probe bf16_engine_rounds.hip ported to the engine the product now defaults to.)
// One 512-thread workgroup per CU runs `rounds` dependent evaluations of the fused 9-layer MLP (8x256, skip into layer 4) on its own 16*MT rows -- the situation of
// k_sphere_trace -- and wave 0 of workgroup 0 accumulates the 100 MHz wall clock between the engine's phase marks.  With wgs > 256: the throughput situation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../mvsdf_amd/csrc x3_engine_rounds.hip -o bin/x3_engine_rounds && bin/x3_engine_rounds [rounds] [wgs]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ unsigned long long g_ph[8];
#ifndef NO_PH
#define MV_PH_DECL unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tp_ = wall_clock64();
#define MV_PH(p) { const unsigned long long t_ = wall_clock64(); ph_[p] += t_ - tp_; tp_ = t_; }
#define MV_PH_END if (blockIdx.x == 0 && tid == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_ph[i_], ph_[i_]); }
#endif                                                          // -DNO_PH: no stamps at all (the stamps cost ~1-2 us per round themselves), per-round time only
#include "tile_engine_bf16s.h"

template <int MT, int NTW, bool CARRY, int NW = 8, bool PPV = false>
__global__ __launch_bounds__(64 * NW) void k_rounds(MvNetBs<3, 3> net, const float* __restrict__ x, int rounds, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x, d0 = 3 + 6 * net.multires;
    float* act = smem;
    float* pe = act + (PPV ? 2 : 1) * ROWS * net.S;         // PPV: the two ping-pong tiles of the product's k_sphere_trace (one barrier per layer)
    float* pts = pe + ((ROWS * d0 + 3) & ~3);
    float* out = pts + ROWS * 4;
    for (int i = tid; i < ROWS * 3; i += 64 * NW) pts[i] = x[(blockIdx.x % 256) * ROWS * 3 + i];
    __syncthreads();
#ifdef DESYNC
    if (blockIdx.x >= 256 && ((blockIdx.x >> 8) & 1)) for (int i = 0; i < DESYNC; ++i) __builtin_amdgcn_s_sleep(127);   // second workgroup of a CU starts late (127 x 64 clocks ~ 3.4 us per count)
#endif
    for (int r = 0; r < rounds; ++r) {
        if constexpr (PPV) mv_sdf_eval_col0_pp<MT, NTW, NW, CARRY>(net, act, pe, pts, out, tid);
        else mv_sdf_eval_col0<MT, NTW, NW, CARRY, 3, 3>(net, act, pe, pts, out, tid);
        if (tid < ROWS) pts[3 * tid] += 1e-3f * out[tid];            // the next round depends on this one
        __syncthreads();
    }
    if (tid < ROWS) y[blockIdx.x * ROWS + tid] = out[tid];
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20, wgs = argc > 2 ? atoi(argv[2]) : 256;
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
        const size_t el = 3 * mv_packed_bf16_elems(N[l], K[l], 0);                 // three weight terms behind each other per k-block
        std::vector<uint16_t> h(el);
        for (size_t i = 0; i < el; ++i) h[i] = mv_f2bf(((rand() & 0xffff) / 65536.0f - 0.5f) * ((i / 512) % 3 == 0 ? 0.12f : ((i / 512) % 3 == 1 ? 4e-4f : 2e-6f)));
        std::vector<float> b(mv_ceil16(N[l]), 0.01f);
        void *dw, *db;
        (void)hipMalloc(&dw, el * 2); (void)hipMemcpy(dw, h.data(), el * 2, hipMemcpyHostToDevice);
        (void)hipMalloc(&db, b.size() * 4); (void)hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        L.wp = (const uint4*)dw; L.bias = (const float*)db;
    }
    net.S = 3 * ((maxk + 8) / 2);
    std::vector<float> hx(256 * 64 * 3);
    for (auto& v : hx) v = (rand() & 0xffff) / 65536.0f - 0.5f;
    float *x, *y;
    (void)hipMalloc(&x, hx.size() * 4); (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&y, (size_t)wgs * 64 * 4);
    auto run = [&](auto kern, int MT, const char* name, int nthreads = 512, int tiles = 1) {
        const int rows = 16 * MT;
        const size_t lds = ((size_t)tiles * rows * net.S + ((rows * d0 + 3) & ~3) + rows * 4 + rows) * 4;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        unsigned long long z[8] = {0};
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ph), z, sizeof z);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(nthreads), lds, 0, net, x, rounds, y);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long ph[8];
        (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof ph);
        const double per = 1e3 * ms / rounds / ((wgs + 255) / 256);
        printf("%s wgs=%d: %.1f us per round and workgroup slot (%.1f us per 16 rows); wave 0 of workgroup 0, us per round: PE %.2f | descriptors + bias %.2f | wait-in %.2f | ring %.2f | last-layer ring %.2f | wait-readers %.2f | epilogue %.2f | end %.2f  (sum %.1f; lds %zu B)\n",
               name, wgs, per, per / MT, ph[0] * 0.01 / rounds, ph[7] * 0.01 / rounds, ph[1] * 0.01 / rounds, ph[6] * 0.01 / rounds, ph[2] * 0.01 / rounds, ph[3] * 0.01 / rounds, ph[4] * 0.01 / rounds,
               ph[5] * 0.01 / rounds, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5] + ph[6] + ph[7]) * 0.01 / rounds, lds);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
    };
    run(k_rounds<1, 2, true, 8, true>, 1, "carried, ping-pong tiles (k_sphere_trace since round 6) MT=1", 512, 2);
    run(k_rounds<2, 2, true, 8, true>, 2, "carried, ping-pong tiles MT=2", 512, 2);
    run(k_rounds<1, 2, true>, 1, "carried, one tile (k_sphere_trace of round 5) MT=1");
    run(k_rounds<1, 1, true, 16>, 1, "carried 16 waves x 1 tile MT=1", 1024);
    run(k_rounds<1, 1, true, 16, true>, 1, "carried 16 waves x 1 tile, ping-pong MT=1", 1024, 2);
    run(k_rounds<2, 1, true, 16>, 2, "carried 16 waves x 1 tile MT=2", 1024);
    run(k_rounds<1, 1, false, 16>, 1, "rolling 16 waves x 1 tile MT=1", 1024);
    run(k_rounds<2, 1, false, 16>, 2, "rolling 16 waves x 1 tile MT=2", 1024);
    run(k_rounds<4, 1, false, 16>, 4, "rolling 16 waves x 1 tile MT=4", 1024);
    run(k_rounds<2, 4, false, 4>, 2, "rolling 4 waves x 4 tiles MT=2", 256);
    run(k_rounds<2, 4, true, 4>, 2, "carried 4 waves x 4 tiles MT=2", 256);
    run(k_rounds<1, 4, false, 4>, 1, "rolling 4 waves x 4 tiles MT=1", 256);
    run(k_rounds<1, 2, false>, 1, "rolling MT=1");
    run(k_rounds<2, 2, true>, 2, "carried MT=2");
    run(k_rounds<2, 2, false>, 2, "rolling MT=2");
    run(k_rounds<4, 2, false>, 4, "rolling MT=4");
    return 0;
}
