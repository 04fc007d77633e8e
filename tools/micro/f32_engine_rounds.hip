// Where does one dependent evaluation ("round") of the fp32 tracing engine spend its time?  (dev probe, gfx950; the fp32 twin of
// bf16_engine_rounds.hip)  One 512-thread workgroup per CU runs `rounds` dependent evaluations of the fused 9-layer MLP (tile_engine.h, the
// shipped 8x256 shape) on its own 16*MT rows -- the situation of k_sphere_trace -- and wave 0 of workgroup 0 accumulates the 100 MHz wall clock
// between the engine's phase marks: PE | descriptors + bias loads | wait for inputs (barrier) | GEMM ring | wait for readers (barrier) | epilogue.
// The matrix-instruction floor at 16 rows is 8 layers x 3.4 us + the one-column last layer.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../mvsdf_amd/csrc f32_engine_rounds.hip -o f32_engine_rounds && ./f32_engine_rounds [rounds] [wgs]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ unsigned long long g_ph[16 * 8];                       // [wave][phase]
#define MV_PH_DECL unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tp_ = wall_clock64();
#define MV_PH(p) { const unsigned long long t_ = wall_clock64(); ph_[p] += t_ - tp_; tp_ = t_; }
#define MV_PH_END if (blockIdx.x == 0 && (tid & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_ph[(tid >> 6) * 8 + i_], ph_[i_]); }
#include "tile_engine.h"

template <int MT, int NTW, int NW, bool XR>
__global__ __launch_bounds__(64 * NW) void k_rounds(MvNet net, const float* __restrict__ x, int rounds, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x, d0 = 3 + 6 * net.multires;
    float* act = smem;
    float* pe = act + ROWS * net.S;
    float* pts = pe + ((ROWS * d0 + 3) & ~3);
    float* out = pts + ROWS * 4;
    for (int i = tid; i < ROWS * 3; i += 64 * NW) pts[i] = x[(blockIdx.x % 256) * ROWS * 3 + i];
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        mv_sdf_eval_col0<MT, NTW, NW, XR>(net, act, pe, pts, out, tid);
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
    MvNet net = {};
    net.n_layers = nl; net.skip_mask = 1u << 4; net.multires = 6;
    int maxk = 0;
    srand(1);
    for (int l = 0; l < nl; ++l) {
        MvLayer& L = net.L[l];
        L.K = K[l]; L.N = N[l]; L.KB = mv_kpad(K[l]) / 16; L.NT = mv_ceil16(N[l]) / 16;
        maxk = mv_kpad(K[l]) > maxk ? mv_kpad(K[l]) : maxk;
        const size_t el = mv_packed_floats(N[l], K[l]);
        std::vector<float> h(el);
        for (size_t i = 0; i < el; ++i) h[i] = ((rand() & 0xffff) / 65536.0f - 0.5f) * 0.12f;
        std::vector<float> b(mv_ceil16(N[l]), 0.01f);
        void *dw, *db;
        hipMalloc(&dw, el * 4); hipMemcpy(dw, h.data(), el * 4, hipMemcpyHostToDevice);
        hipMalloc(&db, b.size() * 4); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        L.wp = (const float4*)dw; L.bias = (const float*)db;
    }
    net.S = ((maxk + 63) & ~63) + 8;
    std::vector<float> hx(256 * 64 * 3);
    for (auto& v : hx) v = (rand() & 0xffff) / 65536.0f - 0.5f;
    float *x, *y;
    hipMalloc(&x, hx.size() * 4); hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&y, (size_t)wgs * 64 * 4);
    auto run = [&](auto kern, int MT, const char* name, int nthreads) {
        const int rows = 16 * MT;
        const size_t lds = ((size_t)rows * net.S + ((rows * d0 + 3) & ~3) + rows * 4 + rows) * 4;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        unsigned long long z[128] = {0};
        for (int rep = 0; rep < 2; ++rep) {
            hipMemcpyToSymbol(HIP_SYMBOL(g_ph), z, sizeof z);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(nthreads), lds, 0, net, x, rounds, y);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long pw[128];
        hipMemcpyFromSymbol(pw, HIP_SYMBOL(g_ph), sizeof pw);
        unsigned long long* ph = pw;
        const double per = 1e3 * ms / rounds / ((wgs + 255) / 256), c = 0.01 / rounds;
        printf("%s wgs=%d: %.1f us per round (%.1f us per 16 rows); wave 0 of workgroup 0, us per round: PE %.2f | descriptors + bias %.2f | wait-in %.2f | ring %.2f | wait-readers %.2f | epilogue %.2f | end %.2f  (sum %.1f)\n",
               name, wgs, per, per / MT, ph[0] * c, ph[7] * c, ph[1] * c, ph[6] * c, ph[3] * c, ph[4] * c, ph[5] * c,
               (ph[0] + ph[1] + ph[3] + ph[4] + ph[5] + ph[6] + ph[7]) * c);
        for (int wv = 1; wv < nthreads / 64; ++wv) {
            const unsigned long long* q = pw + 8 * wv;
            printf("    wave %d: PE %.2f | descriptors + bias %.2f | wait-in %.2f | ring %.2f | wait-readers %.2f | epilogue %.2f\n", wv, q[0] * c, q[7] * c, q[1] * c, q[6] * c, q[3] * c, q[4] * c);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
    };
    run(k_rounds<1, 2, 8, false>, 1, "MT=1 8 waves", 512);
    run(k_rounds<1, 2, 8, true>, 1, "MT=1 8 waves, carried ring", 512);
    run(k_rounds<2, 2, 8, false>, 2, "MT=2 8 waves", 512);
    run(k_rounds<1, 4, 4, false>, 1, "MT=1 4 waves", 256);
    run(k_rounds<1, 1, 16, false>, 1, "MT=1 16 waves", 1024);
    return 0;
}
