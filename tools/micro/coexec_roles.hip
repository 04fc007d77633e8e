// What keeps a second workgroup on a CU from overlapping with the f32x3 tracing engine?  (dev probe, gfx950, late round 6)
// 512 workgroups of 512 threads held to 128 registers, two per CU (the engine's 56 KB of LDS each; without the register cap the second one does not fit and runs AFTER the first).  Workgroups 0..255 ("A") run `rounds` dependent evaluations of the fused 9-layer MLP
// (tile_engine_bf16s.h, MvNetBs<3, 3>, 8x256, 32 rows: the MT = 2 form of k_ray_samples) and take their own time; workgroups 256..511 ("B") run ONE kind of
// instruction stream until every A is done:
//   0 nothing (A alone)   1 VALU, 8 independent pk_fma chains   2 VALU, one dependent fma chain   3 bf16 matrix instructions only (8 accumulators)
//   4 LDS reads (ds_read_b128)   5 LDS writes   6 global loads of the engine's weight stream   7 the engine itself   8 VALU (8 chains) + LDS writes, the epilogue's mix
// A's time per evaluation under each B says which unit the engine's phases are sensitive to.  The (xcc, se, cu) of every workgroup is recorded: the table is only
// printed for CUs that hold exactly one A and one B.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../mvsdf_amd/csrc coexec_roles.hip -o bin/coexec_roles && bin/coexec_roles [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include "tile_engine_bf16s.h"

typedef float f2_t __attribute__((ext_vector_type(2)));

template <int MT>
__global__ __launch_bounds__(512, 4) void k_roles(MvNetBs<3, 3> net, const float* __restrict__ x, int rounds, int role_b, unsigned* __restrict__ done,
                                              unsigned long long* __restrict__ t_out, unsigned* __restrict__ where, float* __restrict__ y, const uint4* __restrict__ wstream, int wstream_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT;
    const int tid = threadIdx.x, d0 = 3 + 6 * net.multires;
    float* act = smem;
    float* pe = act + ROWS * net.S;
    float* pts = pe + ((ROWS * d0 + 3) & ~3);
    float* out = pts + ROWS * 4;
    if (tid == 0) {
        where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
    const bool is_a = blockIdx.x < 256;
    const int role = is_a ? 7 : role_b;
    if (role == 0) return;
    for (int i = tid; i < ROWS * 3; i += 512) pts[i] = x[(blockIdx.x % 256) * ROWS * 3 + i];
    __syncthreads();
    float r = 0.f;
    if (role == 7) {
        const unsigned long long t0 = wall_clock64();
        int n = 0;
        for (;;) {
            mv_sdf_eval_col0<MT, 2, 8, false, 3, 3>(net, act, pe, pts, out, tid);
            if (tid < ROWS) pts[3 * tid] += 1e-3f * out[tid];
            __syncthreads();
            ++n;
            if (is_a) { if (n >= rounds) break; }
            else if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 256u) break;
        }
        if (is_a && tid == 0) {
            t_out[blockIdx.x] = wall_clock64() - t0;
            __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < ROWS) r = out[tid];
    } else {
        f2_t acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f2_t{x[tid & 255] + i, x[(tid + 7) & 255]};
        const f2_t m = f2_t{0.999f, 1.001f}, c = f2_t{1e-3f, -1e-3f};
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        f32x4_ macc[8];
        for (int i = 0; i < 8; ++i) macc[i] = f32x4_{0, 0, 0, 0};
        mv_bf8 fa, fb;
        for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3c00 + (tid & 63) + i); fb[i] = (short)(0x3b00 + i); }
        uint4* lds4 = (uint4*)smem;
        uint4 v = uint4{(unsigned)tid, 1u, 2u, 3u};
        float s1 = x[tid & 255];
        for (int it = 0;; ++it) {
            if ((it & 31) == 0 && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 256u) break;
            if (role == 1 || role == 8) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(m), "v"(c));
            }
            if (role == 2) {
#pragma unroll
                for (int k = 0; k < 32; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s1) : "v"(0.999f), "v"(1e-3f));
            }
            if (role == 3) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int i = 0; i < 8; ++i) macc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, macc[i], 0, 0, 0);
            }
            if (role == 4) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { uint4 q = lds4[(tid + 64 * k + it) & 1023]; v.x ^= q.x; v.y += q.w; }
            }
            if (role == 5 || role == 8) {
#pragma unroll
                for (int k = 0; k < (role == 8 ? 2 : 8); ++k) lds4[(tid + 512 * (k & 1)) & 1023] = v;
                asm volatile("" ::: "memory");
            }
            if (role == 6) {
#pragma unroll
                for (int k = 0; k < 6; ++k) { uint4 q = wstream[((size_t)(it * 6 + k) * 512 + tid) % (size_t)wstream_n]; v.x ^= q.x; v.y += q.w; }
            }
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + macc[i][0];
        r += s1 + (float)v.x + (float)v.y;
    }
    if (tid < ROWS) y[blockIdx.x * ROWS + tid] = r;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    const int W = 256, d0 = 39, nl = 9, MT = 2;
    int K[9], N[9];
    for (int l = 0; l < nl; ++l) { K[l] = W; N[l] = W; }
    K[0] = d0; N[3] = W - d0; N[8] = 1;
    MvNetBs<3, 3> net = {};
    net.n_layers = nl; net.skip_mask = 1u << 4; net.multires = 6;
    int maxk = 0;
    srand(1);
    const uint4* wstream = nullptr; int wstream_n = 0;
    for (int l = 0; l < nl; ++l) {
        MvLayerBf& L = net.L[l];
        L.K = K[l]; L.N = N[l]; L.nsplit = 0; L.KB = mv_bf_kb(K[l], 0); L.NT = mv_ceil16(N[l]) / 16;
        maxk = L.KB * 32 > maxk ? L.KB * 32 : maxk;
        const size_t el = 3 * mv_packed_bf16_elems(N[l], K[l], 0);
        std::vector<uint16_t> h(el);
        for (size_t i = 0; i < el; ++i) h[i] = mv_f2bf(((rand() & 0xffff) / 65536.0f - 0.5f) * ((i / 512) % 3 == 0 ? 0.12f : ((i / 512) % 3 == 1 ? 4e-4f : 2e-6f)));
        std::vector<float> b(mv_ceil16(N[l]), 0.01f);
        void *dw, *db;
        (void)hipMalloc(&dw, el * 2); (void)hipMemcpy(dw, h.data(), el * 2, hipMemcpyHostToDevice);
        (void)hipMalloc(&db, b.size() * 4); (void)hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        L.wp = (const uint4*)dw; L.bias = (const float*)db;
        if (l == 1) { wstream = (const uint4*)dw; wstream_n = (int)(el * 2 / 16); }
    }
    net.S = 3 * ((maxk + 8) / 2);
    std::vector<float> hx(256 * 64 * 3);
    for (auto& v : hx) v = (rand() & 0xffff) / 65536.0f - 0.5f;
    float *x, *y; unsigned* done; unsigned long long* t_out; unsigned* where;
    (void)hipMalloc(&x, hx.size() * 4); (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&y, 512 * 64 * 4); (void)hipMalloc(&done, 4); (void)hipMalloc(&t_out, 256 * 8); (void)hipMalloc(&where, 1024 * 4);
    const int rows = 16 * MT;
    const size_t lds = ((size_t)rows * net.S + ((rows * d0 + 3) & ~3) + rows * 4 + rows) * 4;
    (void)hipFuncSetAttribute((const void*)k_roles<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const char* names[9] = {"nothing (A alone)", "VALU: 8 independent pk_fma chains", "VALU: one dependent fma chain", "bf16 matrix instructions only", "LDS reads", "LDS writes",
                            "global loads of a layer's weight stream", "the engine itself", "VALU (8 chains) + LDS writes"};
    printf("A = %d dependent evaluations of the f32x3 8x256 MLP on 32 rows (8 waves x 2 column tiles, rolling ring), lds %zu B per workgroup\n", rounds, lds);
    for (int wu = 0; wu < 6; ++wu) {                                   // the GPU's clocks ramp for ~15 ms after idle: warm up
        (void)hipMemset(done, 0, 4);
        hipLaunchKernelGGL(k_roles<MT>, dim3(512), dim3(512), lds, 0, net, x, rounds, 7, done, t_out, where, y, wstream, wstream_n);
    }
    (void)hipDeviceSynchronize();
    for (int role = 0; role < 9; ++role) {
        double best = 0; int pairs = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(done, 0, 4);
            hipLaunchKernelGGL(k_roles<MT>, dim3(512), dim3(512), lds, 0, net, x, rounds, role, done, t_out, where, y, wstream, wstream_n);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> t(256); std::vector<unsigned> wh(1024);
            (void)hipMemcpy(t.data(), t_out, 256 * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(wh.data(), where, 1024 * 4, hipMemcpyDeviceToHost);
            std::map<unsigned, int> cu_a, cu_b;
            auto key = [&](int b) { return ((wh[2 * b + 1] & 0xf) << 16) | (wh[2 * b] & 0xff00); };     // xcc | se, sh, cu
            for (int b = 0; b < 256; ++b) cu_a[key(b)]++;
            for (int b = 256; b < 512; ++b) cu_b[key(b)]++;
            double sum = 0; int n = 0;
            for (int b = 0; b < 256; ++b) if (cu_a[key(b)] == 1 && cu_b[key(b)] == 1) { sum += (double)t[b]; ++n; }
            pairs = n;
            best = n ? sum / n * 0.01 / rounds : 0;
        }
        hipError_t e = hipGetLastError();
        printf("B = %-44s: A takes %6.1f us per evaluation (mean over %3d CUs holding one A and one B)%s\n", names[role], best, pairs, e != hipSuccess ? hipGetErrorString(e) : "");
    }
    return 0;
}
