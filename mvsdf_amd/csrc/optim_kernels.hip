// Optimiser tail of one training step on FLAT buffers (reference code/training/idr_train.py:289-302):
//   all_norm = ||grad||_2 ; clip_grad_norm_(params, grad_cap) ; Adam.step()
// as two launches over the flat parameter / gradient / moment buffers (803 k floats at W=256: HBM-bound, ~20 bytes per parameter)
// instead of torch's per-parameter-list multi-tensor passes.  Deterministic: block partial sums are added in a fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "capi_util.h"
#include "../../include/mvsdf_hip.h"

#define OPT_BLOCKS_MAX 1024

__global__ __launch_bounds__(256) void k_sqnorm_partials(const float* __restrict__ g, size_t n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s = fmaf(g[i], g[i], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
    float* p; float* g; float* m; float* v; size_t n;
    float lr, beta1, beta2, eps, bc1, bc2_sqrt;      // bias corrections 1 - beta1^t, sqrt(1 - beta2^t) (host doubles, rounded once)
    float max_norm;                                  // <= 0: no clipping
    float grad_scale;                                // applied to every gradient first (1 / world size after a SUM all-reduce)
    const float* part; int nparts;
    float* norm_out;                                 // [2]: total norm, clip coefficient
    int zero_grad;                                   // 1: leave ZEROS in g instead of the scaled / clipped gradient (the next step's zero_grad() then has nothing to do)
};

__device__ __forceinline__ void mv_adam_one(const AdamArgs& a, float coef, float step_size, float gi, float mi, float vi, float pi, float& g, float& m,
                                            float& v, float& p) {
    g = gi * a.grad_scale * coef;
    m = mi + (g - mi) * (1.0f - a.beta1);                                                  // exp_avg.lerp_(grad, 1 - beta1)
    v = vi * a.beta2 + (1.0f - a.beta2) * g * g;                                           // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = pi - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam_flat(AdamArgs a) {
    __shared__ float coef_s;
    // this thread's first four elements are requested BEFORE the norm is re-derived (their loads do not depend on it): the kernel was a chain of
    // two load round trips (partials, then the parameters); 16-byte accesses when the four buffers allow it
    const size_t n4 = a.n >> 2;
    const bool vec = ((((size_t)a.p | (size_t)a.g | (size_t)a.m | (size_t)a.v) & 15) == 0);
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x, stride4 = (size_t)gridDim.x * 256;
    float4 g4 = {0.f, 0.f, 0.f, 0.f}, m4 = g4, v4 = g4, p4 = g4;
    const bool have = vec && i4 < n4;
    {
        const size_t j = have ? i4 : 0;                          // clamped: no branch around the loads
        if (vec) { g4 = ((const float4*)a.g)[j]; m4 = ((const float4*)a.m)[j]; v4 = ((const float4*)a.v)[j]; p4 = ((const float4*)a.p)[j]; }
    }
    if (threadIdx.x < 64) {                          // every block re-derives the norm from the partials, same order everywhere
        float s = 0.0f;
        for (int i = threadIdx.x; i < a.nparts; i += 64) s += a.part[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (threadIdx.x == 0) {
            const float total = sqrtf(s) * a.grad_scale;
            float coef = 1.0f;
            if (a.max_norm > 0.0f) coef = fminf(a.max_norm / (total + 1e-6f), 1.0f);      // torch.nn.utils.clip_grad_norm_
            coef_s = coef;
            if (blockIdx.x == 0 && a.norm_out) { a.norm_out[0] = total; a.norm_out[1] = coef; }
        }
    }
    __syncthreads();
    const float coef = coef_s;
    const float step_size = a.lr / a.bc1;
    if (vec) {
        for (size_t i = i4; i < n4; i += stride4) {
            if (i != i4) { g4 = ((const float4*)a.g)[i]; m4 = ((const float4*)a.m)[i]; v4 = ((const float4*)a.v)[i]; p4 = ((const float4*)a.p)[i]; }
            float4 go, mo, vo, po;
            mv_adam_one(a, coef, step_size, g4.x, m4.x, v4.x, p4.x, go.x, mo.x, vo.x, po.x);
            mv_adam_one(a, coef, step_size, g4.y, m4.y, v4.y, p4.y, go.y, mo.y, vo.y, po.y);
            mv_adam_one(a, coef, step_size, g4.z, m4.z, v4.z, p4.z, go.z, mo.z, vo.z, po.z);
            mv_adam_one(a, coef, step_size, g4.w, m4.w, v4.w, p4.w, go.w, mo.w, vo.w, po.w);
            if (a.zero_grad) go = float4{0.f, 0.f, 0.f, 0.f};
            ((float4*)a.m)[i] = mo; ((float4*)a.v)[i] = vo; ((float4*)a.g)[i] = go; ((float4*)a.p)[i] = po;
        }
    }
    for (size_t i = (vec ? (n4 << 2) : 0) + (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {   // tail / unaligned buffers
        float g, m, v, p;
        mv_adam_one(a, coef, step_size, a.g[i], a.m[i], a.v[i], a.p[i], g, m, v, p);
        a.m[i] = m; a.v[i] = v; a.g[i] = a.zero_grad ? 0.0f : g; a.p[i] = p;
    }
}

extern "C" {

size_t mvsdf_adam_ws_floats(void) { return OPT_BLOCKS_MAX; }

int mvsdf_adam_step_fused(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step, float max_norm,
                          float grad_scale, int zero_grad, float* norm_out, float* ws, void* stream);

int mvsdf_adam_step_scaled(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step, float max_norm,
                           float grad_scale, float* norm_out, float* ws, void* stream) {
    return mvsdf_adam_step_fused(p, g, m, v, n, lr, beta1, beta2, eps, step, max_norm, grad_scale, 0, norm_out, ws, stream);
}

int mvsdf_adam_step_fused(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step, float max_norm,
                          float grad_scale, int zero_grad, float* norm_out, float* ws, void* stream) {
    if (!p || !g || !m || !v || !ws || n == 0 || step < 1 || !(grad_scale > 0.0f)) return mv_fail(-1, "mvsdf_adam_step: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)((n + 2047) / 2048);
    if (blocks > OPT_BLOCKS_MAX) blocks = OPT_BLOCKS_MAX;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sqnorm_partials, dim3(blocks), dim3(256), 0, s, g, n, ws);
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.max_norm = max_norm; a.grad_scale = grad_scale; a.part = ws; a.nparts = blocks; a.norm_out = norm_out; a.zero_grad = zero_grad ? 1 : 0;
    hipLaunchKernelGGL(k_adam_flat, dim3(blocks), dim3(256), 0, s, a);
    return mv_check(hipGetLastError(), "mvsdf_adam_step");
}

int mvsdf_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step, float max_norm,
                    float* norm_out, float* ws, void* stream) {
    return mvsdf_adam_step_scaled(p, g, m, v, n, lr, beta1, beta2, eps, step, max_norm, 1.0f, norm_out, ws, stream);
}

}  // extern "C"
