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
};

__global__ __launch_bounds__(256) void k_adam_flat(AdamArgs a) {
    __shared__ float coef_s;
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
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        const float g = a.g[i] * a.grad_scale * coef;
        const float m = a.m[i] + (g - a.m[i]) * (1.0f - a.beta1);                          // exp_avg.lerp_(grad, 1 - beta1)
        const float v = a.v[i] * a.beta2 + (1.0f - a.beta2) * g * g;                       // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        a.m[i] = m; a.v[i] = v; a.g[i] = g;
        a.p[i] = a.p[i] - step_size * (m / denom);
    }
}

extern "C" {

size_t mvsdf_adam_ws_floats(void) { return OPT_BLOCKS_MAX; }

int mvsdf_adam_step_scaled(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step, float max_norm,
                           float grad_scale, float* norm_out, float* ws, void* stream) {
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
    a.max_norm = max_norm; a.grad_scale = grad_scale; a.part = ws; a.nparts = blocks; a.norm_out = norm_out;
    hipLaunchKernelGGL(k_adam_flat, dim3(blocks), dim3(256), 0, s, a);
    return mv_check(hipGetLastError(), "mvsdf_adam_step");
}

int mvsdf_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step, float max_norm,
                    float* norm_out, float* ws, void* stream) {
    return mvsdf_adam_step_scaled(p, g, m, v, n, lr, beta1, beta2, eps, step, max_norm, 1.0f, norm_out, ws, stream);
}

}  // extern "C"
