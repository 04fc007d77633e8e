// sample_kernels.hip -- phase-0 depth-surface sampling of IDRNetwork.forward (reference
// code/model/implicit_differentiable_renderer.py:226-247 with utils/my_utils.py:71-95).
//
// Reference: unproject EVERY depth pixel of every view (B x H x W points), normalise, add a jitter copy, keep the points inside the
// eikonal bounding box, then np.random.choice(n, replace=False) + sort -- tens of millions of bytes of elementwise traffic, a boolean
// mask (host sync) and a host-side permutation of millions of indices per step.
// Here: a uniformly random n-subset of the valid pixels is the first n valid elements of a uniformly random permutation of ALL
// pixels, so the kernel walks a keyed bijection of [0, B*H*W) (4-round Feistel network + cycle walking) and unprojects only the
// candidates it visits (~n / valid_fraction of them) -- no full-image pass, no host involvement.  The selected pixel indices are
// sorted (reference: np.sort) and unprojected once more into the output.  Same distribution as the reference (exact numpy RNG
// stream parity is impossible on any device implementation); per-point arithmetic follows the reference op for op.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "capi_util.h"

struct DsurfArgs {
    const float* depths;      // [N][H][W]
    const float* kinv;        // [N][3][3]  inverse intrinsics   (my_utils.py:84)
    const float* einv;        // [N][4][4]  inverse extrinsics   (my_utils.py:93)
    int N, H, W;
    const float* size; const float* center;     // device: [1], [3]
    float bb, jitter_rad;
    unsigned long long seed;
    int n;                    // samples per set
    int half_bits;            // Feistel half width: 2^(2 * half_bits) >= N*H*W
    long long* idx;           // [2][n] selected pixel indices (set 0: on-surface, set 1: jittered)
    long long* counts;        // [2] valid candidates found (n when enough)
    float* pts_on; float* pts_jit;   // [n][3] outputs of the second kernel
};

__device__ __forceinline__ uint32_t ds_hash(uint32_t x) {          // lowbias32 (public-domain integer hash)
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// keyed bijection of [0, 2^(2*hb)): 4-round balanced Feistel network
__device__ __forceinline__ uint32_t ds_feistel(uint32_t v, int hb, uint32_t k0, uint32_t k1) {
    const uint32_t mask = (1u << hb) - 1u;
    uint32_t l = v >> hb, r = v & mask;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const uint32_t f = ds_hash(r ^ (k0 + 0x9e3779b9U * (uint32_t)round) ^ (k1 << (round & 3))) & mask;
        const uint32_t nl = r;
        r = l ^ f; l = nl;
    }
    return (l << hb) | r;
}
__device__ __forceinline__ float ds_uniform(unsigned long long seed, uint32_t pix, int c) {                  // [0, 1)
    const uint32_t h = ds_hash(ds_hash(pix ^ (uint32_t)seed) + 0x632be5abU * (uint32_t)(c + 1) + (uint32_t)(seed >> 32));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// normalised world point of depth pixel `pix` (my_utils.py:71-95, idr.py:235-238); false if depth <= 0
__device__ __forceinline__ bool ds_point(const DsurfArgs& a, uint32_t pix, float (&p)[3]) {
    const int hw = a.H * a.W;
    const int b = pix / hw, rem = pix - b * hw, y = rem / a.W, x = rem - y * a.W;
    const float d = a.depths[pix];
    if (!(d > 0.0f)) return false;
    const float* K = a.kinv + 9 * b;
    const float* E = a.einv + 16 * b;
    const float u = (float)x + 0.5f, v = (float)y + 0.5f;                          // get_pixel_grids
    float ic[3];
    for (int i = 0; i < 3; ++i) ic[i] = K[3 * i] * u + K[3 * i + 1] * v + K[3 * i + 2];
    const float zi = ic[2] + 1e-9f;
    float hom[4];
    for (int i = 0; i < 3; ++i) hom[i] = ic[i] / zi * d;                           // idx_img2cam
    hom[3] = 1.0f;
    float wv[4];
    for (int i = 0; i < 4; ++i) wv[i] = E[4 * i] * hom[0] + E[4 * i + 1] * hom[1] + E[4 * i + 2] * hom[2] + E[4 * i + 3] * hom[3];
    const float ww = wv[3] + 1e-9f;                                                 // idx_cam2world
    const float s = a.size[0];
    for (int i = 0; i < 3; ++i) p[i] = (wv[i] / ww - a.center[i]) / s * 2.0f;       // idr.py:238
    return true;
}
__device__ __forceinline__ void ds_jitter(const DsurfArgs& a, uint32_t pix, float (&p)[3]) {
    for (int c = 0; c < 3; ++c) p[c] = p[c] + ds_uniform(a.seed, pix, c) * a.jitter_rad * 2.0f - a.jitter_rad;   // idr.py:239
}
__device__ __forceinline__ bool ds_inbound(const DsurfArgs& a, const float (&p)[3]) {
    return fabsf(p[0]) < a.bb && fabsf(p[1]) < a.bb && fabsf(p[2]) < a.bb;          // idr.py:242
}

// blockIdx.x = set (0 on-surface, 1 jittered).  1024 threads walk the permutation 1024 candidates per round.
__global__ __launch_bounds__(1024) void k_dsurf_select(DsurfArgs a) {
    __shared__ int wcnt[16];
    __shared__ int base_s;
    const int set = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t total = (uint32_t)a.N * a.H * a.W;
    const uint32_t k0 = ds_hash((uint32_t)a.seed ^ (set ? 0xa511e9b3U : 0x1f83d9abU)), k1 = ds_hash((uint32_t)(a.seed >> 32) + 0x5be0cd19U * (set + 1));
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (uint32_t r0 = 0; r0 < total; r0 += 1024) {
        const uint32_t k = r0 + tid;
        bool ok = false;
        uint32_t pix = 0;
        if (k < total) {
            pix = k;
            do { pix = ds_feistel(pix, a.half_bits, k0, k1); } while (pix >= total);   // cycle walking: a bijection of [0, total)
            float p[3];
            if (ds_point(a, pix, p)) {
                if (set) ds_jitter(a, pix, p);
                ok = ds_inbound(a, p);
            }
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) wcnt[w] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int i = 0; i < w; ++i) off += wcnt[i];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (ok && pos < a.n) a.idx[(size_t)set * a.n + pos] = (long long)pix;
        __syncthreads();
        if (tid == 0) { int s = 0; for (int i = 0; i < 16; ++i) s += wcnt[i]; base_s += s; }
        __syncthreads();
        if (base_s >= a.n) break;
    }
    if (tid == 0) a.counts[set] = base_s < a.n ? base_s : a.n;
}

__global__ void k_dsurf_points(DsurfArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * a.n) return;
    const int set = i / a.n, j = i - set * a.n;
    float p[3] = {0.f, 0.f, 0.f};
    const long long pix = a.idx[i];
    if (j < (int)a.counts[set] && pix >= 0 && pix < (long long)a.N * a.H * a.W) {
        ds_point(a, (uint32_t)pix, p);
        if (set) ds_jitter(a, (uint32_t)pix, p);
    }
    float* o = (set ? a.pts_jit : a.pts_on) + 3 * (size_t)j;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

static int fill_dsurf(DsurfArgs& a, const float* depths, const float* kinv, const float* einv, int N, int H, int W, const float* size,
                      const float* center, float bb, float jitter_rad, unsigned long long seed, int n, long long* idx, long long* counts) {
    if (!depths || !kinv || !einv || !size || !center || !idx || !counts || N <= 0 || H <= 0 || W <= 0 || n <= 0) return -1;
    const long long total = (long long)N * H * W;
    if (total >= (1ll << 30)) return -1;
    memset(&a, 0, sizeof(a));
    a.depths = depths; a.kinv = kinv; a.einv = einv; a.N = N; a.H = H; a.W = W; a.size = size; a.center = center;
    a.bb = bb; a.jitter_rad = jitter_rad; a.seed = seed; a.n = n; a.idx = idx; a.counts = counts;
    int hb = 1;
    while ((1ll << (2 * hb)) < total) ++hb;
    a.half_bits = hb;
    return 0;
}

extern "C" {

int mvsdf_dsurf_select(const float* depths, const float* kinv, const float* einv, int N, int H, int W, const float* size, const float* center,
                       float bb, float jitter_rad, unsigned long long seed, int n, long long* idx, long long* counts, void* stream) {
    DsurfArgs a;
    if (fill_dsurf(a, depths, kinv, einv, N, H, W, size, center, bb, jitter_rad, seed, n, idx, counts)) return mv_fail(-1, "mvsdf_dsurf_select: bad arguments");
    hipLaunchKernelGGL(k_dsurf_select, dim3(2), dim3(1024), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_dsurf_select");
}

int mvsdf_dsurf_points(const float* depths, const float* kinv, const float* einv, int N, int H, int W, const float* size, const float* center,
                       float bb, float jitter_rad, unsigned long long seed, int n, const long long* idx_sorted, const long long* counts,
                       float* pts_on, float* pts_jit, void* stream) {
    DsurfArgs a;
    if (fill_dsurf(a, depths, kinv, einv, N, H, W, size, center, bb, jitter_rad, seed, n, (long long*)idx_sorted, (long long*)counts) || !pts_on || !pts_jit)
        return mv_fail(-1, "mvsdf_dsurf_points: bad arguments");
    a.pts_on = pts_on; a.pts_jit = pts_jit;
    hipLaunchKernelGGL(k_dsurf_points, dim3((2 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check(hipGetLastError(), "mvsdf_dsurf_points");
}

}  // extern "C"
