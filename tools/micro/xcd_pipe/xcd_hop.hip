// Probe (dev): can two workgroups on the SAME XCD hand data to each other through the XCD's L2 without agent-scope fences (no buffer_wbl2 / buffer_inv),
// and what does one hop cost?  Pairs of workgroups (x, x + stride) ping-pong a 4 KiB payload + a flag; the reader polls the flag with cache-bypassing loads
// (sc1) and reads the payload with sc0 sc1 loads; the writer orders payload before flag with s_waitcnt vmcnt(0) only.
//   hipcc --offload-arch=gfx950 -O2 -o xcd_hop xcd_hop.hip && ./xcd_hop [iters] [stride]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ uint4 ld_sc(const uint4* p) { uint4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }

__device__ __forceinline__ unsigned ld_flag(const unsigned* p, int plain) {
    if (!plain) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v;
}
__device__ __forceinline__ void st_flag(unsigned* p, unsigned v, int plain) {
    if (!plain) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__global__ __launch_bounds__(256) void k_hop(uint4* payload, unsigned* flags, int iters, int stride, int plain, unsigned* xcc, unsigned long long* ticks, unsigned* errs) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) xcc[b] = xcc_id();
    const int grp = b / (2 * stride), off = b % (2 * stride);
    const bool first = off < stride;
    const int pair = grp * stride + (off % stride);            // pair index
        uint4* buf = payload + (size_t)pair * 2 * 256;              // two 4 KiB buffers per pair (a->b, b->a)
    unsigned* fl = flags + (size_t)pair * 64;                   // fl[0]: a->b round, fl[32]: b->a round (separate lines)
    unsigned bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (first) {
            buf[tid] = uint4{(unsigned)it, (unsigned)tid, (unsigned)(it * 7 + tid), 0x1234u};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_flag(&fl[0], (unsigned)it, plain);
            // wait for the reply
            if (tid == 0) { int spin = 0; while (ld_flag(&fl[32], plain) < (unsigned)it) { if (++spin > 4000000) { atomicAdd(&errs[1], 1u); break; } } }
            __syncthreads();
            const uint4 v = ld_sc(buf + 256 + tid);
            if (v.x != (unsigned)it || v.y != (unsigned)tid || v.z != (unsigned)(it * 11 + tid)) ++bad;
        } else {
            if (tid == 0) { int spin = 0; while (ld_flag(&fl[0], plain) < (unsigned)it) { if (++spin > 4000000) { atomicAdd(&errs[1], 1u); break; } } }
            __syncthreads();
            const uint4 v = ld_sc(buf + tid);
            if (v.x != (unsigned)it || v.y != (unsigned)tid || v.z != (unsigned)(it * 7 + tid)) ++bad;
            buf[256 + tid] = uint4{(unsigned)it, (unsigned)tid, (unsigned)(it * 11 + tid), 0x4321u};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_flag(&fl[32], (unsigned)it, plain);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (bad) atomicAdd(&errs[0], bad);
    if (tid == 0) ticks[b] = t1 - t0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int stride = argc > 2 ? atoi(argv[2]) : 8;            // partner = block + stride (8: same XCD if workgroups go round-robin over 8 XCDs)
    const int nb = argc > 3 ? atoi(argv[3]) : 256;
    const int plain = argc > 4 ? atoi(argv[4]) : 0;
    uint4* payload; unsigned *flags, *xcc, *errs; unsigned long long* ticks;
    hipMalloc(&payload, (size_t)nb * 256 * 16 * 2); hipMalloc(&flags, (size_t)nb * 64 * 4); hipMalloc(&xcc, nb * 4); hipMalloc(&errs, 8); hipMalloc(&ticks, nb * 8);
    hipMemset(flags, 0, (size_t)nb * 64 * 4); hipMemset(errs, 0, 8); hipMemset(payload, 0, (size_t)nb * 256 * 16 * 2);
    hipLaunchKernelGGL(k_hop, dim3(nb), dim3(256), 0, 0, payload, flags, iters, stride, plain, xcc, ticks, errs);
    hipError_t e = hipDeviceSynchronize();
    unsigned hx[1024], he[2]; unsigned long long ht[1024];
    hipMemcpy(hx, xcc, nb * 4, hipMemcpyDeviceToHost); hipMemcpy(he, errs, 8, hipMemcpyDeviceToHost); hipMemcpy(ht, ticks, nb * 8, hipMemcpyDeviceToHost);
    printf("plain flags %d; sync: %s; stride %d, %d blocks, %d round trips: payload mismatches %u, spin timeouts %u\n", plain, hipGetErrorString(e), stride, nb, iters, he[0], he[1]);
    printf("xcc of blocks 0..15:"); for (int i = 0; i < 16 && i < nb; ++i) printf(" %u", hx[i]); printf("\n");
    int same = 0; for (int i = 0; i + stride < nb; ++i) same += hx[i] == hx[i + stride];
    int rr = 0; for (int i = 0; i < nb; ++i) rr += hx[i] == (unsigned)(i % 8);
    printf("blocks with xcc == blockIdx %% 8: %d of %d; pairs (i, i+stride) on the same xcc: %d\n", rr, nb, same);
    double mx = 0, mn = 1e30; for (int i = 0; i < nb; ++i) { const double us = ht[i] / 100.0; if (us > mx) mx = us; if (us < mn) mn = us; }
    printf("round trip (2 hops of 4 KiB + flag): %.3f .. %.3f us per iteration (100 MHz wall clock)\n", mn / iters, mx / iters);
    return 0;
}
