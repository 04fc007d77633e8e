// How fast can 256 CUs stream the SAME packed weights (1.1 MB of bf16, 9 layers) out of L2, in the tile engine's access pattern?  (dev probe, gfx950)
// One 512-thread workgroup per CU; wave w reads column tiles 2w, 2w+1 of every layer, k-block by k-block (1 KB per wave and load), `rounds`
// times; nothing else happens (the loaded words are xor-ed).  Modes:
//   0  engine order: every wave at k-block kb at the same time (tiles are 8 KB apart: a power of two)
//   1  k-block order rotated by the wave index (wave w starts at k-block w)
//   2  tiles padded by 256 bytes (tile stride 8 KB + 256 B)
//   3  both
//   hipcc --offload-arch=gfx950 -O3 l2_weight_stream.hip -o l2_weight_stream && ./l2_weight_stream [rounds] [wgs]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(512) void k_stream(const uint4* __restrict__ w, int layers, int KB, int tile_stride16, int rounds, unsigned* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r)
        for (int l = 0; l < layers; ++l) {
            const uint4* base = w + (size_t)l * 16 * tile_stride16;
#pragma unroll 1
            for (int k0 = 0; k0 < KB; k0 += 4) {
                uint4 v[4][2];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    int kb = k0 + d;
                    if (MODE & 1) kb = (kb + wv) % KB;
#pragma unroll
                    for (int t = 0; t < 2; ++t) v[d][t] = base[(size_t)(2 * wv + t) * tile_stride16 + kb * 64 + lane];
                }
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int t = 0; t < 2; ++t) { acc.x ^= v[d][t].x; acc.y ^= v[d][t].y; acc.z ^= v[d][t].z; acc.w ^= v[d][t].w; }
            }
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 50, wgs = argc > 2 ? atoi(argv[2]) : 256;
    const int layers = 9, KB = 8;
    uint4* w; unsigned* out;
    const size_t stride_pad = KB * 64 + 16;                         // in uint4: 8 KB + 256 B
    hipMalloc(&w, (size_t)layers * 16 * stride_pad * 16 + 4096);
    hipMemset(w, 1, (size_t)layers * 16 * stride_pad * 16 + 4096);
    hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int stride, const char* name) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, w, layers, KB, stride, rounds, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double bytes = (double)wgs * rounds * layers * 16 * KB * 1024;
        printf("%-44s %.2f us per round (%.0f KB per workgroup), %.1f TB/s over %d workgroups, %.1f B/clk per CU at 2.4 GHz\n", name, 1e3 * ms / rounds,
               layers * 16 * KB * 1.0, bytes / ms / 1e9, wgs, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 * (256.0 / (wgs < 256 ? wgs : 256)));
    };
    run(k_stream<0>, KB * 64, "0 engine order");
    run(k_stream<1>, KB * 64, "1 k-blocks rotated by wave");
    run(k_stream<0>, (int)stride_pad, "2 tiles padded by 256 B");
    run(k_stream<1>, (int)stride_pad, "3 rotated + padded");
    return 0;
}
