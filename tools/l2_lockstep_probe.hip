// L2 -> CU delivery when EVERY workgroup streams the same addresses at the same time (the access pattern of a persistent kernel whose
// workgroups walk one shared weight set in lockstep: denoiser_persist_lp.hip) against waves spread over the region (tools/l2_probe.hip).
// 256 workgroups (one per CU) x 8 waves; wave w of every workgroup reads slice w of each 16-KB k-group (lane-contiguous 1-KB fragments,
// two per wave), k-groups in order; region = 1 MB (one layer's 16-bit weights), repeated.
//   mode 0: lockstep, layout [k-group][16 fragments] (16 KB per k-group: the kernel's layout)
//   mode 1: workgroup i starts at k-group (i * 37) % NG (same layout, de-phased workgroups)
//   mode 2: lockstep, layout [fragment][k-group] with the 16 fragment streams 68 KB apart (one stream per wave-half)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/l2_lockstep_probe tools/l2_lockstep_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ buf, int NG, int iters, int mode, long stream_stride_vec, unsigned* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    int g = mode == 1 ? (int)((blockIdx.x * 37) % NG) : 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH][2];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int gg = (g + d) % NG;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int frag = 2 * w + f;
                const long idx = mode == 2 ? (long)frag * stream_stride_vec + (long)gg * 64 + lane : ((long)gg * 16 + frag) * 64 + lane;
                v[d][f] = buf[idx];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc ^= v[d][0]; acc ^= v[d][1]; }
        g = (g + DEPTH) % NG;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    const int NG = 64;                       // 64 k-groups x 16 KB = 1 MB
    const size_t bytes = 64u << 20;
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {32, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (long pad_kb : {0L, 4L}) {
                if (mode != 2 && pad_kb) continue;
                const long stride_vec = ((long)NG * 1024 + pad_kb * 1024) / 16;
                const int iters = 4000;
                hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, buf, NG, 100, mode, stride_vec, sink);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, buf, NG, iters, mode, stride_vec, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double moved = (double)blocks * 8 * iters * 3 * 2 * 1024.0;
                const double tbs = moved / (ms * 1e-3) / 1e12;
                printf("blocks %3d mode %d pad %ld KB: %6.2f TB/s = %5.1f B/clk per CU (2.4 GHz), %.1f us per MB per workgroup\n", blocks, mode, pad_kb, tbs,
                       tbs * 1e12 / blocks / 2.4e9, ms * 1e3 / (iters * 3.0 / NG));
            }
        }
    }
    return 0;
}
