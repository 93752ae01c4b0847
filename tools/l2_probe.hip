// L2 -> CU fill rate on gfx950: every wave streams the SAME region of `bytes` bytes (L2-resident when small) with
// global_load_dwordx4, DEPTH loads in flight per lane.  Prints aggregate TB/s and B/clk per CU (at 2.4 GHz) per region size.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/l2_probe tools/l2_probe.hip && tools/bin/l2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ buf, long nvec, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const long per = 64L * DEPTH;                       // u32x4 per wave per step
    const long steps = nvec / per;
    long step = (wave * 7919L) % steps;                 // every wave starts somewhere else in the region
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = buf[step * per + d * 64 + lane];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        if (++step == steps) step = 0;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    const size_t maxb = 1ull << 30;
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, maxb); hipMalloc(&sink, 4);
    hipMemset(buf, 1, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {256u << 10, 1u << 20, 2u << 20, 8u << 20, 64u << 20, 1u << 30};
    for (int wpc : {4, 8, 16}) {           // waves per CU
        for (size_t bytes : sizes) {
            const long nvec = bytes / 16;
            const int iters = 2000;
            const int blocks = 256 * wpc / 4;
            hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, nvec, 50, sink);
            hipEventRecord(e0);
            hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, nvec, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double moved = (double)blocks * 4 * iters * 8 * 1024.0;
            const double tbs = moved / (ms * 1e-3) / 1e12;
            printf("waves/CU %2d  region %8.2f MB: %6.2f TB/s = %5.1f B/clk per CU (2.4 GHz)\n", wpc, bytes / 1048576.0, tbs,
                   tbs * 1e12 / 256 / 2.4e9);
        }
    }
    return 0;
}
