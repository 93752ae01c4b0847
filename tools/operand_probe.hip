// How many operand bytes per MFMA can a CU take?  v_mfma_f32_32x32x16_bf16 loops with the A fragments streamed from an L2-resident region
// (global_load_dwordx4, 4 k-groups in flight) and the B fragments read from LDS (ds_read_b128), MT x NT accumulator tiles per wave, WPC waves
// per CU, 256 workgroups.  Prints MFMA-pipe utilisation (of 32 cycles per MFMA per SIMD at the measured time) and operand bytes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/operand_probe tools/operand_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MT, int NT, int SRC>      // SRC: 3 = A global + B LDS, 1 = A global only (B constant), 2 = B LDS only (A constant), 0 = neither
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ wbuf, int ngroups, int iters, float* sink) {
    extern __shared__ u32x4 lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int RING = 4;
    u32x4 A[RING][MT], B[2][NT];
    const u32x4* wp = wbuf + ((long)w * MT) * 64 + lane;
    const long gstride = (long)8 * MT * 64;          // one k-group: all waves' fragments
    for (int s = 0; s < RING - 1; ++s) for (int i = 0; i < MT; ++i) A[s][i] = wp[(long)(s % ngroups) * gstride + i * 64];
    for (int j = 0; j < NT; ++j) B[0][j] = lds[(j * 64 + lane) & 4095];
    int g = RING - 1;
    for (int it = 0; it < iters; it += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            if (SRC & 1) { for (int i = 0; i < MT; ++i) A[(s + RING - 1) % RING][i] = wp[(long)g * gstride + i * 64]; }
            if (++g == ngroups) g = 0;
            if (SRC & 2) { for (int j = 0; j < NT; ++j) B[(s + 1) & 1][j] = lds[((it + s) * 128 + j * 64 + lane) & 4095]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[(SRC & 1) ? s : 0][i]),
                                                                       __builtin_bit_cast(bf16x8, B[(SRC & 2) ? (s & 1) : 0][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0.f;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) t += acc[i][j][0];
    if (t == 12345.f) sink[0] = t;
}

template <int MT, int NT, int SRC>
void run(const u32x4* buf, float* sink, int waves) {
    const int iters = 4096, ngroups = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MT, NT, SRC>), dim3(256), dim3(64 * waves), 65536, 0, buf, ngroups, 64, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MT, NT, SRC>), dim3(256), dim3(64 * waves), 65536, 0, buf, ngroups, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * MT * NT * waves / 4.0;
    const double cyc = ms * 1e-3 * 2.4e9;
    const double bytes = (double)iters * waves * 1024.0 * (((SRC & 1) ? MT : 0) + ((SRC & 2) ? NT : 0));
    printf("waves/CU %d  tiles %dx%d  A %s B %s: %7.3f ms  MFMA pipe %5.1f %% (at 2.4 GHz)  operands %5.1f B/clk per CU  (%4.0f B per MFMA)\n", waves, MT, NT,
           (SRC & 1) ? "L2 " : "reg", (SRC & 2) ? "LDS" : "reg", ms, 100.0 * mfma_per_simd * 32.0 / cyc, bytes / cyc,
           1024.0 * (((SRC & 1) ? MT : 0) + ((SRC & 2) ? NT : 0)) / (MT * NT));
}

int main() {
    u32x4* buf; float* sink;
    hipMalloc(&buf, 64u << 20); hipMalloc(&sink, 4); hipMemset(buf, 0x3f, 64u << 20);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<2, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int waves : {8, 4}) {
        run<2, 2, 0>(buf, sink, waves); run<2, 2, 1>(buf, sink, waves); run<2, 2, 2>(buf, sink, waves); run<2, 2, 3>(buf, sink, waves);
        run<2, 4, 3>(buf, sink, waves); run<4, 2, 3>(buf, sink, waves); run<4, 4, 3>(buf, sink, waves); run<4, 4, 1>(buf, sink, waves); run<4, 4, 2>(buf, sink, waves);
    }
    return 0;
}
