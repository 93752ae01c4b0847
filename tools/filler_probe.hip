// What does a non-MFMA instruction cost next to v_mfma_f32_32x32x2_f32 on gfx950?  One or two waves per SIMD, 256 workgroups; every loop iteration
// issues 8 independent MFMAs (8 accumulators) and F fillers of one kind spread between them:
//   kind 0 = v_add_f32 (VALU), 1 = ds_read_b32, 2 = global_load_dwordx4 (L2-resident 64 KB region), 3 = v_exp_f32, 4 = s_nop 0 (issue slot only)
// Prints ticks per MFMA (wall time x 2.4 GHz / MFMAs per SIMD): 64 = the pipe's floor.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/filler_probe tools/filler_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int F, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(const float* g, float* out, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i * 1e-9f;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)i;
    const f32x4* gp = reinterpret_cast<const f32x4*>(g) + threadIdx.x;
    f32x4 gl[4] = {};
    float dl[4] = {};
    for (int it = 0; it < iters; ++it) {
        const int ldsa = (lane * 4 + (it & 15) * 256) & 16383;
        const f32x4* gpa = gp + (it & 15) * 256;
        asm volatile("" : "+v"(gpa));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            constexpr int per = (F + 7) / 8;       // fillers behind this MFMA
#pragma unroll
            for (int q = 0; q < per; ++q) {
                const int idx = i * per + q;
                if (idx < F) {
                    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[idx & 15]) : "v"(b));
                    if (KIND == 1) asm volatile("ds_read_b32 %0, %1" : "+v"(dl[idx & 3]) : "v"(ldsa));
                    if (KIND == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(gl[idx & 3]) : "v"(gpa));
                    if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[idx & 15]));
                    if (KIND == 4) asm volatile("s_nop 0");
                }
            }
        }
        if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += dl[i] + gl[i][0] + gl[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int F, int WAVES>
static void run(const float* g, float* out, const char* name) {
    const int iters = 4000;
    printf("%s F=%d W=%d ...\n", name, F, WAVES); fflush(stdout);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, F, WAVES><<<256, 64 * WAVES>>>(g, out, 100);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<KIND, F, WAVES><<<256, 64 * WAVES>>>(g, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mfma_per_simd = (double)iters * 8 * (WAVES / 4);
    printf("%-22s F=%2d waves/SIMD %d: %.3f ms  %.1f ticks(2.4GHz)/MFMA  (%.1f extra per filler)\n", name, F, WAVES / 4, best,
           best * 1e-3 * 2.4e9 / mfma_per_simd, F ? (best * 1e-3 * 2.4e9 / mfma_per_simd - 64.0) * 8.0 / F : 0.0);
}

template <int KIND, int WAVES>
static void sweep(const float* g, float* out, const char* name) {
    run<KIND, 0, WAVES>(g, out, name);
    run<KIND, 2, WAVES>(g, out, name);
    run<KIND, 4, WAVES>(g, out, name);
    run<KIND, 8, WAVES>(g, out, name);
    run<KIND, 16, WAVES>(g, out, name);
    run<KIND, 32, WAVES>(g, out, name);
}

int main() {
    float *g, *out;
    if (hipMalloc(&g, 1 << 20) != hipSuccess || hipMemset(g, 0, 1 << 20) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("g=%p out=%p\n", (void*)g, (void*)out); fflush(stdout);
    sweep<0, 4>(g, out, "v_add_f32"); sweep<0, 8>(g, out, "v_add_f32");
    sweep<1, 4>(g, out, "ds_read_b32"); sweep<1, 8>(g, out, "ds_read_b32");
    sweep<2, 4>(g, out, "global_load_dwordx4"); sweep<2, 8>(g, out, "global_load_dwordx4");
    sweep<3, 4>(g, out, "v_exp_f32");
    sweep<4, 4>(g, out, "s_nop 0");
    return 0;
}
