// Input of a denoiser evaluation in ONE launch (round 2): the sampler's c_in scaling and [B, T, 80] -> [B, 80, T] transpose
// (karras_diffusion.py:405, tts_net.py:31: mel_prep_kernel), Denoiser.input_projection = relu(Conv1d(80 -> 256, k = 1)) (model/
// modules.py:575-577, 624: a launch of the generic conv kernel) and the zeroing of the persistent kernel's halo granules (a
// hipMemsetAsync) — three dependent launches of 8 + 20 + 4 us with ~10 us of boundary latency each, four times per T = 4 step.
//
// Workgroup = 64 frames of one utterance, 4 waves.  The 64 x 80 input tile is ONE contiguous 20-KB block of the time-major
// tensor: read as 1280 float4 (coalesced), scaled and transposed into LDS [80][65]; wave w owns output rows 64 w .. 64 w + 63
// (two 32-row m-tiles) x both 32-frame n-tiles: 10 k-groups x 4 k-steps x 4 = 160 v_mfma_f32_32x32x2_f32, the weights (80 KB as
// MFMA A fragments) all requested up front.  Same products in the same k order, same epilogue (acc + bias, ReLU) as the
// generic kernel on mel_prep's output => the same bits (tests/test_gpu_parity.py::test_fused_input_projection_bitwise).
#include <hip/hip_runtime.h>
#include "inproj.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int M = 80;            // mel channels (K of the contraction)
constexpr int C = 256;           // residual channels
constexpr int FN = 64;           // frames per workgroup
constexpr int X_LD = FN + 1;
constexpr int KG = M / 8;        // 10 k-groups

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(256) void inproj_kernel(const InProjArgs a) {
    __shared__ float xs[M * X_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * FN;
    const int T = a.T;

    // weights of this wave's two m-tiles: 20 fragments, all in flight before anything waits
    f32x4 A[KG][2];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) A[g][i] = *reinterpret_cast<const f32x4*>(a.wf + (((long)g * (C / 32) + 2 * w + i) * 64 + lane) * 4);

    {   // stage: the tile is the contiguous block x[b][t0 .. t0 + 63][0 .. 79]
        const float* xb = a.x + ((long)b * T + t0) * M;
        const float sc = a.scale_b ? a.scale_b[b] : a.scale;
        const int nrow = min(FN, T - t0);
        f32x4 v[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e = 4 * (tid + 256 * j);
            const int row = e / M;
            v[j] = row < nrow ? *reinterpret_cast<const f32x4*>(xb + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e = 4 * (tid + 256 * j);
            const int row = e / M, k = e - row * M;
#pragma unroll
            for (int i = 0; i < 4; ++i) xs[(k + i) * X_LD + row] = sc * v[j][i];      // mel_prep_kernel: s * x
        }
    }
    // the persistent kernel's halo granules must be stale (0) when it starts: every workgroup clears its share
    if (a.zero && a.zero_f4 > 0) {
        f32x4* z = reinterpret_cast<f32x4*>(a.zero);
        const long nwg = (long)gridDim.x * gridDim.y, wg = (long)blockIdx.y * gridDim.x + blockIdx.x;
        for (long i = wg * 256 + tid; i < a.zero_f4; i += nwg * 256) z[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* bl = xs + khalf * X_LD + l31;
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float bv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = bl[(8 * g + 2 * kk) * X_LD + j * 32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g][i][kk], bv[j], acc[i][j], 0, 0, 0);
        }

    float* hb = a.h + (long)b * C * T;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.bias[(2 * w + i) * 32 + acc_row(r, lane)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + j * 32 + l31;
            if (t < T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] + bi[r];
                    hb[(long)((2 * w + i) * 32 + acc_row(r, lane)) * T + t] = v > 0.f ? v : 0.f;
                }
            }
        }
    }
}

}  // namespace

// 0 = launched, -2 = shape not covered (the caller runs mel_prep + the generic conv + its own memset), -3 = HIP error
extern "C" int cmtts_launch_inproj(const InProjArgs* ap, void* stream_) {
    const InProjArgs& a = *ap;
    if (a.B <= 0 || a.T <= 0) return 0;
    if (a.M != M || a.C != C || !a.wf || ((uintptr_t)a.x & 15) || ((uintptr_t)a.zero & 15)) return -2;
    hipLaunchKernelGGL(inproj_kernel, dim3((a.T + FN - 1) / FN, a.B), dim3(256), 0, (hipStream_t)stream_, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
