// Low-latency form of the denoiser residual block for SMALL batches (a few utterances): the same arithmetic as
// resblock_fused.hip, spread over four times as many CUs.
//
// resblock_fused_kernel<32> gives a 32-frame tile to ONE workgroup of 16 waves: all 16 output m-tiles of the gated conv
// (K = 768) and of the output projection (K = 256) run on one CU, ~55 us of matrix pipe per layer whatever the batch
// (83 us measured) — one 150-frame utterance keeps 5 of 256 CUs busy for 20 x 83 us per evaluation.  Here a tile is cut
// along the OUTPUT ROWS into four workgroups of 4 waves (one wave per SIMD, one m-tile per wave):
//   kernel 1 (conv):  stage u = cp + (x + d [+ p]) for the tile (every workgroup stages all 256 input rows, from L2),
//                     gated k=3 conv for its 4 m-tiles, z -> HBM scratch [B][256][T];
//   kernel 2 (out):   stage z, output projection for its 4 m-tiles, x' / skip epilogue.
// Every wave executes exactly the instruction sequence its counterpart in the fused kernel executes (same fragment
// stream, same (16-channel chunk, tap, k) order, same epilogue expressions): BITWISE equal
// (tests/test_gpu_parity.py::test_split_resblock_bitwise).  Two launches of ~20 and ~12 us replace one of 83.
#include <hip/hip_runtime.h>
#include "gate.h"
#include "resblock_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int C = 256;
constexpr int FN = 32;          // frames per tile
constexpr int NWS = 4;          // waves per workgroup
constexpr int MS = 4;           // workgroups per tile (4 m-tiles each: 16 m-tiles of 32 rows)
constexpr int RING = 4;
constexpr int U_LD = FN + 4;

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ void load_a(f32x4& dst, const float* wfrag, int group, int mt, int lane) {
    dst = *reinterpret_cast<const f32x4*>(wfrag + ((long)group * (2 * C / 32) + mt) * 256 + lane * 4);
}
__device__ __forceinline__ void load_b(float (&dst)[4], const float* tile, int krow0, int col, int lane) {
    const float* bs = tile + (krow0 + (lane >> 5)) * U_LD + (lane & 31) + col;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dst[kk] = bs[2 * kk * U_LD];
}
__device__ __forceinline__ void mma_group(f32x16& acc, const f32x4& af, const float (&bv)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bv[kk], acc, 0, 0, 0);
}

__global__ __launch_bounds__(64 * NWS) void resblock_split_conv_kernel(const ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) float u_lds[];     // [C][U_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = blockIdx.z * NWS + w;           // this wave's m-tile = wave index of the fused kernel
    const int b = blockIdx.y, t0 = blockIdx.x * FN, T = a.T;
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* xin = a.x_in + (long)b * C * T;
    const float* cp = a.cp + (long)b * a.cp_bstride;
    const float* dp = a.dp + (long)b * a.vec_stride;

    // ---- stage u[m][j] = cp + (x + dp), frames t0-1+j, j in [0, FN+2), zero outside [0, T): lanes = (row parity,
    // frame), 8 rows in flight per thread; the same expression as the fused kernel's staging
    {
        const int t = t0 + l31, t_c = min(t, T - 1);
        constexpr int ROWS_PER_WAVE = C / NWS;       // 64
#pragma unroll 1
        for (int i = 0; i < ROWS_PER_WAVE / 2; i += 8) {
            float xv[8], cv[8], dq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = w * ROWS_PER_WAVE + (i + q) * 2 + khalf;
                xv[q] = xin[(unsigned)(m * T + t_c)];
                cv[q] = cp[(unsigned)(m * T + t_c)];
                dq[q] = dp[m];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = w * ROWS_PER_WAVE + (i + q) * 2 + khalf;
                const float uv = cv[q] + (xv[q] + dq[q]);
                u_lds[m * U_LD + 1 + l31] = t < T ? uv : 0.f;
            }
        }
        for (int i = tid; i < 2 * C; i += 64 * NWS) {   // halo columns: (side, row)
            const int m = i & (C - 1);
            const bool right = i >= C;
            const int th = right ? t0 + FN : t0 - 1;
            const int thc = min(max(th, 0), T - 1);
            const float uh = cp[(unsigned)(m * T + thc)] + (xin[(unsigned)(m * T + thc)] + dp[m]);
            u_lds[m * U_LD + (right ? FN + 1 : 0)] = (th >= 0 && th < T) ? uh : 0.f;
        }
    }
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        constexpr int NG = (C / 8) * 3;                       // 96 k-groups: (16-chunk, tap, 8-half)
        auto kgrp = [&](int it, int& g8, int& tap) {
            it = min(it, NG - 1);
            const int q = it / 6, rr = it - q * 6;
            tap = rr >> 1;
            g8 = 2 * q + (rr & 1);
        };
        f32x4 A[RING];
        float Bv[2][4];
        int g8, tap;
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) {
            kgrp(s, g8, tap);
            load_a(A[s], a.W3f, tap * (C / 8) + g8, mt, lane);
        }
        kgrp(0, g8, tap);
        load_b(Bv[0], u_lds, g8 * 8, tap, lane);
        for (int it = 0; it < NG; it += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                kgrp(it + s + RING - 1, g8, tap);
                load_a(A[(s + RING - 1) % RING], a.W3f, tap * (C / 8) + g8, mt, lane);
                kgrp(it + s + 1, g8, tap);
                load_b(Bv[(s + 1) & 1], u_lds, g8 * 8, tap, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(acc, A[s], Bv[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // gate: the tile is [16 sigmoid rows | 16 tanh rows] of channels [16 mt, +16): registers r and r+8 pair up
    float* zb = a.z + (long)b * C * T;
    const int t = t0 + l31;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int mg = mt * 32 + acc_row(r, lane);
        const float zv = cmtts_gate(acc[r] + a.b3[mg], acc[r + 8] + a.b3[mg + 16]);
        if (t < T) zb[(unsigned)((mt * 16 + acc_row(r, lane)) * T + t)] = zv;
    }
}

__global__ __launch_bounds__(64 * NWS) void resblock_split_out_kernel(const ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) float z_lds[];     // [C][U_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = blockIdx.z * NWS + w;
    const int b = blockIdx.y, t0 = blockIdx.x * FN, T = a.T;
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* xin = a.x_in + (long)b * C * T;
    const float* dv = a.d + (long)b * a.vec_stride;
    const float* zb = a.z + (long)b * C * T;
    {   // stage z[m][n]: columns beyond T are never stored by the epilogue, any finite value will do
        const int t_c = min(t0 + l31, T - 1);
        constexpr int ROWS_PER_WAVE = C / NWS;
#pragma unroll 1
        for (int i = 0; i < ROWS_PER_WAVE / 2; i += 8) {
            float zv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) zv[q] = zb[(unsigned)((w * ROWS_PER_WAVE + (i + q) * 2 + khalf) * T + t_c)];
#pragma unroll
            for (int q = 0; q < 8; ++q) z_lds[(w * ROWS_PER_WAVE + (i + q) * 2 + khalf) * U_LD + l31] = zv[q];
        }
    }
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        constexpr int NG = C / 8;                              // 32 k-groups
        f32x4 A[RING];
        float Bv[2][4];
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) load_a(A[s], a.Wof, s, mt, lane);
        load_b(Bv[0], z_lds, 0, 0, lane);
        for (int it = 0; it < NG; it += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                load_a(A[(s + RING - 1) % RING], a.Wof, min(it + s + RING - 1, NG - 1), mt, lane);
                load_b(Bv[(s + 1) & 1], z_lds, min(it + s + 1, NG - 1) * 8, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(acc, A[s], Bv[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float* xout = a.x_out + (long)b * C * T;
    float* skip = a.skip + (long)b * C * T;
    const bool res_half = mt < 8;              // wave-uniform: rows [0,256) -> x', rows [256,512) -> skip
    const float* src = res_half ? xin : skip;
    float* dst = res_half ? xout : skip;
    const bool need_src = res_half || a.accum_skip;
    const int mrow0 = (mt % 8) * 32;
    float bo[16], dd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bo[r] = a.bo[mt * 32 + acc_row(r, lane)];
        dd[r] = res_half ? dv[mrow0 + acc_row(r, lane)] : 0.f;
    }
    const int t = t0 + l31, t_c = min(t, T - 1);
    float sv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = need_src ? src[(unsigned)((mrow0 + acc_row(r, lane)) * T + t_c)] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = mrow0 + acc_row(r, lane);
        const float o = acc[r] + bo[r];
        float v;
        if (res_half) v = (o + (sv[r] + dd[r])) * CMTTS_RSQRT2;
        else v = a.accum_skip ? o + sv[r] : o;
        if (t < T) dst[(unsigned)(mr * T + t)] = v;
    }
    (void)khalf;
}

}  // namespace

// ResidualBlock.forward as two launches for small batches; needs a->z (scratch [B][256][T]).  0 ok, -2 not served
// (no scratch / shape), -3 HIP error.  The caller decides WHEN (cmtts_api.hip: few 32-frame tiles).
extern "C" int cmtts_launch_resblock_split(const ResArgs* ap, void* stream_) {
    const ResArgs& a = *ap;
    hipStream_t stream = (hipStream_t)stream_;
    if (!a.z || a.B <= 0 || a.T <= 0 || (long)C * a.T >= (1L << 31)) return -2;
    const size_t lds = (size_t)C * U_LD * sizeof(float);
    dim3 grid((a.T + FN - 1) / FN, a.B, MS);
    hipLaunchKernelGGL(resblock_split_conv_kernel, grid, dim3(64 * NWS), lds, stream, a);
    hipLaunchKernelGGL(resblock_split_out_kernel, grid, dim3(64 * NWS), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
