// Stacked conditioner projections for gfx950: cp[b][l*256 + m][t] = sum_k Wc_l[m][k] * cond[b][k][t] + bc_l[m] for ALL
// residual layers at once (ResidualBlock.conditioner_projection, model/blocks.py:663,676) — a k=1 "convolution" with
// M = 20 * 256 output rows and only K = 256 input channels.  The generic kernel (conv_mfma.hip) re-stages the same
// X tile for each of the 40 m-tiles and pays a prologue and an epilogue per 128x128 tile for 16 short iterations
// (71 TFLOP/s).  Here the X tile of a workgroup ([256 channels][64 frames]) is staged in LDS ONCE and the workgroup
// walks over all M: 8 waves, each 2x2 MFMA tiles per pass (512 rows per pass), weights streamed L2 -> VGPR in
// A-fragment order through a register ring exactly as in denoiser_persist.hip; stores of pass p overlap the MFMAs of
// pass p+1.  Same (16-channel chunk, k) accumulation order as the generic kernel: BITWISE equal
// (tests/test_gpu_parity.py::test_cond_gemm_bitwise).
#include <hip/hip_runtime.h>
#include "cond_gemm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int K = 256;          // input channels (encoder hidden)
constexpr int NW = 8;
constexpr int MT = 2;
constexpr int FN = 64;
constexpr int NT = FN / 32;
constexpr int X_LD = FN + 4;
constexpr int RING = 4;
constexpr int NG = K / 8;       // k-groups of 8 channels

__device__ __forceinline__ float ldg(const float* base, unsigned idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 4u));
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(64 * NW, 2) void cond_gemm_kernel(const CondGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [K][X_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t0 = blockIdx.x * FN;
    const int l31 = lane & 31, khalf = lane >> 5;
    // columns: frames of utterance blockIdx.y (row stride T), or — a.flat — the columns of ALL utterances in one axis (c = b * T + t:
    // a k = 1 contraction is column-local, so a tile may span utterances; no per-utterance padding to a multiple of 64)
    const int T = a.T;
    const long ncol = a.flat ? (long)a.B * T : T;
    auto col_x = [&](long c) -> long { return a.flat ? (c / T) * (long)K * T + (c % T) : (long)blockIdx.y * K * T + c; };
    auto col_y = [&](long c) -> long { return a.flat ? (c / T) * (long)a.M * T + (c % T) : (long)blockIdx.y * a.M * T + c; };
    const float* xin = a.X;
    float* yb = a.Y;

    {   // stage X[k][t0 .. t0+63] (zero beyond the last column): lane = column, 32 rows per wave, 8 in flight
        const long t = t0 + lane;
        const long xo = col_x(t < ncol ? t : ncol - 1);
#pragma unroll 1
        for (int i = 0; i < K / NW; i += 8) {
            float xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = xin[xo + (long)(w * (K / NW) + i + q) * T];
#pragma unroll
            for (int q = 0; q < 8; ++q) xs[(w * (K / NW) + i + q) * X_LD + lane] = t < ncol ? xv[q] : 0.f;
        }
    }
    const int MTn = a.M / 32;
    auto load_a = [&](f32x4 (&dst)[MT], int mt0, int g) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
            dst[i] = *reinterpret_cast<const f32x4*>(a.Wf + (((long)g * MTn + mt0 + i) * 64 + lane) * 4);
    };
    auto load_b = [&](float (&dst)[4][NT], int g) {
        const float* bs = xs + (g * 8 + khalf) * X_LD + l31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j) dst[kk][j] = bs[2 * kk * X_LD + j * 32];
    };
    const int npass_all = a.M / (32 * MT * NW);      // 512 rows per pass
    // gridDim.z workgroups share a frame tile, each walking a contiguous share of the passes (the phoneme-level factor of the
    // conditioner projections has few tiles: the rows spread the work instead); an element's accumulation chain does not change
    const int ppg = (npass_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int p_first = (int)blockIdx.z * ppg;
    const int npass = min(npass_all, p_first + ppg);
    if (p_first >= npass) return;
    f32x4 A[RING][MT];
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load_a(A[s], (p_first * NW + w) * MT, s);
    __syncthreads();

    for (int p = p_first; p < npass; ++p) {
        const int mt0 = (p * NW + w) * MT;           // this wave's first m-tile in this pass
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float Bv[2][4][NT];
        load_b(Bv[0], 0);
        const int mt_next = (min(p + 1, npass - 1) * NW + w) * MT;
#pragma unroll 1
        for (int it = 0; it < NG; it += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                const int nx = it + s + RING - 1;    // prefetch runs into the next pass's first k-groups
                if (nx < NG) load_a(A[(s + RING - 1) % RING], mt0, nx);
                else load_a(A[(s + RING - 1) % RING], mt_next, nx - NG);
                load_b(Bv[(s + 1) & 1], min(it + s + 1, NG - 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][i][kk], Bv[s & 1][kk][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // bias + store (the stores drain under the next pass)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m0 = (mt0 + i) * 32;
            float bi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bi[r] = ldg(a.bias, (unsigned)(m0 + acc_row(r, lane)));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const long t = t0 + j * 32 + l31;
                const long yo = col_y(t < ncol ? t : ncol - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t < ncol) yb[yo + (long)(m0 + acc_row(r, lane)) * T] = acc[i][j][r] + bi[r];
            }
        }
    }
}

}  // namespace

// Returns 0, -2 (shape not supported: use the generic kernel) or -3.
extern "C" int cmtts_launch_cond_gemm(const CondGemmArgs* ap, void* stream_) {
    const CondGemmArgs& a = *ap;
    if (a.K != K || a.M % (32 * MT * NW) != 0 || (long)a.M * a.T >= (1L << 30) || a.B <= 0 || a.T <= 0) return -2;
    if (a.flat && (long)a.B * a.T >= (1L << 30)) return -2;
    // a workgroup walks all M rows of its 64 frames alone (~350 us whatever the batch): below ~half a chip of frame tiles
    // the generic kernel, which spreads M over workgroups, finishes sooner (one 150-frame utterance: 347 -> ~40 us).
    // Both are bitwise equal (tests), so the choice never changes a result.
    if (!a.force && (long)((a.T + FN - 1) / FN) * a.B < 128) return -2;
    static bool attr_set = false;
    const size_t lds = (size_t)K * X_LD * sizeof(float);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(cond_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const int zsplit = a.row_split > 1 ? a.row_split : 1;
    const dim3 grid = a.flat ? dim3((unsigned)(((long)a.B * a.T + FN - 1) / FN), 1, zsplit) : dim3((a.T + FN - 1) / FN, a.B, zsplit);
    hipLaunchKernelGGL(cond_gemm_kernel, grid, dim3(64 * NW), lds, (hipStream_t)stream_, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

namespace {
// cp[b][r][t] = (ph > 0 ? P1[b][r][ph - 1] : 0) + P2[r][idx],  ph = mel2ph[b][t], idx = p_idx[b][t]:
// the conditioner projections of all layers expanded from their phoneme-level and pitch-table factors (cmtts_api.hip: cond_factored).
// HBM-bound on the [B][M][T] write; the factors are L2 / Infinity-Cache resident (consecutive frames read the same or the
// neighbouring phoneme and a neighbouring pitch bucket).  One thread = one frame, CEX_ROWS rows per workgroup, 8 rows in flight.
constexpr int CEX_ROWS = 64;
// V = frames per thread: 4 when T % 4 == 0 (16-byte stores; the 4 gathers of a row are independent loads), else 1
template <int V>
__global__ __launch_bounds__(256) void cond_expand_kernel(const float* __restrict__ p1, int ldp, int L, const float* __restrict__ p2, int ld2,
                                                          const int64_t* __restrict__ mel2ph, const int64_t* __restrict__ pidx,
                                                          float* __restrict__ cp, int M, int T) {
    // a wave = 64 threads along the frame axis (64 V consecutive frames), the four waves take a quarter of the rows each
    const int t = (blockIdx.x * 64 + (threadIdx.x & 63)) * V;
    const int r0 = blockIdx.y * CEX_ROWS + (threadIdx.x >> 6) * (CEX_ROWS / 4), b = blockIdx.z;
    if (t >= T) return;
    int ph[V], ix[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int64_t p64 = mel2ph[(long)b * T + t + e];
        ph[e] = (int)(p64 > L ? L : p64);
        const int i = (int)pidx[(long)b * T + t + e];
        ix[e] = i < 0 ? 0 : (i >= ld2 ? ld2 - 1 : i);
    }
    const float* a = p1 + ((long)b * M + r0) * ldp;
    const float* q = p2 + (long)r0 * ld2;
    float* o = cp + ((long)b * M + r0) * T + t;
#pragma unroll 1
    for (int r = 0; r < CEX_ROWS / 4; r += 4) {
        float av[4][V], qv[4][V];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < V; ++e) {
                av[j][e] = a[(long)(r + j) * ldp + (ph[e] > 0 ? ph[e] - 1 : 0)];
                qv[j][e] = q[(long)(r + j) * ld2 + ix[e]];
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (V == 4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (ph[e] > 0 ? av[j][e] : 0.f) + qv[j][e];
                *reinterpret_cast<f32x4*>(o + (long)(r + j) * T) = v;
            } else {
                o[(long)(r + j) * T] = (ph[0] > 0 ? av[j][0] : 0.f) + qv[j][0];
            }
        }
    }
}
}  // namespace

extern "C" int cmtts_launch_cond_expand(const float* p1, int ldp, int L, const float* p2, int ld2, const int64_t* mel2ph, const int64_t* pidx,
                                        float* cp, int B, int M, int T, void* stream_) {
    if (M % CEX_ROWS != 0 || B <= 0 || T <= 0 || L <= 0) return -2;
    if ((T & 3) == 0 && ((uintptr_t)cp & 15) == 0)
        hipLaunchKernelGGL(cond_expand_kernel<4>, dim3((T / 4 + 63) / 64, M / CEX_ROWS, B), dim3(256), 0, (hipStream_t)stream_, p1, ldp, L, p2, ld2,
                           mel2ph, pidx, cp, M, T);
    else
        hipLaunchKernelGGL(cond_expand_kernel<1>, dim3((T + 63) / 64, M / CEX_ROWS, B), dim3(256), 0, (hipStream_t)stream_, p1, ldp, L, p2, ld2,
                           mel2ph, pidx, cp, M, T);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
