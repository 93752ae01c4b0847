// Stacked conditioner projections with 16-bit MFMA operands (bf16 / fp16, fp32 accumulate) — the 16-bit twin of cond_gemm.hip for
// set_precision("bf16" | "fp16") models (round 3): cp[b][l*C + m][t] = sum_k q16(Wc_l[m][k]) * q16(cond[b][k][t]) + bc_l[m] for all residual
// layers at once (ResidualBlock.conditioner_projection, model/blocks.py:663,676).  With fp32 operands this GEMM was 0.38-0.41 ms of a 4.4-ms
// bf16 sampler call (72 % of the fp32 matrix pipe); at the 16-bit rate it is bound by its 335-MB output and by the L2 -> CU weight stream.
//   * the x^T tile [64 frames][256 + 8 channels] is converted while staged (v_cvt_pk, round to nearest even) — ONE image per workgroup;
//   * 8 waves walk all M rows: per pass a wave owns 2 m-tiles x 2 n-tiles, weights stream L2 -> VGPR in MFMA A-fragment order through a
//     hand-issued ring that runs on across the passes (conv_loop16.h explains why by hand); a B fragment is one ds_read_b128;
//   * k-groups in ascending order, fp32 bias add, stores of pass p drain under the MFMAs of pass p + 1.
// The oracle's operands16 modes quantise the same two operands (oracle/cmtts_oracle.py denoiser_forward).
#include <hip/hip_runtime.h>
#include "cond_gemm.h"
#include "cvt16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int K = 256;          // input channels (encoder hidden)
constexpr int NW = 8;
constexpr int MT = 2;
constexpr int FN = 64;
constexpr int NT = FN / 32;
constexpr int RS = K + 8;       // image row in 16-bit elements: a multiple of 16 bytes
constexpr int G = K / 16;       // MFMA k-groups
// k-groups (of MT fragments per operand set) in the ring per wave: a divisor of G, so that a group keeps its slot from pass to pass
template <int MODE> constexpr int ring_of() { return MODE == 3 ? 4 : 8; }
static_assert(G % 8 == 0 && G % 4 == 0, "ring slots carry over from one pass to the next");

template <int MODE>
__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if (MODE == 1)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// MODE 3 ("fp16x3", fp32-class): two images (hi, lo) of x^T, the lo fragment set behind the hi set, three fp16 MFMAs per product, small terms first
template <int MODE>
__global__ __launch_bounds__(64 * NW) void cond_gemm16_kernel(const CondGemmArgs a, const u32x4* __restrict__ wfrag) {
    constexpr int NS = MODE == 3 ? 2 : 1;
    constexpr int MM = MODE == 3 ? 2 : MODE;
    constexpr int RING = ring_of<MODE>();
    constexpr int XIMG = FN * RS;
    extern __shared__ __attribute__((aligned(16))) unsigned short xs16[];     // [NS][FN][RS]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FN;
    const int T = a.T;
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* xin = a.X + (long)b * K * T;
    float* yb = a.Y + (long)b * a.M * T;
    const int MTn = a.M / 32;
    const int npass = (MTn + MT * NW - 1) / (MT * NW);       // the last pass may leave waves without rows (M a multiple of 64 only)

    const long wset = (long)G * MTn * 64;                  // fragments per operand set
    u32x4 A[RING][MT][NS];
    auto issue_a = [&](u32x4 (&dst)[MT][NS], int p, int g) {
        const int mt0 = min((p * NW + w) * MT, MTn - MT);   // a wave without rows in this pass re-reads the last tiles (its results are dropped)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const u32x4* ptr = wfrag + q * wset + ((long)g * MTn + mt0 + i) * 64 + lane;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[i][q]) : "v"(ptr) : "memory");
            }
    };
    {   // stage x^T[t0 .. t0+63][k] (zero beyond T): lane = frame, wave w converts channel pairs 16 w .. 16 w + 15
        const int t = t0 + lane;
        const unsigned t_c = (unsigned)min(t, T - 1);
        float v[16][2];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[p][h] = xin[(unsigned)((w * 32 + 2 * p + h) * T) + t_c];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const unsigned hi = t < T ? pack16<MM>(v[p][0], v[p][1]) : 0u;
            *reinterpret_cast<unsigned*>(xs16 + lane * RS + w * 32 + 2 * p) = hi;
            if (MODE == 3) {
                const cvt_f16x2 h = __builtin_bit_cast(cvt_f16x2, hi);
                *reinterpret_cast<unsigned*>(xs16 + XIMG + lane * RS + w * 32 + 2 * p) = t < T ? pack16<2>(v[p][0] - (float)h[0], v[p][1] - (float)h[1]) : 0u;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) issue_a(A[s], 0, s);
    __syncthreads();

    const unsigned short* bl = xs16 + l31 * RS + khalf * 8;
    for (int p = 0; p < npass; ++p) {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int pn = min(p + 1, npass - 1);              // the prefetch runs into the next pass's first k-groups (the last pass re-reads its own)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int nx = g + RING - 1;
            if (nx < G) issue_a(A[nx % RING], p, nx);
            else issue_a(A[nx % RING], pn, nx - G);
            // "at most (RING - 1) MT younger operations outstanding" = group g has landed: vector memory operations complete in order, and the
            // previous pass's stores and bias loads in between only make the wait stricter
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(A[g % RING][0][0]) : "n"((RING - 1) * MT * NS));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (i + q) asm volatile("" : "+v"(A[g % RING][i][q]));
            u32x4 Bf[NT][NS];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < NS; ++q) Bf[j][q] = *reinterpret_cast<const u32x4*>(bl + q * XIMG + j * 32 * RS + g * 16);
            if (NS == 2) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][j] = mma16<MM>(A[g % RING][i][NS - 1], Bf[j][0], acc[i][j]);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][j] = mma16<MM>(A[g % RING][i][0], Bf[j][NS - 1], acc[i][j]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j] = mma16<MM>(A[g % RING][i][0], Bf[j][0], acc[i][j]);
        }
        // bias + store (the stores drain under the next pass)
        const int mt0 = (p * NW + w) * MT;
        if (mt0 + MT > MTn) continue;                      // wave-uniform
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m0 = (mt0 + i) * 32;
            float bi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bi[r] = a.bias[m0 + acc_row(r, lane)];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = t0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t < T) yb[(unsigned)((m0 + acc_row(r, lane)) * T + t)] = acc[i][j][r] + bi[r];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring's tail (re-issued fragments of the last pass)
}

template <int MODE>
int launch_cg16(const CondGemmArgs& a, const void* wfrag, hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = (size_t)(MODE == 3 ? 2 : 1) * FN * RS * sizeof(unsigned short);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(cond_gemm16_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    hipLaunchKernelGGL(cond_gemm16_kernel<MODE>, dim3((a.T + FN - 1) / FN, a.B), dim3(64 * NW), lds, s, a, (const u32x4*)wfrag);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// wf16: to_fragment16 of the stacked weights [M][K] with one tap ([K/16][M/32][64][8] 16-bit elements); a->Wf is ignored.  mode 1 = bf16, 2 = fp16, 3 = fp16x3 (to_fragment16_split: the hi set followed by the lo set).
// Returns 0, -2 (shape not supported: the caller runs the fp32 kernels) or -3.
extern "C" int cmtts_launch_cond_gemm16(const CondGemmArgs* ap, const void* wf16, int mode, void* stream_) {
    const CondGemmArgs& a = *ap;
    if (!wf16 || a.K != K || a.M % (32 * MT) != 0 || (long)a.M * a.T >= (1L << 30) || a.B <= 0 || a.T <= 0 || mode < 1 || mode > 3) return -2;
    if (!a.force && (long)((a.T + FN - 1) / FN) * a.B < 128) return -2;        // few frame tiles: the generic kernel spreads M over workgroups (cond_gemm.hip)
    if (mode == 3) return launch_cg16<3>(a, wf16, (hipStream_t)stream_);
    return mode == 1 ? launch_cg16<1>(a, wf16, (hipStream_t)stream_) : launch_cg16<2>(a, wf16, (hipStream_t)stream_);
}
