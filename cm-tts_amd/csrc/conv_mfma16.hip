// 16-bit-operand implicit-GEMM Conv1D for gfx950: the HiFi-GAN ResBlock convs (94 % of the generator's
// FLOPs) on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation.  Activations stay fp32 in HBM; they are
// converted (after the fused x/pre_div and leaky-ReLU) while being staged into LDS.  BASELINE.json
// configs[2] ("bf16 ... + HiFi-GAN universal vocoder"); the reference's vocoder is fp32-only, so parity is
// stated against the fp32 golden wav with its own tolerance.
//
//  * X tile in LDS is TRANSPOSED: [frame][32 input channels] 16-bit, row stride 72 B, so the B fragment
//    of lane l (8 consecutive channels of frame l&31) is two conflict-free ds_read_b64 and a dilated tap
//    is a row offset.  One tile per 32-channel chunk, double buffered, one barrier per chunk; the next
//    chunk's global loads are issued before this chunk's MFMAs.
//  * Weights are pre-packed in A-fragment order [tap][k-group of 16][m-tile][lane][8] and streamed
//    L2 -> VGPR (one global_load_dwordx4 per MFMA operand), two fragments ahead.
//  * Same tiles / wave layout / fused epilogue (conv_epilogue.h) as the fp32 kernel.
// At 16x the fp32 MFMA rate these convs are HBM-bound (3 fp32 tensors per conv), not MFMA-bound.
#include <hip/hip_runtime.h>
#include <limits.h>
#include "conv_args.h"
#include "conv_epilogue.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KC = 32;     // input channels per LDS stage (2 MFMA k-groups)
constexpr int RSX = 36;    // 16-bit elements per LDS row (72 B)

template <int MODE>
__device__ __forceinline__ unsigned cvt16(float f) {
    if (MODE == 1) {
        const unsigned u = __float_as_uint(f);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    } else {
        const _Float16 h = (_Float16)f;
        return (unsigned)__builtin_bit_cast(unsigned short, h);
    }
}

template <int MODE>
__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if (MODE == 1)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256, 3) void conv1d_mfma16_kernel(const ConvArgs a, const u32x4* __restrict__ wfrag) {
    constexpr int MT = BM / (WM * 32);
    constexpr int NT = BN / (WN * 32);
    constexpr int XJ = BN / 64 + 1;       // frames per lane of a staged channel row (halo <= 64)
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];   // [2][XW][RSX]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int adil = a.dil < 0 ? -a.dil : a.dil;
    const int XW = BN + (a.taps - 1) * adil;
    const int tap_min = a.dil < 0 ? (a.taps - 1) * a.dil : 0;
    const int tbase = n0 - a.pad + tap_min;
    const float* Xb = a.X + z * a.x_zs0;
    const int G = (a.K + 15) / 16;                 // k-groups
    const int MTn = (a.M + 31) / 32;               // m-tiles in the packed weights
    const int nchunks = (a.K + KC - 1) / KC;

    // ---- staging: wave `wid` converts channel pairs 4*wid .. 4*wid+3 of the chunk, lanes run over frames
    float xr[4][2][XJ];
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ch = chunk * KC + (wid * 4 + p) * 2 + h;
                const float* xrow = Xb + (long)min(ch, a.K - 1) * a.ldx;
#pragma unroll
                for (int j = 0; j < XJ; ++j) xr[p][h][j] = xrow[min(max(tbase + lane + 64 * j, 0), a.Tin - 1)];
            }
    };
    auto store_x = [&](int buf, int chunk) {
        unsigned short* xb = xs + buf * XW * RSX;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ch = chunk * KC + (wid * 4 + p) * 2;
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                const int col = lane + 64 * j;
                const int t = tbase + col;
                const bool okt = t >= 0 && t < a.Tin;
                float v0 = (okt && ch < a.K) ? xr[p][0][j] : 0.f;
                float v1 = (okt && ch + 1 < a.K) ? xr[p][1][j] : 0.f;
                if (a.pre_div != 1.0f) { v0 = v0 / a.pre_div; v1 = v1 / a.pre_div; }
                v0 = v0 > 0.f ? v0 : v0 * a.pre_slope;
                v1 = v1 > 0.f ? v1 : v1 * a.pre_slope;
                if (col < XW) *reinterpret_cast<unsigned*>(xb + col * RSX + (wid * 4 + p) * 2) = cvt16<MODE>(v0) | (cvt16<MODE>(v1) << 16);
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_a = [&](u32x4 (&dst)[MT], int tap, int kg) {
        const int kgc = min(kg, G - 1);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mt = min(m0 / 32 + wm * MT + i, MTn - 1);
            dst[i] = wfrag[((long)(tap * G + kgc) * MTn + mt) * 64 + lane];
        }
    };
    auto load_b = [&](u32x4 (&dst)[NT], const unsigned short* xb, int tap, int kgl) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const unsigned short* p = xb + ((wn * NT + j) * 32 + l31 + tap * a.dil - tap_min) * RSX + kgl * 16 + khalf * 8;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(p);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(p + 4);
            dst[j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };

    load_x(0);
    store_x(0, 0);
    __syncthreads();
    const int nq = a.taps * 2;                     // (tap, k-group-in-chunk) pairs per chunk
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool has_next = chunk + 1 < nchunks;
        if (has_next) load_x(chunk + 1);
        const unsigned short* xb = xs + (chunk & 1) * XW * RSX;
        u32x4 A[2][MT], Bv[2][NT];
        load_a(A[0], 0, chunk * 2);
        load_b(Bv[0], xb, 0, 0);
        for (int q = 0; q < nq; q += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int qn = min(q + s + 1, nq - 1);
                load_a(A[(s + 1) & 1], qn >> 1, chunk * 2 + (qn & 1));
                load_b(Bv[(s + 1) & 1], xb, qn >> 1, qn & 1);
                // a chunk whose second k-group lies beyond K contributes zeros (X rows are zero-filled)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mma16<MODE>(A[s][i], Bv[s][j], acc[i][j]);
            }
        }
        if (has_next) store_x((chunk + 1) & 1, chunk + 1);
        __syncthreads();
    }

    const int rbase = 4 * khalf;
    const ConvOut& o = a.out[0];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            epi_tile(o, acc[i][j], m0 + (wm * MT + i) * 32, rbase, n0 + (wn * NT + j) * 32 + l31, a.M, a.N, z, 0);
}

template <int BM, int BN, int WM, int WN>
int launch16(const ConvArgs& a, const void* wfrag, int mode, int nbatch, hipStream_t stream) {
    const int adil = a.dil < 0 ? -a.dil : a.dil;
    const int halo = (a.taps - 1) * adil;
    if (halo > 64) return -2;
    const size_t lds = (size_t)2 * (BN + halo) * RSX * sizeof(unsigned short);
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nbatch);
    if (mode == 1)
        hipLaunchKernelGGL((conv1d_mfma16_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), lds, stream, a, (const u32x4*)wfrag);
    else
        hipLaunchKernelGGL((conv1d_mfma16_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), lds, stream, a, (const u32x4*)wfrag);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// Plain Conv1d with 16-bit operands: a->A is ignored, `wfrag` = fragment-order weights
// [taps][ceil(K/16)][ceil(M/32)][64][8] (zero padded), mode 1 = bf16, 2 = fp16.  Supports zdiv == 1,
// split == INT_MAX, dil > 0 (the HiFi-GAN ResBlock convs).
extern "C" int cmtts_launch_conv16(const ConvArgs* ap, const void* wfrag, int mode, int nbatch, void* stream_) {
    const ConvArgs& a = *ap;
    hipStream_t stream = (hipStream_t)stream_;
    if (a.M <= 0 || a.N <= 0 || nbatch <= 0) return 0;
    if (a.zdiv != 1 || a.split != INT_MAX || a.dil <= 0 || (mode != 1 && mode != 2)) return -2;
    // 128-frame tiles everywhere: these convs are HBM-bound, what matters is loads in flight (3 workgroups/CU)
    if (a.M > 64) return launch16<128, 128, 2, 2>(a, wfrag, mode, nbatch, stream);
    if (a.M > 32) return launch16<64, 128, 2, 2>(a, wfrag, mode, nbatch, stream);
    return launch16<32, 128, 1, 4>(a, wfrag, mode, nbatch, stream);
}
