// X-resident 16-bit Conv1D for the text side's K = 256 contractions (the opt-in "text16": in- / out-projection, k = 9 FFN conv, k = 3 / 5
// predictor convs of a bf16 / fp16 model): the 16-bit counterpart of conv_xres.hip.  The chunked kernel (conv_mfma16.hip) stages 32 channels
// at a time behind a barrier — eight round trips per tile for sequences of 85 phonemes, 18 % of the 16-bit pipe.  Here a workgroup (4 waves,
// 128 output rows) stages the whole x^T tile [96 + k - 1 columns][256 + 8 channels] of one utterance ONCE (converted while staged), every wave
// owns one 32-row m-tile over three 32-column n-tiles, and the weights stream L2 -> VGPR in A-fragment order through a hand-issued ring
// (conv_loop16.h explains why by hand); a B fragment is one ds_read_b128; no barrier in the K loop.
// Same conversions, same (32-channel chunk, tap, k-group) accumulation order and the same epilogue (conv_epilogue.h: bias, alpha, none / ReLU /
// GELU, residual, length mask) as conv_mfma16.hip's text path => BITWISE equal to it (tests/test_gpu_parity.py::test_text16_xresident_bitwise).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "conv_args.h"
#include "conv_epilogue.h"
#include "cvt16.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KC = 256;         // input channels
constexpr int RS = KC + 8;      // image row in 16-bit elements: a multiple of 16 bytes
constexpr int BN = 96, NT = 3;  // columns per workgroup
constexpr int RING = 8;
constexpr int G = KC / 16;

template <int MODE>
__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if (MODE == 1)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int LO, int N, int SEG, class F>
__device__ __forceinline__ void seg_loop(F& body) {
    constexpr int HI = LO + SEG < N ? LO + SEG : N;
#pragma unroll
    for (int it = LO; it < HI; ++it) body(it);
    if constexpr (HI < N) seg_loop<HI, N, SEG>(body);
}

template <int KT, int MODE>
__global__ __launch_bounds__(256, 2) void conv_xt16_kernel(const ConvArgs a, const u32x4* __restrict__ wfrag) {
    constexpr int XROWS = BN + KT - 1;
    constexpr int NG = G * KT;
    extern __shared__ __attribute__((aligned(16))) unsigned short xt[];      // [XROWS][RS], row j <-> t = n0 - pad + j
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int n0 = blockIdx.x * BN, z = blockIdx.z;
    const int mt = blockIdx.y * 4 + w;
    const int MTn = a.M / 32;
    const float* Xb = a.X + z * a.x_zs0;
    {   // stage x^T: wave w converts channels 64 w .. 64 w + 63 of every column; lanes run over columns; 32 loads in flight per lane
#pragma unroll
        for (int jb = 0; jb < (XROWS + 63) / 64; ++jb) {
            const int j = jb * 64 + lane;
            const int t = n0 - a.pad + j;
            const bool ok = t >= 0 && t < a.Tin;
            const unsigned t_c = (unsigned)min(max(n0 - a.pad + min(j, XROWS - 1), 0), a.Tin - 1);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                float v[16][2];
#pragma unroll
                for (int p = 0; p < 16; ++p)
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[p][h] = Xb[(unsigned)((w * 64 + hb * 32 + 2 * p + h) * a.ldx) + t_c];
                if (j < XROWS) {
#pragma unroll
                    for (int p = 0; p < 16; ++p)      // conv_mfma16.hip's staging with pre_div = pre_slope = 1: v * 1 (or * 0 outside the sequence)
                        *reinterpret_cast<unsigned*>(xt + j * RS + w * 64 + hb * 32 + 2 * p) = pack16<MODE>(v[p][0] * (ok ? 1.f : 0.f), v[p][1] * (ok ? 1.f : 0.f));
                }
            }
        }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto grp = [&](int it, int& chunk, int& tap, int& kgl) {
        chunk = it / (2 * KT);
        const int rr = it - chunk * (2 * KT);
        tap = rr >> 1;
        kgl = rr & 1;
    };
    u32x4 A[RING];
    auto issue_a = [&](u32x4& dst, int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const u32x4* ptr = wfrag + ((long)(tap * G + 2 * chunk + kgl) * MTn + mt) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s)
        if (s < NG) issue_a(A[s], s);
    __syncthreads();
    const unsigned short* bl = xt + l31 * RS + khalf * 8;
    auto load_b = [&](u32x4 (&dst)[NT], int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const unsigned short* p = bl + tap * RS + chunk * 32 + kgl * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) dst[j] = *reinterpret_cast<const u32x4*>(p + j * 32 * RS);
    };
    u32x4 Bf[2][NT];
    load_b(Bf[0], 0);
    auto body = [&](int it) {
        if (it + RING - 1 < NG) {
            issue_a(A[(it + RING - 1) % RING], it + RING - 1);
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(A[it % RING]) : "n"(RING - 1));
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[it % RING]));
        }
        if (it + 1 < NG) load_b(Bf[(it + 1) & 1], it + 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mma16<MODE>(A[it % RING], Bf[it & 1][j], acc[j]);
        if (it + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
    };
    seg_loop<0, NG, 36>(body);

    const ConvOut& o = a.out[0];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 32 + l31;
        if (o.act == ACT_GELU_ERF) epi_tile_simple<ACT_GELU_ERF>(o, acc[j], mt * 32, 4 * khalf, n, a.M, a.N, z);
        else if (o.act == ACT_RELU) epi_tile_simple<ACT_RELU>(o, acc[j], mt * 32, 4 * khalf, n, a.M, a.N, z);
        else epi_tile_simple<ACT_NONE>(o, acc[j], mt * 32, 4 * khalf, n, a.M, a.N, z);
    }
}

template <int KT, int MODE>
int launch_xt16(const ConvArgs& a, const void* wfrag, int nbatch, hipStream_t stream) {
    const size_t lds = (size_t)(BN + KT - 1) * RS * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xt16_kernel<KT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.N + BN - 1) / BN, a.M / 128, nbatch);
    hipLaunchKernelGGL((conv_xt16_kernel<KT, MODE>), grid, dim3(256), lds, stream, a, (const u32x4*)wfrag);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE>
int dispatch_xt16(const ConvArgs& a, const void* wfrag, int nbatch, hipStream_t s) {
    switch (a.taps) {
        case 1: return launch_xt16<1, MODE>(a, wfrag, nbatch, s);
        case 3: return launch_xt16<3, MODE>(a, wfrag, nbatch, s);
        case 5: return launch_xt16<5, MODE>(a, wfrag, nbatch, s);
        case 9: return launch_xt16<9, MODE>(a, wfrag, nbatch, s);
        default: return -2;
    }
}

}  // namespace

int g_conv_xt16 = 1;      // internal switch "text_xt16": 0 = the chunked kernel for every text16 conv (same bits)

// The text-side convs with K = 256 input channels (ConvArgs::text_epi), `wfrag` = to_fragment16 weights, mode 1 = bf16, 2 = fp16.
// 0 = launched, -2 = shape not covered (cmtts_launch_conv16 then runs the chunked kernel: same bits), -3 = HIP error.
extern "C" int cmtts_launch_conv_xt16(const ConvArgs* ap, const void* wfrag, int mode, int nbatch, void* stream_) {
    const ConvArgs& a = *ap;
    if (!g_conv_xt16 || !a.text_epi || a.K != KC || a.M % 128 != 0 || a.dil != 1 || a.pad != (a.taps - 1) / 2 || a.pre_div != 1.f || a.pre_slope != 1.f || a.zdiv != 1 ||
        a.x16 || a.y16 || (mode != 1 && mode != 2) || (long)a.K * a.ldx >= (1L << 31))
        return -2;
    const ConvOut& o = a.out[0];
    if (o.ostride != 1 || o.ooff_base != 0 || o.ooff_mul != 0 || o.row_off != 0 || o.div != 1.0f || o.accum || o.bvec || o.Tout != a.N ||
        (o.act != ACT_NONE && o.act != ACT_RELU && o.act != ACT_GELU_ERF))
        return -2;
    return mode == 1 ? dispatch_xt16<1>(a, wfrag, nbatch, (hipStream_t)stream_) : dispatch_xt16<2>(a, wfrag, nbatch, (hipStream_t)stream_);
}
