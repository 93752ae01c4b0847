// conv_xl16_kernel: ONE conv of a wide (C = 128 / 256) HiFi-GAN ResBlock with 16-bit operands, X-resident (split out of
// resblock_pair16.hip: a translation unit of its own compiles in minutes instead of nine).
#include <hip/hip_runtime.h>
#define CL16_PAD 8          // image rows a multiple of 16 bytes: a B fragment is one ds_read_b128 (bf16 vocoder 14.93 -> 14.80 ms; no gain in the pair kernels)
#include "conv_loop16.h"
#include "xcd_map.h"
#ifndef XCD_MAP
#define XCD_MAP 1
#endif
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// ONE conv of a wide (C = 128 / 256) ResBlock with 16-bit operands, X-resident: the 16-bit twin of conv_xl_kernel
// (resblock_pair.hip) and the replacement of conv_mfma16.hip's chunked kernel for these stages.  Why a new kernel: in the
// chunked kernel the next chunk's activation loads (HBM) and the weight fragments (L2 hits) share one in-order vmcnt queue,
// so every weight fragment issued behind an activation load waited for HBM, and the compiler had sunk the "one step ahead"
// weight loads next to their uses anyway: the K loop ran at 45-50 % of the 16-bit pipe on top of the HBM time instead of under
// it (profiles/r02_vocoder_bf16.md, ablation).  Here the whole x^T tile [BN + halo][C] is staged first (one HBM round trip per
// tile, two or three workgroups per CU overlap it with the others' MFMAs), and the K loop touches only L2 (weights, hand-issued
// ring: conv_loop16) and LDS.  IO = 1: fp32 x in -> xt out as convert(leaky_relu(xt)) in 16 bits; IO = 2: that 16-bit xt in ->
// ((acc + b) + res) + y_old in fp32 — conv_mfma16.hip's conversions, accumulation order and epilogue => the same bits.
template <int C, int KT, int MODE, int IO, int BN, int MT>
__global__ __launch_bounds__(C * 2 / MT, MT == 2 ? 2 : (C == 128 ? 3 : 2)) void conv_xl16_kernel(const ConvXlArgs a) {
    constexpr int RS = C + CL16_PAD;
    constexpr int NWAVES = C / 32 / MT;             // MT m-tiles per wave
    constexpr int NT = BN / 32;                     // all n-tiles of the workgroup's columns
    constexpr int XROWS = BN + (KT - 1) * 5;        // widest halo: dilation 5
    extern __shared__ __attribute__((aligned(16))) unsigned short xl16[];   // [XROWS][RS]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * BN;
    const int T = a.T, dil = a.dil;
    const int pad = dil * ((KT - 1) / 2);
    const int xw = BN + 2 * pad;
    const int tbase = t0 - pad;
    constexpr int XBLK = (XROWS + 63) / 64;
    constexpr int PAIRS = C / 2 / NWAVES;           // channel pairs per wave: 16 (32 with two m-tiles per wave)
    // Staging: ALL loads of the tile are issued before the first LDS store — one HBM round trip per workgroup instead of one per
    // 64-column block.  It matters beyond this workgroup: the CU's vector-memory path returns in order across waves, so while a
    // staging batch is waiting for HBM the co-resident workgroups' weight fragments (L2 hits) queue behind it and their K loops
    // stall (ablation, profiles/r02_vocoder_bf16.md: staging and K loop of DIFFERENT workgroups add up instead of overlapping).
    // Lanes past the tile's last column re-read that column (same cache line, no traffic) instead of the next 64-column block.
    if (IO == 2) {   // 16-bit activated input (what conv1's epilogue wrote): copy, zero outside [0, T)
        const unsigned short* xb = reinterpret_cast<const unsigned short*>(a.x) + (long)b * a.bstride;
        unsigned v[XBLK][PAIRS][2];
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const unsigned t_c = (unsigned)min(max(tbase + min(jb * 64 + lane, xw - 1), 0), T - 1);
#pragma unroll
            for (int p = 0; p < PAIRS; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) v[jb][p][h] = (xb + (long)((w * PAIRS + p) * 2 + h) * a.ld)[t_c];
        }
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const int j = jb * 64 + lane;
            const int t = tbase + j;
            const bool in = t >= 0 && t < T;
            if (j < xw) {
#pragma unroll
                for (int p = 0; p < PAIRS; ++p)
                    *reinterpret_cast<unsigned*>(xl16 + j * RS + (w * PAIRS + p) * 2) = in ? (v[jb][p][0] | (v[jb][p][1] << 16)) : 0u;
            }
        }
    } else {         // fp32 input: leaky_relu, convert, transpose (conv_mfma16.hip's staging arithmetic)
        const float* xb = a.x + (long)b * a.bstride;
        const float slope = a.slope;
        float v[XBLK][PAIRS][2];
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const unsigned t_c = (unsigned)min(max(tbase + min(jb * 64 + lane, xw - 1), 0), T - 1);
#pragma unroll
            for (int p = 0; p < PAIRS; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) v[jb][p][h] = (xb + (long)((w * PAIRS + p) * 2 + h) * a.ld)[t_c];
        }
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const int j = jb * 64 + lane;
            const int t = tbase + j;
            const bool in = t >= 0 && t < T;
            const float fpos = in ? 1.f : 0.f, fneg = in ? slope : 0.f;
            if (j < xw) {
#pragma unroll
                for (int p = 0; p < PAIRS; ++p) {
                    const float v0 = v[jb][p][0], v1 = v[jb][p][1];
                    *reinterpret_cast<unsigned*>(xl16 + j * RS + (w * PAIRS + p) * 2) =
                        pack16<MODE>(v0 * (v0 > 0.f ? fpos : fneg), v1 * (v1 > 0.f ? fpos : fneg));
                }
            }
        }
    }
    __syncthreads();

    f32x16 acc[MT][NT];
    if constexpr (MT == 1) conv_loop16<C, KT, NT, MODE>(acc[0], (const u32x4*)a.wf, xl16, dil, w, 0, lane);
    else conv_loop16m<C, KT, MT, NT, MODE>(acc, (const u32x4*)a.wf, xl16, dil, w * MT, 0, lane);

#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m0 = (w * MT + i) * 32;
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.bias[m0 + acc_row(r, lane)];
        if (IO == 1) {
            unsigned short* y16 = reinterpret_cast<unsigned short*>(a.y) + (long)b * a.bstride;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = t0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bi[r];
                    v = v * (v > 0.f ? 1.f : a.slope);
                    if (t < T) y16[(long)(m0 + acc_row(r, lane)) * a.ld + t] = (unsigned short)pack16<MODE>(v, 0.f);
                }
            }
        } else {
            float* yb = a.y + (long)b * a.bstride;
            const float* rb = a.res ? a.res + (long)b * a.bstride : nullptr;
            // the residual (and, when accumulating, the old y) of JB n-tiles per round trip: 64 loads in flight per lane either way
            unsigned rowoff[16];                    // 32-bit element offsets (C * ld < 2^31, checked by the launcher): one VGPR per address
#pragma unroll
            for (int r = 0; r < 16; ++r) rowoff[r] = (unsigned)(m0 + acc_row(r, lane)) * (unsigned)a.ld;
            auto epi = [&](auto jbc, auto accc) {
                constexpr int JB = decltype(jbc)::value;
                constexpr bool ACC = decltype(accc)::value;
#pragma unroll
                for (int j0 = 0; j0 < NT; j0 += JB) {
                    float rv[JB][16], yv[ACC ? JB : 1][16];
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) {
                        const unsigned t_c = (unsigned)min(t0 + (j0 + jj) * 32 + l31, T - 1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            rv[jj][r] = rb ? rb[rowoff[r] + t_c] : 0.f;
                            if (ACC) yv[jj][r] = yb[rowoff[r] + t_c];
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) {
                        const int t = t0 + (j0 + jj) * 32 + l31;
                        if (t < T) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                yb[rowoff[r] + (unsigned)t] = ((acc[i][j0 + jj][r] + bi[r]) + rv[jj][r]) + (ACC ? yv[jj][r] : 0.f);
                        }
                    }
                }
            };
            // n-tiles per round trip: what the register budget allows (C = 128: 168 VGPRs at three workgroups per CU)
            constexpr int JB0 = C == 128 ? 2 : NT, JB1 = C == 128 ? 1 : (NT >= 2 ? 2 : 1);
            if (a.accum) epi(std::integral_constant<int, JB1>{}, std::true_type{});
            else epi(std::integral_constant<int, JB0>{}, std::false_type{});
        }
    }
}

// columns per workgroup at C = 256: 128 (one 93-KB workgroup of 8 waves per CU, four n-tiles per weight fragment) beats 64 (two
// 59-KB workgroups, two n-tiles per fragment): bf16 vocoder 16.95 -> 16.61 ms
#ifndef XL16_BN256
#define XL16_BN256 128
#endif
template <int C, int KT, int MODE, int IO>
int launch_xl16(const ConvXlArgs& a, hipStream_t stream) {
    constexpr int BN = (C == 128 || XL16_BN256 == 128) ? 128 : 64;
    constexpr int MT = 1;                            // (C = 128 with two waves of 2 x 4 tiles, conv_loop16m: same K-loop slope, slower staging: 460 / 538 vs 429 / 483 us at k = 11)
    const size_t lds = (size_t)(BN + (KT - 1) * 5) * (C + CL16_PAD) * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xl16_kernel<C, KT, MODE, IO, BN, MT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.T + BN - 1) / BN, a.B);
    hipLaunchKernelGGL((conv_xl16_kernel<C, KT, MODE, IO, BN, MT>), grid, dim3(C * 2 / MT), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE, int IO>
int dispatch_xl16(const ConvXlArgs& a, hipStream_t s) {
    if (a.C == 128) {
        if (a.k == 3) return launch_xl16<128, 3, MODE, IO>(a, s);
        if (a.k == 7) return launch_xl16<128, 7, MODE, IO>(a, s);
        if (a.k == 11) return launch_xl16<128, 11, MODE, IO>(a, s);
    } else if (a.C == 256) {
        if (a.k == 3) return launch_xl16<256, 3, MODE, IO>(a, s);
        if (a.k == 7) return launch_xl16<256, 7, MODE, IO>(a, s);
        if (a.k == 11) return launch_xl16<256, 11, MODE, IO>(a, s);
    }
    return -2;
}

}  // namespace

// One conv of a wide ResBlock, X-resident, 16-bit operands.  io 1: x fp32 [B][C][ld] -> y = 16-bit convert(leaky_relu(conv + b))
// ([B][C][ld] halves, batch stride bstride in ELEMENTS of the respective type); io 2: x = that 16-bit tensor -> y fp32 =
// ((conv + b) + res) + (accum ? y : 0).  wf: [tap][C/16][C/32][64][8] fragments (to_fragment16).  mode 1 = bf16, 2 = fp16.
extern "C" int cmtts_launch_conv_xl16(const ConvXlArgs* ap, int mode, int io, void* stream_) {
    const ConvXlArgs& a = *ap;
    hipStream_t s = (hipStream_t)stream_;
    if (a.B <= 0 || a.T <= 0) return 0;
    if (a.dil < 1 || a.dil > 5 || (const void*)a.x == (const void*)a.y || (mode != 1 && mode != 2) || (io != 1 && io != 2) || a.cin || a.relu)
        return -2;
    if (io == 1 && (a.res || a.accum)) return -2;
    if ((long)a.C * a.ld >= (1L << 31)) return -2;
    if (mode == 1) return io == 1 ? dispatch_xl16<1, 1>(a, s) : dispatch_xl16<1, 2>(a, s);
#ifdef XL16_BF16_ONLY          // quick experimental builds (tools/)
    return -2;
#else
    return io == 1 ? dispatch_xl16<2, 1>(a, s) : dispatch_xl16<2, 2>(a, s);
#endif
}
