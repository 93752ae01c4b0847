// HiFi-GAN ResBlock pair with 16-bit MFMA operands (bf16 / fp16, fp32 accumulate) for the narrow stages (C = 64, 32):
//   xt = conv1(leaky_relu(x));  y = conv2(leaky_relu(xt)) + x      (hifigan/models.py:96-103)
// in ONE launch — the 16-bit twin of resblock_pair.hip.  At 16x the fp32 MFMA rate these convs are bound by bytes, not
// by the matrix pipe: the two-launch path moves 5 tensor passes per pair (read x, write xt16, read xt16, read x, write y);
// here x is staged once (converted while staged, as conv_mfma16.hip does), xt stays on chip as the 16-bit activated image
// conv2 would have read from HBM, the residual re-read of x is served by L2 (this workgroup fetched the same lines a few
// microseconds earlier) and y is written once.
//
//   * LDS: x^T tile [256 + 2*r1][C] and xt^T tile [256 + k - 1][C], 16-bit, row stride C + 4 halves: a B fragment (lane l:
//     8 consecutive channels of column l & 31) is two conflict-free ds_read_b64 and a dilated tap is a row offset;
//   * every wave owns ONE 32-row m-tile x NT 32-column n-tiles (C = 64: 1 x 4, C = 32: 1 x 2); weights stream L2 -> VGPR in
//     MFMA A-fragment order, one global_load_dwordx4 per k-group of 16 channels feeding NT MFMAs, 4-deep ring;
//   * same conversions (v_cvt_pk_{bf16,f16}_f32 of leaky_relu(.)), same (32-channel chunk, tap, k-group) accumulation
//     order and the same epilogue expressions as conv_mfma16.hip => BITWISE equal to the two-launch 16-bit path.
#include <hip/hip_runtime.h>
#include "resblock_pair.h"
#include "xcd_map.h"
#ifndef XCD_MAP
#define XCD_MAP 1
#endif
#include "cvt16.h"

#include "conv_loop16.h"

namespace {


template <int C, int KT, int MODE>
__global__ __launch_bounds__(C == 128 ? 512 : 256, 2) void resblock_pair16_kernel(const PairArgs a) {
    constexpr int RS = C + CL16_PAD;
    constexpr int NWAVES = C == 128 ? 8 : 4;                // C = 128 (round 2): 8 waves, one 151-KB workgroup per CU
    constexpr int NT = (C / 32) * (N1 / 32) / NWAVES;       // 32x32 tiles per wave: one m-tile x NT n-tiles
    constexpr int WPM = NWAVES / (C / 32);                  // waves per m-tile
    constexpr int R2 = (KT - 1) / 2;
    constexpr int TT = N1 - 2 * R2;
    constexpr int XROWS = N1 + 2 * R1MAX;                   // the xt^T tile has N1 + KT - 1 rows
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* Xs = smem16;                            // [XROWS][RS]  convert(leaky(x)),  row j <-> t = t0 - R2 - r1 + j
    unsigned short* XTs = smem16 + XROWS * RS;              // [XTROWS][RS] convert(leaky(xt)), row c <-> t = t0 - R2 + c
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = w / WPM, nq = w % WPM;
    const int l31 = lane & 31;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * TT;
    const int T = a.T, dil = a.dil;
    const int r1 = dil * R2;
    const int xw = N1 + 2 * r1;
    const float* xb = a.x + (long)b * a.bstride;
    const float slope = a.slope;

    // ---- stage x^T: wave w converts channel pairs w*(C/8) .. of every column; lanes run over columns (coalesced rows)
    {
        const int tbase = t0 - R2 - r1;
        constexpr int PAIRS = C / 2 / NWAVES;               // channel pairs per wave: 8 / 4
        constexpr int XBLK = (XROWS + 63) / 64;             // 5 column blocks
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const int j = jb * 64 + lane;
            const int t = tbase + j;
            const int t_c = min(max(t, 0), T - 1);
            const bool in = t >= 0 && t < T;
            const float fpos = in ? 1.f : 0.f, fneg = in ? slope : 0.f;     // conv_mfma16.hip's staging arithmetic
            float v[PAIRS][2];
#pragma unroll
            for (int p = 0; p < PAIRS; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) v[p][h] = xb[(long)((w * PAIRS + p) * 2 + h) * a.ld + t_c];
            if (j < xw) {
#pragma unroll
                for (int p = 0; p < PAIRS; ++p) {
                    const float v0 = v[p][0], v1 = v[p][1];
                    const unsigned pk = pack16<MODE>(v0 * (v0 > 0.f ? fpos : fneg), v1 * (v1 > 0.f ? fpos : fneg));
                    *reinterpret_cast<unsigned*>(Xs + j * RS + (w * PAIRS + p) * 2) = pk;
                }
            }
        }
    }
    __syncthreads();

    f32x16 acc[NT];
    const int col0 = nq * (NT * 32);
    conv_loop16<C, KT, NT, MODE>(acc, (const u32x4*)a.w1f, Xs, dil, mt, col0, lane);
    {   // xt = acc + b1 -> convert(leaky(xt)) (what conv2's staging reads from HBM on the two-launch path), zero outside [0, T)
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.b1[mt * 32 + acc_row(r, lane)];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = col0 + j * 32 + l31;
            const int t = t0 - R2 + c;
            const bool in = t >= 0 && t < T;
            if constexpr (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[j][4 * q + e] + bi[4 * q + e];
                        v[e] = v[e] * (v[e] > 0.f ? 1.f : slope);
                    }
                    const u32x2 pk = pack16x4<MODE>(v);
                    *reinterpret_cast<u32x2*>(XTs + c * RS + mt * 32 + acc_row(4 * q, lane)) = in ? pk : (u32x2){0u, 0u};
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[j][r] + bi[r];
                    v = v * (v > 0.f ? 1.f : slope);
                    XTs[c * RS + mt * 32 + acc_row(r, lane)] = in ? (unsigned short)pack16<MODE>(v, 0.f) : (unsigned short)0;
                }
            }
        }
    }
    __syncthreads();

    conv_loop16<C, KT, NT, MODE>(acc, (const u32x4*)a.w2f, XTs, 1, mt, col0, lane);
    {   // ((acc + b2) + x) + y_old: conv_mfma16.hip's epilogue; all loads of a 32x32 tile before its stores
        float* yb = a.y + (long)b * a.bstride;
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.b2[mt * 32 + acc_row(r, lane)];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int o = col0 + j * 32 + l31;
            const int t = t0 + o;
            const bool ok = o < TT && t < T;
            const int t_c = min(t, T - 1);
            float rv[16], yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long off = (long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c;
                rv[r] = xb[off];
                yv[r] = a.accum ? yb[off] : 0.f;
            }
            if (ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t] = ((acc[j][r] + bi[r]) + rv[r]) + yv[r];
            }
        }
    }
}

template <int C, int KT, int MODE>
int launch_pair16(const PairArgs& a, hipStream_t stream) {
    constexpr int TT = N1 - (KT - 1);
    const size_t lds = (size_t)(N1 + 2 * R1MAX + N1 + KT - 1) * (C + CL16_PAD) * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_pair16_kernel<C, KT, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.T + TT - 1) / TT, a.B);
    hipLaunchKernelGGL((resblock_pair16_kernel<C, KT, MODE>), grid, dim3(C == 128 ? 512 : 256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE>
int dispatch16(const PairArgs& a, hipStream_t s) {
    if (a.C == 128) {
        if (a.k == 3) return launch_pair16<128, 3, MODE>(a, s);
        if (a.k == 7) return launch_pair16<128, 7, MODE>(a, s);
        if (a.k == 11) return launch_pair16<128, 11, MODE>(a, s);
    }
    if (a.C == 64) {
        if (a.k == 3) return launch_pair16<64, 3, MODE>(a, s);
        if (a.k == 7) return launch_pair16<64, 7, MODE>(a, s);
        if (a.k == 11) return launch_pair16<64, 11, MODE>(a, s);
    } else if (a.C == 32) {
        if (a.k == 3) return launch_pair16<32, 3, MODE>(a, s);
        if (a.k == 7) return launch_pair16<32, 7, MODE>(a, s);
        if (a.k == 11) return launch_pair16<32, 11, MODE>(a, s);
    }
    return -2;
}

}  // namespace

// w1f / w2f: the 16-bit fragment-order weights conv_mfma16.hip uses ([tap][C/16][C/32][64][8]).  mode 1 = bf16, 2 = fp16.
extern "C" int cmtts_launch_resblock_pair16(const PairArgs* ap, int mode, void* stream_) {
    const PairArgs& a = *ap;
    hipStream_t s = (hipStream_t)stream_;
    if (a.B <= 0 || a.T <= 0) return 0;
    if (a.dil * (a.k - 1) / 2 > R1MAX || a.x == a.y || (mode != 1 && mode != 2)) return -2;
    return mode == 1 ? dispatch16<1>(a, s) : dispatch16<2>(a, s);
}
