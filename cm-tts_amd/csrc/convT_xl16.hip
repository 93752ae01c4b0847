// convT_xl16_kernel: the HiFi-GAN upsamplers (ConvTranspose1d) with 16-bit operands (split out of resblock_pair16.hip).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "conv_loop16.h"
#include "xcd_map.h"
#ifndef XCD_MAP
#define XCD_MAP 1
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// HiFi-GAN upsampler with 16-bit operands: the stacked two-tap form of convT_xl_kernel (resblock_pair.hip) on the 16-bit pipe.
// y[co][s m + r - s/2] = b[co] + sum_ci sum_{q in {0,1}} W16[ci][co][r + s q] a16(x[ci][m - q]),  a16(x) = convert(leaky_relu(x / pre_div)):
// the x^T image [64 + 1 columns][CIN + 4] is staged once (true division, slope, convert), wave w = phase w % s of channel block w / s,
// fp32 accumulation and bias.  Part of set_precision("bf16" | "fp16") of the vocoder since round 2 (the oracle's operands16 modes
// quantise the same two operands: oracle/cmtts_oracle.py hifigan_generator).
struct ConvT16Args {
    const float* x;
    float* y;
    const void* wf;       // [2][CIN/16][s CO / 32][64][8] 16-bit fragments of the two-tap stacked weights (row = phase * CO + channel)
    const float* bias;
    long xbstride, ybstride;
    int B, CO, Ti, To, ldx, ldy, s;
    float pre_div, slope;
};

// MODE 3 ("fp16x3", round 3): `src` is the hi image with the lo image `img` elements further, `wfrag` the hi fragment set with the lo set `wset`
// fragments further; every product is three fp16 MFMAs, small terms first (resblock_pair16x3.inc).
template <int CIN, int NT, int MODE>
__device__ __forceinline__ void conv_loopT16(f32x16 (&acc)[NT], const u32x4* __restrict__ wfrag, const unsigned short* __restrict__ src,
                                             int mt, int mtiles, int lane, int img = 0, long wset = 0) {
    constexpr int RS = CIN + 4;
    constexpr int NS = MODE == 3 ? 2 : 1;
    constexpr int MM = MODE == 3 ? 2 : MODE;
    constexpr int RINGT = MODE == 3 ? 4 : RING;
    constexpr int G = CIN / 16;
    constexpr int NG = G * 2;                       // (32-channel chunk, tap, k-group)
    const int l31 = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto grp = [&](int it, int& chunk, int& tap, int& kgl) {
        chunk = it >> 2;
        tap = (it >> 1) & 1;
        kgl = it & 1;
    };
    const unsigned short* bl = src + l31 * RS + khalf * 8;      // src = row of column c + 1 (tap 0 reads x[m], tap 1 x[m - 1])
    auto load_b = [&](u32x4 (&dst)[NT][NS], int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const unsigned short* p = bl - tap * RS + chunk * 32 + kgl * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const u32x2 lo = *reinterpret_cast<const u32x2*>(p + q * img + j * 32 * RS);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(p + q * img + j * 32 * RS + 4);
                dst[j][q] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            }
    };
    u32x4 A[RINGT][NS];
    auto issue_a = [&](u32x4 (&dst)[NS], int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const u32x4* ptr = wfrag + ((long)(tap * G + 2 * chunk + kgl) * mtiles + mt) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[0]) : "v"(ptr) : "memory");
        if (NS == 2) {
            const u32x4* ptr2 = ptr + wset;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[NS - 1]) : "v"(ptr2) : "memory");
        }
    };
#pragma unroll
    for (int s = 0; s < RINGT - 1; ++s)
        if (s < NG) issue_a(A[s], s);
    u32x4 Bf[2][NT][NS];
    load_b(Bf[0], 0);
    auto body = [&](int it) {
        if (it + RINGT - 1 < NG) {
            issue_a(A[(it + RINGT - 1) % RINGT], it + RINGT - 1);
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(A[it % RINGT][0]) : "n"((RINGT - 1) * NS));
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[it % RINGT][0]));
        }
        if (NS == 2) asm volatile("" : "+v"(A[it % RINGT][NS - 1]));        // the lo fragment of the pair: same wait
        if (it + 1 < NG) load_b(Bf[(it + 1) & 1], it + 1);
        if (NS == 2) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = mma16<MM>(A[it % RINGT][NS - 1], Bf[it & 1][j][0], acc[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = mma16<MM>(A[it % RINGT][0], Bf[it & 1][j][NS - 1], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mma16<MM>(A[it % RINGT][0], Bf[it & 1][j][0], acc[j]);
        if (it + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT * NS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT * (NS == 2 ? 3 : 1), 0);
    };
    seg_loop<0, NG, 32>(body);
}

// BN = input columns per workgroup: 64; 32 for the fp16x3 C_in = 512 stage, whose two 512-channel images of 65 rows plus the output tile exceed the LDS
template <int CIN, int NW, int MODE, int BN = 64>
__global__ __launch_bounds__(64 * NW, 2) void convT_xl16_kernel(const ConvT16Args a, int mtiles) {
    constexpr int RS = CIN + 4;
    constexpr int NT = BN / 32;
    constexpr int XROWS = BN + 1;
    constexpr int NS = MODE == 3 ? 2 : 1;                                   // fp16x3: (hi, lo) images
    constexpr int MM = MODE == 3 ? 2 : MODE;
    constexpr int XIMG = (XROWS * RS + 7) & ~7;
    extern __shared__ __attribute__((aligned(16))) unsigned short xt16[];   // [NS][XROWS][RS], row j <-> m = t0 - 1 + j
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_, t0 = bx_ * BN;
    const int Ti = a.Ti;
    const float* xb = a.x + (long)b * a.xbstride;
    {
        constexpr int PAIRS = CIN / 2 / NW;
        constexpr int PB = PAIRS < 16 ? PAIRS : 16;
#pragma unroll
        for (int jb = 0; jb < (XROWS + 63) / 64; ++jb) {
            const int j = jb * 64 + lane;
            const int m = t0 - 1 + j;
            // lanes past the image's last row (all but one of the second block) re-read that row's column: same cache line, no traffic — clamped
            // to Ti - 1 only, they fetched the NEXT tile's 63 columns as well: twice the input bytes (374 -> 2xx us at C_in = 128)
            const int m_c = min(max(t0 - 1 + min(j, XROWS - 1), 0), Ti - 1);
            const bool ok = m >= 0 && m < Ti;
            for (int p0 = 0; p0 < PAIRS; p0 += PB) {
                float v[PB][2];
#pragma unroll
                for (int p = 0; p < PB; ++p)
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[p][h] = xb[(long)((w * PAIRS + p0 + p) * 2 + h) * a.ldx + m_c];
                if (j < XROWS) {
#pragma unroll
                    for (int p = 0; p < PB; ++p) {
                        float u0 = ok ? v[p][0] : 0.f, u1 = ok ? v[p][1] : 0.f;
                        if (a.pre_div != 1.0f) { u0 = u0 / a.pre_div; u1 = u1 / a.pre_div; }
                        u0 = u0 > 0.f ? u0 : u0 * a.slope;
                        u1 = u1 > 0.f ? u1 : u1 * a.slope;
                        const unsigned hi = pack16<MM>(u0, u1);
                        *reinterpret_cast<unsigned*>(xt16 + j * RS + (w * PAIRS + p0 + p) * 2) = hi;
                        if (MODE == 3) {
                            const cvt_f16x2 h = __builtin_bit_cast(cvt_f16x2, hi);
                            *reinterpret_cast<unsigned*>(xt16 + XIMG + j * RS + (w * PAIRS + p0 + p) * 2) = pack16<2>(u0 - (float)h[0], u1 - (float)h[1]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const int S = a.s;
    const int phase = w % S, cbl = w / S;
    const int per = NW / S;
    const int passes = (a.CO / 32) / per / gridDim.z;
    const int pd = S / 2;
    float* yb = a.y + (long)b * a.ybstride;
    // Output through LDS: a wave's accumulator tile is ONE stride phase of 32 output columns — written directly, every 4-byte store lands S * 4
    // bytes from its neighbour's and each 128-byte line of y is completed by S different waves at different times (the 16-bit kernel spent most
    // of its time there: 546 us for 0.67 GB at C_in = 256).  The waves of a workgroup hold all S phases of the same columns, so they meet in an
    // LDS tile [rows][phase][PS] (phase-major: conflict-free writes) and the workgroup stores whole rows, consecutive lanes = consecutive samples
    // (stride 8: 546 -> 371 us at C_in = 256, 224 -> 182 at 512; bitwise the same values).
    constexpr int PS = 40;                               // phase stride in floats (32 columns + 8: the interleaving reads are 2-way at worst)
    float* ot = reinterpret_cast<float*>(xt16 + NS * XIMG);       // [per * 32][S][PS]
    const int lgS = __ffs(S) - 1;
    const int rowlen = S * 32;                           // output samples per row per n-tile
    for (int ps = 0; ps < passes; ++ps) {
        const int cb0 = (blockIdx.z * passes + ps) * per;
        const int cb = cb0 + cbl;
        const int mt = phase * (a.CO / 32) + cb;
        f32x16 acc[NT];
        conv_loopT16<CIN, NT, MODE>(acc, (const u32x4*)a.wf, xt16 + RS, mt, mtiles, lane, XIMG, (long)2 * (CIN / 16) * mtiles * 64);
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.bias[cb * 32 + acc_row(r, lane)];
        if (S < 4) {      // two phases: neighbouring stores are 8 bytes apart and the direct form is (slightly) faster: 374 / 326 vs 389 / 334 us
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb * 32 + acc_row(r, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = t0 + j * 32 + l31;
                    const int t = n * S + phase - pd;
                    if (n <= Ti && t >= 0 && t < a.To) yb[(long)co * a.ldy + t] = acc[j][r] + bi[r];
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            __syncthreads();                             // the tile is free (previous n-tile stored)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[((cbl * 32 + acc_row(r, lane)) * S + phase) * PS + l31] = acc[j][r] + bi[r];
            __syncthreads();
            const int tb = (t0 + j * 32) * S - pd;       // time of the tile's first sample
#pragma unroll
            for (int q = 0; q < 16; ++q) {               // per * 32 rows x S * 32 samples = 16 per thread
                const int idx = q * (64 * NW) + tid;
                const int row = idx >> (lgS + 5), c = idx & (rowlen - 1);
                const int m = c >> lgS, ph = c & (S - 1);
                const int t = tb + c;
                const float v = ot[(row * S + ph) * PS + m];
                if (t0 + j * 32 + m <= Ti && t >= 0 && t < a.To) yb[(long)(cb0 * 32 + row) * a.ldy + t] = v;
            }
        }
    }
}

template <int CIN, int NW, int MODE, int BN = 64>
int launch_convT16(const ConvT16Args& a, hipStream_t stream) {
    if (a.s & (a.s - 1)) return -2;                    // the output tile's interleave uses shifts
    const size_t lds = (MODE == 3 ? 2 : 1) * (((size_t)(BN + 1) * (CIN + 4) + 7) & ~(size_t)7) * sizeof(unsigned short) + (a.s >= 4 ? (size_t)(NW / a.s) * 32 * a.s * 40 * sizeof(float) : 0);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(convT_xl16_kernel<CIN, NW, MODE, BN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -3;
        attr_lds = lds;
    }
    const int mtiles = a.s * a.CO / 32;
    if (NW % a.s || mtiles % NW) return -2;
    const int npass = mtiles / NW;
    const long tiles = (long)((a.Ti + 1 + BN - 1) / BN) * a.B;
    int zs = 1;
    // channel-block split of the grid: only until every CU has two workgroups — each split stages the x tile again, and staging is what this
    // kernel waits for (C_in = 256: 336 / 402 / 500 us with 1 / 2 / 4 splits; C_in = 512, 288 tiles: 180 / 158 / 160 / 184 with 1 / 2 / 4 / 8)
    while (tiles * zs < 512 && zs * 2 <= npass && npass % (zs * 2) == 0) zs *= 2;
    {   // CMTTS_CONVT_ZS=<n>: channel-block split of the grid, for experiments (tools/)
        static const char* e = getenv("CMTTS_CONVT_ZS");
        if (e && atoi(e) > 0 && npass % atoi(e) == 0) zs = atoi(e);
    }
    dim3 grid((a.Ti + 1 + BN - 1) / BN, a.B, zs);
    hipLaunchKernelGGL((convT_xl16_kernel<CIN, NW, MODE, BN>), grid, dim3(64 * NW), lds, stream, a, mtiles);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE>
int dispatch_convT16(const ConvT16Args& a, int cin, hipStream_t s) {
    const int mtiles = a.s * a.CO / 32;
    if (cin == 512 && mtiles >= 8) {
        if constexpr (MODE == 3) return launch_convT16<512, 8, MODE, 32>(a, s);       // 32-column tiles: two 512-channel images + the output tile fit (109 KB)
        else return launch_convT16<512, 8, MODE>(a, s);
    }
    if (cin == 256 && mtiles >= 8) return launch_convT16<256, 8, MODE>(a, s);
    if (cin == 128 && mtiles >= 4) return launch_convT16<128, 4, MODE>(a, s);
    if (cin == 64 && mtiles >= 2) return launch_convT16<64, 2, MODE>(a, s);
    return -2;
}

}  // namespace

// HiFi-GAN upsampler with 16-bit operands (convT_xl16_kernel): arguments as cmtts_launch_convT, wf16 = to_fragment16 of the two-tap
// stacked weights ([2][cin/16][s co / 32][64][8]), mode 1 = bf16, 2 = fp16, 3 = fp16x3 (the hi fragment set followed by the lo set).  0 = launched, -2 = shape not covered, -3 = HIP error.
extern "C" int cmtts_launch_convT16(const float* x, float* y, const void* wf16, const float* bias, long xbstride, long ybstride, int B,
                                    int cin, int co, int Ti, int To, int ldx, int ldy, int s, float pre_div, float slope, int mode,
                                    void* stream_) {
    if (B <= 0 || Ti <= 0) return 0;
    if (!wf16 || (s * co) % 32 || To != Ti * s || s < 2 || (s & 1) || mode < 1 || mode > 3) return -2;
    ConvT16Args a{x, y, wf16, bias, xbstride, ybstride, B, co, Ti, To, ldx, ldy, s, pre_div, slope};
    if (mode == 3) return dispatch_convT16<3>(a, cin, (hipStream_t)stream_);
    return mode == 1 ? dispatch_convT16<1>(a, cin, (hipStream_t)stream_) : dispatch_convT16<2>(a, cin, (hipStream_t)stream_);
}
