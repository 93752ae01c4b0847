// resblock16_kernel: a WHOLE HiFi-GAN ResBlock of a narrow stage (C = 64, 32) with 16-bit operands in one launch (split out of
// resblock_pair16.hip: three translation units compile in parallel instead of one for a quarter of an hour).
#include <hip/hip_runtime.h>
#include "conv_loop16.h"
#include "xcd_map.h"
#ifndef XCD_MAP
#define XCD_MAP 1
#endif

#ifndef RB16_W
#define RB16_W 384
#endif
#ifndef RB16_OCC
#define RB16_OCC 2
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// A WHOLE ResBlock (hifigan/models.py:84-109: three [leaky -> conv(k, d) -> leaky -> conv(k, 1) -> + x] iterations, d = 1, 3, 5)
// of the narrow stages (C = 64, 32) with 16-bit operands in ONE launch: x is read once (plus the halo), the MRF sum is written
// / accumulated once — 2 tensor passes where the three pair launches move 6 (and the layer-granular path 15).  The pairs sit
// on the HBM floor (profiles/r02_vocoder_bf16.md), so bytes are what is left to remove.
//
//   * tile: 384 columns = NOUT outputs + 2 H of halo, H = 12 (k - 1) / 2 (the sum of the six convs' reaches); every conv is
//     evaluated on all 384 columns (edge columns of a conv's output that lie beyond its valid range are computed from stale
//     neighbours and are never read by a valid column downstream: the halo accounting guarantees it);
//   * LDS: the 16-bit activated images of x and xt, [384 + 2 x 25 margin rows][C + 4], common row origin; conv1 of a pair reads
//     the x image and writes the xt image, conv2 reads xt and writes leaky(x_new) back into the x image (x is dead by then);
//   * the fp32 residual stream of a wave's own columns (one m-tile x NT n-tiles, fixed for all six convs) lives in REGISTERS in
//     the accumulator layout: x_new = (acc + b2) + x_res replaces it pair after pair, and is what the last pair stores;
//   * K loops = conv_loop16 (hand-issued weight ring), six per tile, a barrier after each.
// Element by element the arithmetic is the pair kernel's (conversions, accumulation order, epilogue expressions, zero outside
// [0, T)) => bitwise equal to three pair launches (tests/test_gpu_parity.py::test_vocoder_pair16_kernel_bitwise).
// -DRB16_STAMP (tools/rb16_exp.sh, tools/rb16_phases.py): every wave stamps the cycle counter at its phase boundaries; results unchanged
#ifndef RB16_STAMP
#define RB16_STAMP 0
#endif
long long* g_rb16_dbg = nullptr;

struct Rb16Args {
    const float* x;       // [B][C][ld]
    float* y;             // [B][C][ld] MRF sum (y += result when accum)
    const void* w1f[3];
    const void* w2f[3];
    const float* b1[3];
    const float* b2[3];
    long bstride;
    int B, C, T, ld;
    int accum;
    float slope;
    long long* dbg;       // RB16_STAMP builds: [grid][waves][16] cycle stamps
};

template <int C, int KT, int MODE>
__global__ __launch_bounds__(C * 8, RB16_OCC) void resblock16_kernel(const Rb16Args a) {      // C = 64: 8 waves (two per SIMD: one 118-KB workgroup per CU)
    constexpr int RS = C + CL16_PAD;
    constexpr int R = (KT - 1) / 2;
    constexpr int H = 12 * R;                               // (1 + 3 + 5) R for the dilated convs + 3 R for the plain ones
    // columns every conv is evaluated on (C = 64 with 192 columns, 4 waves and two workgroups per CU — one's epilogues under the other's K loops — was
    // tried: k = 7 1365 -> 1511 us, k = 3 739 -> 704: the recomputed halo columns cost more than the overlap gives)
    constexpr int W = RB16_W;
    constexpr int NOUT = W - 2 * H;                         // 264 / 312 / 360
    constexpr int MARGIN = 5 * R;                           // largest reach of one conv: rows a conv may read beyond the tile
    constexpr int ROWS = W + 2 * MARGIN;
    constexpr int NWAVES = C / 8;                           // 4 (C = 32) / 8 (C = 64)
    constexpr int NTHR = 64 * NWAVES;
    constexpr int WPM = 4;                                  // waves per m-tile
    constexpr int NT = (W / 32) / WPM;                      // 3 n-tiles per wave
    static_assert(NT * WPM * 32 == W, "tile split");
    extern __shared__ __attribute__((aligned(16))) unsigned short rb16[];
    unsigned short* Xs = rb16;                              // [ROWS][RS] convert(leaky(x)),  row MARGIN + c <-> t = tb + c
    unsigned short* XTs = rb16 + ROWS * RS;                 // [ROWS][RS] convert(leaky(xt))
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = w / WPM, nq = w % WPM;
    const int l31 = lane & 31;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * NOUT;
    const int tb = t0 - H;                                  // time of tile column 0
    const int T = a.T;
    const float slope = a.slope;
    const float* xb = a.x + (long)b * a.bstride;
    const int col0 = nq * (NT * 32);
    auto stamp = [&](int slot) {
        if (RB16_STAMP && a.dbg && lane == 0)
            a.dbg[(((long)b * gridDim.x + blockIdx.x) * NWAVES + w) * 16 + slot] = (long long)__builtin_readcyclecounter();
    };
    stamp(0);

    {   // stage convert(leaky(x)) for rows 0 .. ROWS-1 (t = tb - MARGIN + row): the margins hold real neighbours of the tile
        constexpr int PAIRS = C / 2 / NWAVES;               // channel pairs per wave: 4
        constexpr int XBLK = (ROWS + 63) / 64;
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const int j = jb * 64 + lane;
            const int t = tb - MARGIN + j;
            const int t_c = min(max(t, 0), T - 1);
            const bool in = t >= 0 && t < T;
            const float fpos = in ? 1.f : 0.f, fneg = in ? slope : 0.f;
            float v[PAIRS][2];
#pragma unroll
            for (int p = 0; p < PAIRS; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) v[p][h] = xb[(long)((w * PAIRS + p) * 2 + h) * a.ld + t_c];
            if (j < ROWS) {
#pragma unroll
                for (int p = 0; p < PAIRS; ++p) {
                    const float v0 = v[p][0], v1 = v[p][1];
                    *reinterpret_cast<unsigned*>(Xs + j * RS + (w * PAIRS + p) * 2) =
                        pack16<MODE>(v0 * (v0 > 0.f ? fpos : fneg), v1 * (v1 > 0.f ? fpos : fneg));
                }
            }
        }
        // the xt image's margin rows are read by the edge columns of conv2 and never written: clear them once
        for (int i = tid; i < 2 * MARGIN * (RS / 2); i += NTHR) {
            const int r = i / (RS / 2), c2 = i - r * (RS / 2);
            const int row = r < MARGIN ? r : W + r;          // rows 0 .. MARGIN-1 and MARGIN + W .. ROWS-1
            *reinterpret_cast<unsigned*>(XTs + row * RS + 2 * c2) = 0u;
        }
    }
    // fp32 residual stream of this wave's own tiles, accumulator layout (0 outside [0, T): never stored, never read as data)
    float res[NT][16];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = tb + col0 + j * 32 + l31;
        const int t_c = min(max(t, 0), T - 1);
        const bool in = t >= 0 && t < T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = xb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c];
            res[j][r] = in ? v : 0.f;
        }
    }
    stamp(1);
    __syncthreads();
    stamp(2);

    f32x16 acc[NT];
    constexpr int DIL[3] = {1, 3, 5};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        // conv1 (dilation d): output column c reads x rows (MARGIN + c) + (tap - R) d
        conv_loop16<C, KT, NT, MODE>(acc, (const u32x4*)a.w1f[p], Xs + (MARGIN - R * DIL[p]) * RS, DIL[p], mt, col0, lane);
        stamp(3 + 4 * p);
        {
            float bi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bi[r] = a.b1[p][mt * 32 + acc_row(r, lane)];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c = col0 + j * 32 + l31;
                const int t = tb + c;
                const bool in = t >= 0 && t < T;
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {     // (16 two-byte stores per tile made the epilogues, not the K loops, the longest phase of this kernel)
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[j][4 * q + e] + bi[4 * q + e];
                            v[e] = v[e] * (v[e] > 0.f ? 1.f : slope);
                        }
                        const u32x2 pk = pack16x4<MODE>(v);
                        *reinterpret_cast<u32x2*>(XTs + (MARGIN + c) * RS + mt * 32 + acc_row(4 * q, lane)) = in ? pk : (u32x2){0u, 0u};
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[j][r] + bi[r];
                        v = v * (v > 0.f ? 1.f : slope);
                        XTs[(MARGIN + c) * RS + mt * 32 + acc_row(r, lane)] = in ? (unsigned short)pack16<MODE>(v, 0.f) : (unsigned short)0;
                    }
                }
            }
        }
        __syncthreads();
        stamp(4 + 4 * p);
        // conv2 (dilation 1): output column c reads xt rows (MARGIN + c) + (tap - R)
        conv_loop16<C, KT, NT, MODE>(acc, (const u32x4*)a.w2f[p], XTs + (MARGIN - R) * RS, 1, mt, col0, lane);
        stamp(5 + 4 * p);
        {
            float bi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bi[r] = a.b2[p][mt * 32 + acc_row(r, lane)];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c = col0 + j * 32 + l31;
                const int t = tb + c;
                const bool in = t >= 0 && t < T;
                // the next pair's conv1 operand: what its staging would have made of x_new.  The product must be ROUNDED TO fp32 before the
                // conversion, as in the staging pass of a pair launch (x_new went through HBM there): left visible, the compiler fuses multiply +
                // convert into one v_fma_mixlo_f16 with a single rounding — rare 1-ulp fp16 differences that spread downstream
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float u[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * q + e;
                            const float xn = (acc[j][r] + bi[r]) + res[j][r];      // the pair kernel's (acc + b2) + x
                            res[j][r] = in ? xn : 0.f;
                            u[e] = xn * (xn > 0.f ? 1.f : slope);
                            asm volatile("" : "+v"(u[e]));
                        }
                        if (p < 2) {
                            const u32x2 pk = pack16x4<MODE>(u);
                            *reinterpret_cast<u32x2*>(Xs + (MARGIN + c) * RS + mt * 32 + acc_row(4 * q, lane)) = in ? pk : (u32x2){0u, 0u};
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float xn = (acc[j][r] + bi[r]) + res[j][r];
                        res[j][r] = in ? xn : 0.f;
                        if (p < 2) {
                            float u = xn * (xn > 0.f ? 1.f : slope);
                            asm volatile("" : "+v"(u));
                            Xs[(MARGIN + c) * RS + mt * 32 + acc_row(r, lane)] = in ? (unsigned short)pack16<MODE>(u, 0.f) : (unsigned short)0;
                        }
                    }
                }
            }
        }
        if (p < 2) __syncthreads();
        stamp(6 + 4 * p);
    }
    // the block's output on the NOUT central columns: y (+)= x_3; the old sum of all tiles is requested before the first store
    float* yb = a.y + (long)b * a.bstride;
    if (a.accum) {
        float yv[NT][16];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t_c = min(max(tb + col0 + j * 32 + l31, 0), T - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[j][r] = yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = col0 + j * 32 + l31;
            const int t = tb + c;
            if (c >= H && c < H + NOUT && t < T) {     // t >= 0 follows from c >= H
#pragma unroll
                for (int r = 0; r < 16; ++r) yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t] = res[j][r] + yv[j][r];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = col0 + j * 32 + l31;
            const int t = tb + c;
            if (c >= H && c < H + NOUT && t < T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t] = res[j][r] + 0.f;
            }
        }
    }
    stamp(15);
}

template <int C, int KT, int MODE>
int launch_rb16(const Rb16Args& a, hipStream_t stream) {
    constexpr int R = (KT - 1) / 2;
    constexpr int NOUT = RB16_W - 24 * R;
    const size_t lds = (size_t)2 * (RB16_W + 10 * R) * (C + CL16_PAD) * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resblock16_kernel<C, KT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.T + NOUT - 1) / NOUT, a.B);
    Rb16Args c = a;
    c.dbg = g_rb16_dbg;
    hipLaunchKernelGGL((resblock16_kernel<C, KT, MODE>), grid, dim3(C * 8), lds, stream, c);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE>
int dispatch_rb16(const Rb16Args& a, int k, hipStream_t s) {
#ifdef RB16_DEV               // quick experimental builds: ONE bf16 instance (-DRB16_DEV_C=64 -DRB16_DEV_K=7), everything else returns -2
    if (MODE == 1 && a.C == RB16_DEV_C && k == RB16_DEV_K) return launch_rb16<RB16_DEV_C, RB16_DEV_K, 1>(a, s);
    return -2;
#endif
    if (a.C == 64) {
        if (k == 3) return launch_rb16<64, 3, MODE>(a, s);
        if (k == 7) return launch_rb16<64, 7, MODE>(a, s);
        if (k == 11) return launch_rb16<64, 11, MODE>(a, s);
    } else if (a.C == 32) {
        if (k == 3) return launch_rb16<32, 3, MODE>(a, s);
        if (k == 7) return launch_rb16<32, 7, MODE>(a, s);
        if (k == 11) return launch_rb16<32, 11, MODE>(a, s);
    }
    return -2;
}

}  // namespace

extern "C" void cmtts_rb16_set_debug(long long* dbg) { g_rb16_dbg = dbg; }

// A whole ResBlock (three pairs, dilations 1, 3, 5, kernel k) of a narrow stage in one launch, 16-bit operands (resblock16_kernel).
// w1f / w2f / b1 / b2: the three pairs' conv1 / conv2 fragments ([tap][C/16][C/32][64][8]) and biases.  x must not alias y.
extern "C" int cmtts_launch_resblock16(const float* x, float* y, const void* const* w1f, const void* const* w2f, const float* const* b1,
                                       const float* const* b2, long bstride, int B, int C, int T, int ld, int k, int accum, float slope,
                                       int mode, void* stream_) {
    if (B <= 0 || T <= 0) return 0;
    if ((mode != 1 && mode != 2) || x == y) return -2;
    Rb16Args a;
    a.x = x; a.y = y;
    for (int p = 0; p < 3; ++p) { a.w1f[p] = w1f[p]; a.w2f[p] = w2f[p]; a.b1[p] = b1[p]; a.b2[p] = b2[p]; }
    a.bstride = bstride; a.B = B; a.C = C; a.T = T; a.ld = ld; a.accum = accum; a.slope = slope; a.dbg = nullptr;
    return mode == 1 ? dispatch_rb16<1>(a, k, (hipStream_t)stream_) : dispatch_rb16<2>(a, k, (hipStream_t)stream_);
}
