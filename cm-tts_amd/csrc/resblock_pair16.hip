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
#include "cvt16.h"

#include "conv_loop16.h"

namespace {

// bf16 epilogues store FOUR channels per LDS instruction (accumulator registers 4q .. 4q+3 = four consecutive channels of a column = 8 contiguous bytes
// of the [column][channel] image): two packed converts + one ds_write_b64 instead of four converts + four ds_write_b16.  fp16 keeps the per-value
// form: there the compiler fuses `leaky multiply -> convert` into a single-rounding v_fma_mixlo_f16 on the two-launch path, and no packed spelling
// tried reproduced its bits (3e-4 on the wav; caught by test_vocoder_pair16_kernel_bitwise).
template <int MODE>
__device__ __forceinline__ u32x2 pack16x4(const float (&v)[4]) {
    return (u32x2){pack16<MODE>(v[0], v[1]), pack16<MODE>(v[2], v[3])};
}

template <int C, int KT, int MODE>
__global__ __launch_bounds__(C == 128 ? 512 : 256, 2) void resblock_pair16_kernel(const PairArgs a) {
    constexpr int RS = C + 4;
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
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TT;
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

// ---------------------------------------------------------------------------------------------------------------------
// Persistent form with REGISTER-RESIDENT weights.  The kernel above streams both convs' weights per 246-column tile (C = 64,
// k = 11: 2 x 90 KB per tile and per m-tile-sharing wave): at 16-bit MFMA rates that stream, not the tensor passes, sets
// the time of the k >= 7 pairs (0.83 ms per k = 11 pair vs 0.64 for the two-launch path).  Here a workgroup of 4 waves — one
// per SIMD, the whole 512-entry register file each — loads its A fragments of BOTH convs once (one m-tile x all of K:
// 2 * (C/16) * k fragments = 88 ... 352 registers) and then walks over a contiguous range of tiles:
//
//   * steady state touches HBM/L2 only for activations: the next tile's raw x arrives by LDS-DMA (global_load_lds_dword,
//     no registers) in a double-buffered fp32 staging area while the current tile is multiplied; a convert pass
//     (LeakyReLU, zero padding, v_cvt_pk, transpose) turns it into the 16-bit x^T image the MFMAs read;
//   * three raw s_barriers per tile (lgkmcnt only: a __syncthreads would drain the DMA in flight), one vmcnt(0) per tile
//     just before the barrier that publishes the landed DMA;
//   * fully unrolled K loops with compile-time fragment indices, B fragments one group ahead.
// Same conversions, accumulation order and epilogue as above => bitwise equal to it and to the two-launch path.
template <int C> struct P16 {
    static constexpr int N1 = C == 64 ? 128 : 256;         // 8 accumulator tiles per workgroup either way
    static constexpr int NT = 2;
    static constexpr int XROWS = N1 + 2 * R1MAX;
    static constexpr int XWF = (XROWS + 63) / 64 * 64;     // fp32 staging row stride: whole 64-lane DMA blocks
};

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int C, int KT, int MODE>
__global__ __launch_bounds__(256, 1) void resblock_pair16p_kernel(const PairArgs a, int tiles_per_utt, int total_tiles) {
    using P = P16<C>;
    constexpr int N1 = P::N1, NT = P::NT, XROWS = P::XROWS, XWF = P::XWF;
    constexpr int RS = C + 4;
    constexpr int G = C / 16, MTn = C / 32, NG = G * KT;
    constexpr int WPM = 4 / MTn;
    constexpr int R2 = (KT - 1) / 2;
    constexpr int TT = N1 - 2 * R2;
    constexpr int NBLK = XWF / 64;
    constexpr bool RES_PREFETCH = 2 * NG * 4 <= 256;        // C = 64, k = 11 keeps 352 registers of weights: residual loaded late
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    float* Xf = reinterpret_cast<float*>(smem_p);                                   // [2][C][XWF] raw x (LDS-DMA target)
    unsigned short* Xs = reinterpret_cast<unsigned short*>(smem_p + (size_t)2 * C * XWF * 4);   // [XROWS][RS]
    unsigned short* XTs = Xs + XROWS * RS;                                           // [XTROWS][RS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = w / WPM, nq = w % WPM;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int T = a.T, dil = a.dil;
    const int r1 = dil * R2;
    const int xw = N1 + 2 * r1;
    const float slope = a.slope;
    const int col0 = nq * (NT * 32);

    // ---- this workgroup's contiguous tile range
    const int G_ = gridDim.x, g = blockIdx.x;
    const int tile_lo = (int)((long)total_tiles * g / G_), tile_hi = (int)((long)total_tiles * (g + 1) / G_);
    if (tile_lo >= tile_hi) return;

    // ---- resident weights: A fragments of this wave's m-tile for every (chunk, tap, k-group) of both convs
    u32x4 A1[NG], A2[NG];
    {
        const u32x4* w1 = (const u32x4*)a.w1f;
        const u32x4* w2 = (const u32x4*)a.w2f;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int chunk = it / (2 * KT), rr = it - chunk * (2 * KT), tap = rr >> 1, kgl = rr & 1;
            const long idx = ((long)(tap * G + 2 * chunk + kgl) * MTn + mt) * 64 + lane;
            A1[it] = w1[idx];
            A2[it] = w2[idx];
        }
    }
    float b1r[16], b2r[16];         // biases: resident unless the weights leave no room (then re-read per tile: L1/L2 hits)
    if (RES_PREFETCH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { b1r[r] = a.b1[mt * 32 + acc_row(r, lane)]; b2r[r] = a.b2[mt * 32 + acc_row(r, lane)]; }
    }

    // LDS-DMA of one tile's raw x rows into staging buffer `buf`: wave w moves rows w, w+4, ...; a wave-instruction moves 64
    // consecutive columns of a row (clamped addresses: out-of-sequence columns are zeroed by the convert pass)
    auto dma_tile = [&](int tile, int buf) {
        const int b = tile / tiles_per_utt, ti = tile - b * tiles_per_utt;
        const int tbase = ti * TT - R2 - r1;
        const float* xb = a.x + (long)b * a.bstride;
#pragma unroll 1
        for (int c = w; c < C; c += 4) {
            const float* xrow = xb + (long)c * a.ld;
#pragma unroll
            for (int jb = 0; jb < NBLK; ++jb) {
                const int t_c = min(max(tbase + jb * 64 + lane, 0), T - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xrow + t_c),
                                                 (__attribute__((address_space(3))) void*)(Xf + ((size_t)buf * C + c) * XWF + jb * 64), 4, 0, 0);
            }
        }
    };
    auto load_b = [&](u32x4 (&dst)[NT], const unsigned short* src, int it, int dl) {
        const int chunk = it / (2 * KT), rr = it - chunk * (2 * KT), tap = rr >> 1, kgl = rr & 1;
        const unsigned short* p = src + (col0 + l31 + tap * dl) * RS + khalf * 8 + chunk * 32 + kgl * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(p + j * 32 * RS);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(p + j * 32 * RS + 4);
            dst[j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };

    dma_tile(tile_lo, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int buf = (tile - tile_lo) & 1;
        const int b = tile / tiles_per_utt, ti = tile - b * tiles_per_utt;
        const int t0 = ti * TT;
        const float* xb = a.x + (long)b * a.bstride;
        float* yb = a.y + (long)b * a.bstride;
        if (tile + 1 < tile_hi) dma_tile(tile + 1, buf ^ 1);          // lands while this tile is multiplied

        // ---- convert pass: staging (fp32, [C][XWF]) -> x^T image (16-bit, [col][C]), LeakyReLU + zero padding applied
        {
            const float* xf = Xf + (size_t)buf * C * XWF;
            const int tbase = t0 - R2 - r1;
#pragma unroll 1
            for (int p = w; p < C / 2; p += 4) {
#pragma unroll
                for (int jb = 0; jb < NBLK; ++jb) {
                    const int j = jb * 64 + lane;
                    const int t = tbase + j;
                    const bool in = t >= 0 && t < T;
                    const float fpos = in ? 1.f : 0.f, fneg = in ? slope : 0.f;
                    const float v0 = xf[(2 * p) * XWF + j], v1 = xf[(2 * p + 1) * XWF + j];
                    const unsigned pk = pack16<MODE>(v0 * (v0 > 0.f ? fpos : fneg), v1 * (v1 > 0.f ? fpos : fneg));
                    if (j < xw) *reinterpret_cast<unsigned*>(Xs + j * RS + 2 * p) = pk;
                }
            }
        }
        lds_barrier();

        f32x16 acc[NT];
        // ---- conv1 on the x^T image
        {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            u32x4 Bf[2][NT];
            load_b(Bf[0], Xs, 0, dil);
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                if (it + 1 < NG) load_b(Bf[(it + 1) & 1], Xs, it + 1, dil);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = mma16<MODE>(A1[it], Bf[it & 1][j], acc[j]);
                if (it + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            }
            if (!RES_PREFETCH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) b1r[r] = a.b1[mt * 32 + acc_row(r, lane)];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c = col0 + j * 32 + l31;
                const int t = t0 - R2 + c;
                const bool in = t >= 0 && t < T;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[j][r] + b1r[r];
                    v = v * (v > 0.f ? 1.f : slope);
                    XTs[c * RS + mt * 32 + acc_row(r, lane)] = in ? (unsigned short)pack16<MODE>(v, 0.f) : (unsigned short)0;
                }
            }
        }
        lds_barrier();

        // ---- conv2 on the xt^T image; the residual operand is requested first and arrives under the K loop
        float rv[NT][16];
        if (RES_PREFETCH) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t_c = min(t0 + col0 + j * 32 + l31, T - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[j][r] = xb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c];
            }
        }
        {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            u32x4 Bf[2][NT];
            load_b(Bf[0], XTs, 0, 1);
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                if (it + 1 < NG) load_b(Bf[(it + 1) & 1], XTs, it + 1, 1);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = mma16<MODE>(A2[it], Bf[it & 1][j], acc[j]);
                if (it + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            }
        }
        // ---- epilogue: ((acc + b2) + x) + y_old
        if (!RES_PREFETCH) {
#pragma unroll
            for (int r = 0; r < 16; ++r) b2r[r] = a.b2[mt * 32 + acc_row(r, lane)];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int o = col0 + j * 32 + l31;
            const int t = t0 + o;
            const bool ok = o < TT && t < T;
            const int t_c = min(t, T - 1);
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long off = (long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c;
                if (!RES_PREFETCH) rv[j][r] = xb[off];
                yv[r] = a.accum ? yb[off] : 0.f;
            }
            if (ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t] = ((acc[j][r] + b2r[r]) + rv[j][r]) + yv[r];
            }
        }
        // the next tile's DMA must have landed (every wave's share) before anyone converts it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
}

template <int C, int KT, int MODE>
int launch_pair16p(const PairArgs& a, int n_cus, hipStream_t stream) {
    using P = P16<C>;
    constexpr int TT = P::N1 - (KT - 1);
    const size_t lds = (size_t)2 * C * P::XWF * 4 + (size_t)(P::XROWS + P::N1 + KT - 1) * (C + 4) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_pair16p_kernel<C, KT, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const int tiles_per_utt = (a.T + TT - 1) / TT;
    const long total = (long)tiles_per_utt * a.B;
    if (total >= (1L << 30)) return -2;
    const int grid = (int)(total < n_cus ? total : n_cus);
    hipLaunchKernelGGL((resblock_pair16p_kernel<C, KT, MODE>), dim3(grid), dim3(256), lds, stream, a, tiles_per_utt, (int)total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE>
int dispatch16p(const PairArgs& a, int n_cus, hipStream_t s) {
    if (a.C == 64) {
        if (a.k == 3) return launch_pair16p<64, 3, MODE>(a, n_cus, s);
        if (a.k == 7) return launch_pair16p<64, 7, MODE>(a, n_cus, s);
        if (a.k == 11) return launch_pair16p<64, 11, MODE>(a, n_cus, s);
    } else if (a.C == 32) {
        if (a.k == 3) return launch_pair16p<32, 3, MODE>(a, n_cus, s);
        if (a.k == 7) return launch_pair16p<32, 7, MODE>(a, n_cus, s);
        if (a.k == 11) return launch_pair16p<32, 11, MODE>(a, n_cus, s);
    }
    return -2;
}

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
};

template <int C, int KT, int MODE>
__global__ __launch_bounds__(C * 8, 2) void resblock16_kernel(const Rb16Args a) {      // C = 64: 8 waves (two per SIMD: one 118-KB workgroup per CU)
    constexpr int RS = C + 4;
    constexpr int R = (KT - 1) / 2;
    constexpr int H = 12 * R;                               // (1 + 3 + 5) R for the dilated convs + 3 R for the plain ones
    // columns every conv is evaluated on (C = 64 with 192 columns, 4 waves and two workgroups per CU — one's epilogues under the other's K loops — was
    // tried: k = 7 1365 -> 1511 us, k = 3 739 -> 704: the recomputed halo columns cost more than the overlap gives)
    constexpr int W = 384;
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
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * NOUT;
    const int tb = t0 - H;                                  // time of tile column 0
    const int T = a.T;
    const float slope = a.slope;
    const float* xb = a.x + (long)b * a.bstride;
    const int col0 = nq * (NT * 32);

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
    __syncthreads();

    f32x16 acc[NT];
    constexpr int DIL[3] = {1, 3, 5};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        // conv1 (dilation d): output column c reads x rows (MARGIN + c) + (tap - R) d
        conv_loop16<C, KT, NT, MODE>(acc, (const u32x4*)a.w1f[p], Xs + (MARGIN - R * DIL[p]) * RS, DIL[p], mt, col0, lane);
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
        // conv2 (dilation 1): output column c reads xt rows (MARGIN + c) + (tap - R)
        conv_loop16<C, KT, NT, MODE>(acc, (const u32x4*)a.w2f[p], XTs + (MARGIN - R) * RS, 1, mt, col0, lane);
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
    }
    // the block's output on the NOUT central columns: y (+)= x_3
    float* yb = a.y + (long)b * a.bstride;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int c = col0 + j * 32 + l31;
        const int t = tb + c;
        const bool ok = c >= H && c < H + NOUT && t < T;     // t >= 0 follows from c >= H
        const int t_c = min(max(t, 0), T - 1);
        float yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = a.accum ? yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c] : 0.f;
        if (ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t] = res[j][r] + yv[r];
        }
    }
}

template <int C, int KT, int MODE>
int launch_rb16(const Rb16Args& a, hipStream_t stream) {
    constexpr int R = (KT - 1) / 2;
    constexpr int NOUT = 384 - 24 * R;
    const size_t lds = (size_t)2 * (384 + 10 * R) * (C + 4) * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resblock16_kernel<C, KT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.T + NOUT - 1) / NOUT, a.B);
    hipLaunchKernelGGL((resblock16_kernel<C, KT, MODE>), grid, dim3(C * 8), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE>
int dispatch_rb16(const Rb16Args& a, int k, hipStream_t s) {
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

template <int CIN, int NT, int MODE>
__device__ __forceinline__ void conv_loopT16(f32x16 (&acc)[NT], const u32x4* __restrict__ wfrag, const unsigned short* __restrict__ src,
                                             int mt, int mtiles, int lane) {
    constexpr int RS = CIN + 4;
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
    auto load_b = [&](u32x4 (&dst)[NT], int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const unsigned short* p = bl - tap * RS + chunk * 32 + kgl * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(p + j * 32 * RS);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(p + j * 32 * RS + 4);
            dst[j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };
    u32x4 A[RING];
    auto issue_a = [&](u32x4& dst, int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const u32x4* ptr = wfrag + ((long)(tap * G + 2 * chunk + kgl) * mtiles + mt) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s)
        if (s < NG) issue_a(A[s], s);
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
        if (it + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
    };
    seg_loop<0, NG, 32>(body);
}

template <int CIN, int NW, int MODE>
__global__ __launch_bounds__(64 * NW, 2) void convT_xl16_kernel(const ConvT16Args a, int mtiles) {
    constexpr int RS = CIN + 4;
    constexpr int BN = 64, NT = 2;
    constexpr int XROWS = BN + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned short xt16[];   // [XROWS][RS], row j <-> m = t0 - 1 + j
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31;
    const int b = blockIdx.y, t0 = blockIdx.x * BN;
    const int Ti = a.Ti;
    const float* xb = a.x + (long)b * a.xbstride;
    {
        constexpr int PAIRS = CIN / 2 / NW;
        constexpr int PB = PAIRS < 16 ? PAIRS : 16;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int j = jb * 64 + lane;
            const int m = t0 - 1 + j;
            const int m_c = min(max(m, 0), Ti - 1);
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
                        *reinterpret_cast<unsigned*>(xt16 + j * RS + (w * PAIRS + p0 + p) * 2) = pack16<MODE>(u0, u1);
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
    for (int ps = 0; ps < passes; ++ps) {
        const int cb = (blockIdx.z * passes + ps) * per + cbl;
        const int mt = phase * (a.CO / 32) + cb;
        f32x16 acc[NT];
        conv_loopT16<CIN, NT, MODE>(acc, (const u32x4*)a.wf, xt16 + RS, mt, mtiles, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb * 32 + acc_row(r, lane);
            const float bi = a.bias[co];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = t0 + j * 32 + l31;
                const int t = n * S + phase - pd;
                if (n <= Ti && t >= 0 && t < a.To) yb[(long)co * a.ldy + t] = acc[j][r] + bi;
            }
        }
    }
}

template <int CIN, int NW, int MODE>
int launch_convT16(const ConvT16Args& a, hipStream_t stream) {
    const size_t lds = (size_t)65 * (CIN + 4) * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(convT_xl16_kernel<CIN, NW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const int mtiles = a.s * a.CO / 32;
    if (NW % a.s || mtiles % NW) return -2;
    const int npass = mtiles / NW;
    const long tiles = (long)((a.Ti + 1 + 63) / 64) * a.B;
    int zs = 1;
    while (tiles * zs < 4096 && zs * 2 <= npass && npass % (zs * 2) == 0) zs *= 2;
    dim3 grid((a.Ti + 1 + 63) / 64, a.B, zs);
    hipLaunchKernelGGL((convT_xl16_kernel<CIN, NW, MODE>), grid, dim3(64 * NW), lds, stream, a, mtiles);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int MODE>
int dispatch_convT16(const ConvT16Args& a, int cin, hipStream_t s) {
    const int mtiles = a.s * a.CO / 32;
    if (cin == 512 && mtiles >= 8) return launch_convT16<512, 8, MODE>(a, s);
    if (cin == 256 && mtiles >= 8) return launch_convT16<256, 8, MODE>(a, s);
    if (cin == 128 && mtiles >= 4) return launch_convT16<128, 4, MODE>(a, s);
    if (cin == 64 && mtiles >= 2) return launch_convT16<64, 2, MODE>(a, s);
    return -2;
}

template <int C, int KT, int MODE>
int launch_pair16(const PairArgs& a, hipStream_t stream) {
    constexpr int TT = N1 - (KT - 1);
    const size_t lds = (size_t)(N1 + 2 * R1MAX + N1 + KT - 1) * (C + 4) * sizeof(unsigned short);
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

// Persistent form (register-resident weights, LDS-DMA staging): one workgroup per CU walks a contiguous range of tiles.
extern "C" int cmtts_launch_resblock_pair16p(const PairArgs* ap, int mode, int n_cus, void* stream_) {
    const PairArgs& a = *ap;
    hipStream_t s = (hipStream_t)stream_;
    if (a.B <= 0 || a.T <= 0) return 0;
    if (a.dil * (a.k - 1) / 2 > R1MAX || a.x == a.y || (mode != 1 && mode != 2) || n_cus < 1) return -2;
    return mode == 1 ? dispatch16p<1>(a, n_cus, s) : dispatch16p<2>(a, n_cus, s);
}



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
    a.bstride = bstride; a.B = B; a.C = C; a.T = T; a.ld = ld; a.accum = accum; a.slope = slope;
    return mode == 1 ? dispatch_rb16<1>(a, k, (hipStream_t)stream_) : dispatch_rb16<2>(a, k, (hipStream_t)stream_);
}

// HiFi-GAN upsampler with 16-bit operands (convT_xl16_kernel): arguments as cmtts_launch_convT, wf16 = to_fragment16 of the two-tap
// stacked weights ([2][cin/16][s co / 32][64][8]), mode 1 = bf16, 2 = fp16.  0 = launched, -2 = shape not covered, -3 = HIP error.
extern "C" int cmtts_launch_convT16(const float* x, float* y, const void* wf16, const float* bias, long xbstride, long ybstride, int B,
                                    int cin, int co, int Ti, int To, int ldx, int ldy, int s, float pre_div, float slope, int mode,
                                    void* stream_) {
    if (B <= 0 || Ti <= 0) return 0;
    if (!wf16 || (s * co) % 32 || To != Ti * s || s < 2 || (s & 1) || (mode != 1 && mode != 2)) return -2;
    ConvT16Args a{x, y, wf16, bias, xbstride, ybstride, B, co, Ti, To, ldx, ldy, s, pre_div, slope};
    return mode == 1 ? dispatch_convT16<1>(a, cin, (hipStream_t)stream_) : dispatch_convT16<2>(a, cin, (hipStream_t)stream_);
}
