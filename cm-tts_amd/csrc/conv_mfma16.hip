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
//  * Own epilogue (bias, residual, accumulate): all loads of a tile before its stores.
// At 16x the fp32 MFMA rate these convs are HBM-bound (3 fp32 tensors per conv), not MFMA-bound.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "conv_args.h"
#include "conv_epilogue.h"
#include "cvt16.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Activations stream through once per launch while every workgroup re-reads the same few hundred KB of weights: the
// streams are marked non-temporal so that they do not push the weights out of the 4-MB L2 of an XCD (CONV16_NT=0: plain)
#ifndef CONV16_NT
#define CONV16_NT 0
#endif
#if CONV16_NT
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define NT_LOAD(p) (*(p))
#define NT_STORE(v, p) (*(p) = (v))
#endif

namespace {

constexpr int KC = 32;     // input channels per LDS stage (2 MFMA k-groups)
constexpr int RSX = 36;    // 16-bit elements per LDS row (72 B)

template <int MODE>
__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if (MODE == 1)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// IO: 0 = fp32 in / fp32 out, 1 = fp32 in / 16-bit activated out (conv1 of a ResBlock pair), 2 = 16-bit activated in /
// fp32 out (conv2): the intermediate xt of a pair crosses HBM in 16 bits, with exactly the value conv2's staging would
// have produced from an fp32 xt (convert(leaky_relu(xt))), so results do not change.
// MODE 3 ("fp16x3", IO 0 only): operands carried as hi = fp16(v), lo = fp16(v - hi) — two LDS images of the X tile, two
// fragment sets of the weights (lo follows hi) — and every product as three fp16 MFMAs (a_lo b_hi + a_hi b_lo + a_hi b_hi,
// fp32 accumulate): fp32-class results at 3/16 of the fp32 matrix cost.
template <int BM, int BN, int WM, int WN, int MODE, int IO>
__global__ __launch_bounds__(256, (BN > 128 || MODE == 3) ? 2 : 3) void conv1d_mfma16_kernel(const ConvArgs a, const u32x4* __restrict__ wfrag) {
    constexpr int MT = BM / (WM * 32);
    constexpr int NT = BN / (WN * 32);
    constexpr int MM = MODE == 3 ? 2 : MODE;              // MFMA element type
    constexpr int NS = MODE == 3 ? 2 : 1;                 // operand sets: (hi) [, lo]
    static_assert(MODE != 3 || IO == 0, "fp16x3 keeps fp32 activations in HBM");
    constexpr int XJ = BN / 64 + 1;       // frames per lane of a staged channel row (halo <= 64)
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];   // [2 buffers][NS images][XJ * 64][RSX]
    constexpr int IMG = (XJ * 64) * RSX;                  // elements of one image

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int tap_min = a.dil < 0 ? (a.taps - 1) * a.dil : 0;
    const int tbase = n0 - a.pad + tap_min;
    const float* Xb = a.X + z * a.x_zs0;
    const int G = (a.K + 15) / 16;                 // k-groups
    const int MTn = (a.M + 31) / 32;               // m-tiles in the packed weights
    const int nchunks = (a.K + KC - 1) / KC;

    // ---- staging: wave `wid` converts channel pairs 4*wid .. 4*wid+3 of the chunk, lanes run over frames.
    // K is a multiple of 32 (checked by the launcher) and the LDS tile has XJ*64 rows, so nothing here is
    // predicated: out-of-range frames load a clamped address and are multiplied by 0.  Leaky ReLU and the
    // zero padding are one select + one multiply: v * (v > 0 ? fpos : fneg).
    float fpos[XJ], fneg[XJ];
    int tcl[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int t = tbase + lane + 64 * j;
        const bool okt = t >= 0 && t < a.Tin;
        fpos[j] = okt ? 1.f / a.pre_div : 0.f;
        fneg[j] = okt ? a.pre_slope / a.pre_div : 0.f;
        // columns beyond the tile's halo are never read: point them at one line instead of fetching them
        tcl[j] = min(max(min(t, tbase + BN + (a.taps - 1) * a.dil - 1), 0), a.Tin - 1);
    }
    float xr[4][2][XJ];                 // IO == 2: the 16-bit values, zero-extended, in the same registers
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long row = (long)(chunk * KC + (wid * 4 + p) * 2 + h) * a.ldx;
                if (IO == 2) {
                    const unsigned short* xrow = reinterpret_cast<const unsigned short*>(a.X) + z * a.x_zs0 + row;
#pragma unroll
                    for (int j = 0; j < XJ; ++j) xr[p][h][j] = __uint_as_float((unsigned)NT_LOAD(xrow + tcl[j]));
                } else {
                    const float* xrow = Xb + row;
#pragma unroll
                    for (int j = 0; j < XJ; ++j) xr[p][h][j] = NT_LOAD(xrow + tcl[j]);
                }
            }
    };
    auto store_x = [&](int buf) {
        unsigned short* xb = xs + buf * NS * IMG;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                unsigned pk;
                if (IO == 2) {
                    pk = fpos[j] != 0.f ? (__float_as_uint(xr[p][0][j]) | (__float_as_uint(xr[p][1][j]) << 16)) : 0u;
                } else {
                    const float v0 = xr[p][0][j], v1 = xr[p][1][j];
                    const float a0 = v0 * (v0 > 0.f ? fpos[j] : fneg[j]), a1 = v1 * (v1 > 0.f ? fpos[j] : fneg[j]);
                    pk = pack16<MM>(a0, a1);
                    if (MODE == 3) {
                        const cvt_f16x2 h = __builtin_bit_cast(cvt_f16x2, pk);
                        *reinterpret_cast<unsigned*>(xb + IMG + (lane + 64 * j) * RSX + (wid * 4 + p) * 2) =
                            pack16<2>(a0 - (float)h[0], a1 - (float)h[1]);
                    }
                }
                *reinterpret_cast<unsigned*>(xb + (lane + 64 * j) * RSX + (wid * 4 + p) * 2) = pk;
            }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const long wset = (long)a.taps * G * MTn * 64;        // u32x4 per fragment set (MODE 3: the lo set follows the hi set)
    auto load_a = [&](u32x4 (&dst)[NS][MT], int tap, int kg) {
        const int kgc = min(kg, G - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int mt = min(m0 / 32 + wm * MT + i, MTn - 1);
                dst[s][i] = wfrag[s * wset + ((long)(tap * G + kgc) * MTn + mt) * 64 + lane];
            }
    };
    auto load_b = [&](u32x4 (&dst)[NS], const unsigned short* xb, int q, int j) {    // q = tap * 2 + k-group-in-chunk
        const int tap = q >> 1, kgl = q & 1;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned short* p = xb + s * IMG + ((wn * NT + j) * 32 + l31 + tap * a.dil - tap_min) * RSX + kgl * 16 + khalf * 8;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(p);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(p + 4);
            dst[s] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };

    load_x(0);
    store_x(0);
    __syncthreads();
    const int nq = a.taps * 2;                     // (tap, k-group-in-chunk) pairs per chunk
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool has_next = chunk + 1 < nchunks;
        if (has_next) load_x(chunk + 1);
        const unsigned short* xb = xs + (chunk & 1) * NS * IMG;
        // A fragments one (tap, k-group) ahead; B fragments one MFMA column ahead (rolling pair), so a wave
        // never holds more than two B fragments.
        u32x4 A[2][NS][MT], Bf[2][NS];
        load_a(A[0], 0, chunk * 2);
        load_b(Bf[0], xb, 0, 0);
        for (int q = 0; q < nq; q += 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int qn = min(q + s + 1, nq - 1);
                load_a(A[(s + 1) & 1], qn >> 1, chunk * 2 + (qn & 1));
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                                    const int cur = (s * NT + j) & 1;
                    if (j + 1 < NT) load_b(Bf[cur ^ 1], xb, q + s, j + 1);
                    else load_b(Bf[cur ^ 1], xb, qn, 0);
                    // a chunk whose second k-group lies beyond K contributes zeros (X rows are zero-filled)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        if (MODE == 3) {      // small terms first
                            acc[i][j] = mma16<MM>(A[s][1][i], Bf[cur][0], acc[i][j]);
                            acc[i][j] = mma16<MM>(A[s][0][i], Bf[cur][1], acc[i][j]);
                        }
                        acc[i][j] = mma16<MM>(A[s][0][i], Bf[cur][0], acc[i][j]);
                    }
                }
            }
        }
        if (has_next) store_x((chunk + 1) & 1);
        __syncthreads();
    }

    // ---- the text-side FFN convs (round 3, opt-in "text16"): bias, alpha, GELU, residual, length mask — conv_epilogue.h's compile-time form
    if (IO == 0 && MODE != 3 && a.text_epi) {      // wave-uniform
        const ConvOut& o = a.out[0];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int mb = m0 + (wm * MT + i) * 32, n = n0 + (wn * NT + j) * 32 + l31;
                if (o.act == ACT_GELU_ERF) epi_tile_simple<ACT_GELU_ERF>(o, acc[i][j], mb, 4 * khalf, n, a.M, a.N, z);
                else if (o.act == ACT_RELU) epi_tile_simple<ACT_RELU>(o, acc[i][j], mb, 4 * khalf, n, a.M, a.N, z);
                else epi_tile_simple<ACT_NONE>(o, acc[i][j], mb, 4 * khalf, n, a.M, a.N, z);
            }
        return;
    }
    // ---- epilogue, specialised for the ResBlock convs (bias [+ residual] [+ accumulate], unit stride):
    // every load of a 32x32 tile is issued before its first store, and nothing fences one tile from the
    // next, so the residual / old-Y loads of tile k+1 fly while tile k is being written.
    const ConvOut& o = a.out[0];
    float* __restrict__ yb = o.Y + z * o.y_zs0;
    const float* __restrict__ rb = o.res ? o.res + z * o.r_zs0 : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float bi[16];
        unsigned mrow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            mrow[r] = (unsigned)min(m, a.M - 1);
            bi[r] = o.bias[mrow[r]];
        }
        const bool full_m = m0 + (wm * MT + i) * 32 + 32 <= a.M;      // wave-uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + (wn * NT + j) * 32 + l31;
            const unsigned nc = (unsigned)min(n, a.N - 1);
            float rv[16], yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                rv[r] = rb ? NT_LOAD(rb + mrow[r] * (unsigned)o.ldr + nc) : 0.f;
                yv[r] = o.accum ? NT_LOAD(yb + mrow[r] * (unsigned)o.ldy + nc) : 0.f;
            }
            if (IO == 1) {      // conv1 of a pair: xt leaves as convert(leaky_relu(xt)), what conv2's staging would compute
                unsigned short* y16 = reinterpret_cast<unsigned short*>(o.Y) + z * o.y_zs0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    float v = acc[i][j][r] + bi[r];
                    v = v * (v > 0.f ? 1.f : a.y16_slope);
                    if (n < a.N && m < a.M) NT_STORE((unsigned short)pack16<MM>(v, 0.f), y16 + (unsigned)m * (unsigned)o.ldy + (unsigned)n);
                }
            } else if (full_m) {
                if (n < a.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) NT_STORE(((acc[i][j][r] + bi[r]) + rv[r]) + yv[r], yb + mrow[r] * (unsigned)o.ldy + (unsigned)n);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (n < a.N && m < a.M) NT_STORE(((acc[i][j][r] + bi[r]) + rv[r]) + yv[r], yb + (unsigned)m * (unsigned)o.ldy + (unsigned)n);
                }
            }
        }
    }
}


template <int BM, int BN, int WM, int WN>
int launch16(const ConvArgs& a, const void* wfrag, int mode, int nbatch, hipStream_t stream) {
    const int adil = a.dil < 0 ? -a.dil : a.dil;
    const int halo = (a.taps - 1) * adil;
    if (halo > 64) return -2;
    const size_t lds = (size_t)(mode == 3 ? 2 : 1) * 2 * (BN + 64) * RSX * sizeof(unsigned short);
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nbatch);
    const int io = a.y16 ? 1 : (a.x16 ? 2 : 0);
    if (mode == 3) {
        if (io != 0) return -2;
        hipLaunchKernelGGL((conv1d_mfma16_kernel<BM, BN, WM, WN, 3, 0>), grid, dim3(256), lds, stream, a, (const u32x4*)wfrag);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
#define CMTTS_L16(M_, IO_) hipLaunchKernelGGL((conv1d_mfma16_kernel<BM, BN, WM, WN, M_, IO_>), grid, dim3(256), lds, stream, a, (const u32x4*)wfrag)
    if (mode == 1) { if (io == 1) CMTTS_L16(1, 1); else if (io == 2) CMTTS_L16(1, 2); else CMTTS_L16(1, 0); }
    else { if (io == 1) CMTTS_L16(2, 1); else if (io == 2) CMTTS_L16(2, 2); else CMTTS_L16(2, 0); }
#undef CMTTS_L16
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// Plain Conv1d with 16-bit operands: a->A is ignored, `wfrag` = fragment-order weights
// [taps][ceil(K/16)][ceil(M/32)][64][8] (zero padded), mode 1 = bf16, 2 = fp16.  Supports zdiv == 1,
// split == INT_MAX, dil > 0, K % 32 == 0, 0 <= pre_slope <= 1 and an epilogue of bias [+ residual]
// [+ accumulate] only (the HiFi-GAN ResBlock convs).
extern "C" int cmtts_launch_conv16(const ConvArgs* ap, const void* wfrag, int mode, int nbatch, void* stream_) {
    const ConvArgs& a = *ap;
    hipStream_t stream = (hipStream_t)stream_;
    if (a.M <= 0 || a.N <= 0 || nbatch <= 0) return 0;
    const ConvOut& o = a.out[0];
    if (a.zdiv != 1 || a.split != INT_MAX || a.dil <= 0 || mode < 1 || mode > 3 || a.K % KC != 0) return -2;
    if ((a.x16 && a.y16) || (a.y16 && (o.res || o.accum)) || (a.x16 && (a.pre_div != 1.f))) return -2;
    if (a.text_epi) {     // bias, alpha, none / GELU, residual, length mask; fp32 in and out, bf16 / fp16 operands
        if (mode == 3 || a.x16 || a.y16 || o.bvec || o.accum || (o.act != ACT_NONE && o.act != ACT_GELU_ERF && o.act != ACT_RELU) || o.div != 1.f || o.rmul != 0.f || o.ostride != 1 ||
            o.ooff_base != 0 || o.ooff_mul != 0 || o.row_off != 0 || o.Tout != a.N)
            return -2;
    } else if (!o.bias || o.bvec || o.lens || o.alpha != 1.f || o.act != ACT_NONE || o.div != 1.f || o.rmul != 0.f || o.ostride != 1 || o.ooff_base != 0 ||
        o.row_off != 0 || o.Tout != a.N)
        return -2;
    // 128-frame tiles everywhere: these convs are HBM-bound, what matters is loads in flight (3 workgroups/CU)
    if (a.text_epi) {     // K = 256: the X-resident kernel (conv_xt16.hip: one staging round trip per tile, no barrier in the K loop; same bits)
        const int rx = cmtts_launch_conv_xt16(ap, wfrag, mode, nbatch, stream_);
        if (rx != -2) return rx;
    }
    // the text side's short sequences (85 phonemes, 170, ...): 96-column tiles where they pad less than 128-column ones
    if (a.text_epi && a.M > 64 && (a.N + 95) / 96 * 96 < (a.N + 127) / 128 * 128) return launch16<128, 96, 4, 1>(a, wfrag, mode, nbatch, stream);
    if (a.M > 64) {
        static const char* tile = getenv("CMTTS_C16_TILE");     // experiment switch: "22" = round-1 tiling (2 x 2 waves)
        if (tile && !strcmp(tile, "22")) return launch16<128, 128, 2, 2>(a, wfrag, mode, nbatch, stream);
        return launch16<128, 128, 4, 1>(a, wfrag, mode, nbatch, stream);
    }
    if (a.M > 32) return launch16<64, 128, 2, 2>(a, wfrag, mode, nbatch, stream);
    return launch16<32, 128, 1, 4>(a, wfrag, mode, nbatch, stream);
}
