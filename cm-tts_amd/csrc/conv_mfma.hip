// fp32 MFMA implicit-GEMM Conv1D for gfx950 (CDNA4) — the one dense-contraction kernel of the
// CM-TTS inference path (denoiser residual blocks, FFT-block QKV/FFN, attention QK^T / PV,
// variance predictors, HiFi-GAN convs and polyphase transposed convs).
//
// Design:
//  * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain) at the 157 TFLOP/s matrix rate.
//    A operand lane l = A[m = l&31][k = l>>5], B operand lane l = X[k = l>>5][n = l&31],
//    C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
//  * 256-thread workgroups = 4 wave64; each wave owns an (MT*32) x (NT*32) output tile held in
//    MT*NT f32x16 accumulators; tiles 128x128 / 64x256 / 32x256 (M x N).
//  * K loop over (16-channel chunk, tap).  The X tile [16][BN + halo] is staged ONCE per chunk in
//    LDS (zero padding and the fused pre-activation applied while staging) and re-read at a
//    per-tap column offset; the k-major weight tile [16][BM] is staged per (chunk, tap).
//    Both operands are read from LDS with conflict-free ds_read_b32 (32 consecutive floats per
//    half-wave).  Global loads for iteration i+1 are issued before the MFMAs of iteration i and
//    written to the other LDS buffer after them: one barrier per iteration.
//  * Epilogue fused: bias, alpha, ReLU/GELU/tanh, residual (+ per-batch channel vector), division,
//    accumulate, length mask, strided output (transposed conv phases), split outputs
//    (residual/skip halves of the denoiser block) and the sigmoid*tanh gate.
#include <hip/hip_runtime.h>
#include <limits.h>
#include "conv_args.h"
#include "conv_epilogue.h"
#include "gate.h"

namespace {

// KC = input channels per K chunk (template parameter): 16 for the large launches, 64 for the small
// latency-bound ones (4x fewer barrier-separated iterations, 4x the bytes in flight per iteration)

// TB = taps staged (and multiplied) per barrier-separated iteration: 1 for the large launches; 5 for the small k>1
// launches (text-side FFN / predictors), whose 8 MFMAs per wave per tap cannot hide a barrier.  The (chunk, tap, k)
// accumulation order is the same for every TB.
// Debugging aid (tools/conv_phases.py): cycle counts per wave of the launches whose (M, K) match — [workgroup][wave][8]:
// start, after the prologue, sum of the MFMA blocks, sum of (LDS stores + barrier), before the epilogue, end.
__device__ long long* d_conv_dbg = nullptr;
__device__ int d_conv_dbg_m = 0, d_conv_dbg_k = 0;

template <int BM, int BN, int WM, int WN, int EPI, int KC, int TB>
__global__ __launch_bounds__(256, (KC > 16 || TB > 1) ? 2 : 3) void conv1d_mfma_kernel(const ConvArgs a) {
    constexpr int MT = BM / (WM * 32);
    constexpr int NT = BN / (WN * 32);
    constexpr int XJ = (BN + 64 + 63) / 64;         // columns per lane of an X row (halo <= 64)
    constexpr int WQ = BM / 4;                      // float4 per weight-tile row
    constexpr int WV = (KC * WQ + 255) / 256;       // float4 per thread per weight tile
    static_assert(WM * WN == 4, "4 waves");
    static_assert(EPI != EPI_GATED || MT == 2, "gated epilogue pairs the two m-tiles of a wave");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int zq = z / a.zdiv, zr = z - zq * a.zdiv;

    const int adil = a.dil < 0 ? -a.dil : a.dil;
    const int XW = BN + (a.taps - 1) * adil;
    const int tap_min = a.dil < 0 ? (a.taps - 1) * a.dil : 0;
    const int tbase = n0 - a.pad + tap_min;

    float* Ws = smem;                         // [2][TB][KC][BM]
    float* Xs = smem + 2 * TB * KC * BM;      // [2][KC][XW]

    const float* Ab = a.A + zq * a.a_zs0 + zr * a.a_zs1;
    const float* Xb = a.X + zq * a.x_zs0 + zr * a.x_zs1;

    const int nchunks = (a.K + KC - 1) / KC;
    const int tgroups = (a.taps + TB - 1) / TB;      // iterations per chunk
    const int niter = nchunks * tgroups;

    float4 wreg[TB][WV];
    constexpr int XR = KC / 4;                      // X rows staged per wave
    float xreg[XR][XJ];

    // Guarded loads are written as UNCONDITIONAL loads from a clamped (always in-bounds) address
    // followed by a select: a load under a per-lane branch makes hipcc emit an exec-masked branch and
    // a vmcnt(0) per element, which serialises the whole prefetch (cdna_hip_programming.md §5 trap c).
    // The raw loaded values stay untouched in registers until store time (after the MFMAs): the
    // zero-fill selects and the pre-activation are applied in store_*, so nothing waits on vmcnt
    // between issuing the prefetch and the MFMA block.
    // Address arithmetic of the prefetch in 32 bits on precomputed per-thread terms (round 2: issuing the 36 loads of a
    // 64-channel iteration cost 1.9 k cycles against 2 k of MFMAs — 64-bit multiplies and clamps per element; every operand
    // slice of this path is far below 2^32 elements): wave-uniform base pointer + unsigned per-lane offset
    int wrow[WV];
    unsigned wcol[WV];
#pragma unroll
    for (int v = 0; v < WV; ++v) {
        const int i = min(tid + v * 256, KC * WQ - 1);
        wrow[v] = i / WQ;
        wcol[v] = (unsigned)min(m0 + (i - wrow[v] * WQ) * 4, a.a_cols - 4);
    }
    const unsigned a_ldu = (unsigned)a.a_ld, a_tsu = (unsigned)a.a_tap_stride;
    auto load_w = [&](int chunk, int tap0) {
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int tap = min(tap0 + tb, a.taps - 1);          // taps beyond the kernel are loaded again and never used
            const unsigned tbase_w = (unsigned)tap * a_tsu;
#pragma unroll
            for (int v = 0; v < WV; ++v) {
                const unsigned krow_c = (unsigned)min(chunk * KC + wrow[v], a.K - 1);
                wreg[tb][v] = *reinterpret_cast<const float4*>(Ab + (tbase_w + krow_c * a_ldu + wcol[v]));
            }
        }
    };
    auto store_w = [&](int buf, int chunk) {
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int v = 0; v < WV; ++v) {
                const int i = tid + v * 256;
                const int row = i / WQ, c4 = i - row * WQ;
                const bool ok = chunk * KC + row < a.K && m0 + c4 * 4 < a.a_cols;
                float4 w = wreg[tb][v];
                if (!ok) w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < KC * WQ) *reinterpret_cast<float4*>(Ws + (buf * TB + tb) * KC * BM + i * 4) = w;
            }
    };
    unsigned xcol[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) xcol[j] = (unsigned)min(max(tbase + lane + 64 * j, 0), a.Tin - 1);
    const unsigned ldxu = (unsigned)a.ldx;
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const unsigned rowoff = (unsigned)min(chunk * KC + wid + 4 * r, a.K - 1) * ldxu;      // wave-uniform
#pragma unroll
            for (int j = 0; j < XJ; ++j) xreg[r][j] = Xb[rowoff + xcol[j]];
        }
    };
    // The pre-activation is chosen ONCE per call (wave-uniform branches around three copies of the element loop), and the
    // column validity is a per-lane mask computed before the loop: written inside the loop, "if (pre_div != 1) v /= pre_div"
    // became an unconditional IEEE division + select, the slope a scalar load + lgkmcnt(0), and "col < XW" an exec-mask
    // branch — ~28 instructions per staged element, 3.6 k cycles per 64-channel chunk against its 2 k cycles of MFMAs
    // (round 2, found in the ISA of the 64 x 64 configuration: the FFT blocks' out-projection / FFN linear, the predictors)
    const float pre_div = a.pre_div, pre_slope = a.pre_slope;
    bool colok[XJ], tok[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int col = lane + 64 * j, t = tbase + col;
        colok[j] = col < XW;
        tok[j] = t >= 0 && t < a.Tin;
    }
    auto store_x = [&](int buf, int chunk) {
        float* xs = Xs + buf * KC * XW + wid * XW + lane;
        if (pre_div == 1.0f && pre_slope == 1.0f) {
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                const bool kok = chunk * KC + wid + 4 * r < a.K;
#pragma unroll
                for (int j = 0; j < XJ; ++j) {
                    const float v = (kok && tok[j]) ? xreg[r][j] : 0.f;
                    if (j == 0 || colok[j]) xs[4 * r * XW + 64 * j] = v;      // XW >= BN >= 64: the first 64 columns always exist
                }
            }
        } else if (pre_div == 1.0f) {
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                const bool kok = chunk * KC + wid + 4 * r < a.K;
#pragma unroll
                for (int j = 0; j < XJ; ++j) {
                    float v = (kok && tok[j]) ? xreg[r][j] : 0.f;
                    v = v > 0.f ? v : v * pre_slope;
                    if (j == 0 || colok[j]) xs[4 * r * XW + 64 * j] = v;      // XW >= BN >= 64: the first 64 columns always exist
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                const bool kok = chunk * KC + wid + 4 * r < a.K;
#pragma unroll
                for (int j = 0; j < XJ; ++j) {
                    float v = (kok && tok[j]) ? xreg[r][j] : 0.f;
                    v = v / pre_div;
                    v = v > 0.f ? v : v * pre_slope;
                    if (j == 0 || colok[j]) xs[4 * r * XW + 64 * j] = v;      // XW >= BN >= 64: the first 64 columns always exist
                }
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

    long long* dbg = d_conv_dbg;
    if (dbg && (a.M != d_conv_dbg_m || a.K != d_conv_dbg_k)) dbg = nullptr;
    long long t_start = 0, t_pro = 0, t_mma = 0, t_sync = 0, tq = 0;
    if (dbg) t_start = (long long)__builtin_readcyclecounter();
    // prologue: stage iteration 0
    load_w(0, 0);
    load_x(0);
    store_w(0, 0);
    store_x(0, 0);
    __syncthreads();

    if (dbg) t_pro = (long long)__builtin_readcyclecounter();
    int chunk = 0, tap = 0;
    const int a_off = wm * MT * 32 + (lane & 31);
    const int b_off = wn * NT * 32 + (lane & 31);
    const int khalf = lane >> 5;

    for (int it = 0; it < niter; ++it) {
        int ntap = tap + TB, nchunk = chunk;                  // tap = first tap of this iteration's group
        if (ntap >= a.taps) { ntap = 0; nchunk = chunk + 1; }
        const bool has_next = it + 1 < niter;
        const bool next_x = has_next && ntap == 0;
        if (has_next) load_w(nchunk, ntap);
        if (next_x) load_x(nchunk);
        __builtin_amdgcn_sched_barrier(0);           // prefetch loads stay above the MFMA block
        if (dbg) tq = (long long)__builtin_readcyclecounter();
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            if (tap + tb < a.taps) {                 // wave-uniform
                const float* wsc = Ws + ((it & 1) * TB + tb) * KC * BM + a_off;
                const float* xsc = Xs + (chunk & 1) * KC * XW + b_off + ((tap + tb) * a.dil - tap_min);
                // operands of k-step kk+1 are read from LDS before the MFMAs of k-step kk are issued
                const float* wk = wsc + khalf * BM;
                const float* xk = xsc + khalf * XW;
                float av[MT], bv[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = wk[i * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = xk[j * 32];
#pragma unroll
                for (int kk = 0; kk < KC / 2; ++kk) {
                    float nav[MT], nbv[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) nav[i] = kk + 1 < KC / 2 ? wk[(kk + 1) * 2 * BM + i * 32] : 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) nbv[j] = kk + 1 < KC / 2 ? xk[(kk + 1) * 2 * XW + j * 32] : 0.f;
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i) av[i] = nav[i];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[j] = nbv[j];
                }
            }
        }

        if (dbg) { const long long t = (long long)__builtin_readcyclecounter(); t_mma += t - tq; tq = t; }
        if (has_next) store_w((it + 1) & 1, nchunk);
        if (next_x) store_x(nchunk & 1, nchunk);
        __syncthreads();
        if (dbg) t_sync += (long long)__builtin_readcyclecounter() - tq;
        tap = ntap;
        chunk = nchunk;
    }
    const long long t_epi = dbg ? (long long)__builtin_readcyclecounter() : 0;
    auto dbg_out = [&]() {
        if (dbg && lane == 0) {
            long long* p = dbg + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wid) * 8;
            p[0] = t_start; p[1] = t_pro; p[2] = t_mma; p[3] = t_sync; p[4] = t_epi; p[5] = (long long)__builtin_readcyclecounter();
        }
    };

    // ------------------------------------------------------------------ epilogue
    const int rbase = 4 * khalf;
    const int col = lane & 31;
    if constexpr (EPI == EPI_GATED) {
        // packed rows: each 64-row group = [32 sigmoid-gate rows | 32 tanh-filter rows] of the same
        // 32 output channels (weights are permuted at import time).  out rows = M/2.
        const ConvOut& o = a.out[0];
        const int zrow0 = (m0 + wm * 64) / 2;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + (wn * NT + j) * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + rbase;
                const int mg = m0 + wm * 64 + row;          // packed gate row
                const int mf = mg + 32;                     // packed filter row
                if (mf < a.M && n < a.N) {
                    const float g = acc[0][j][r] + o.bias[mg];
                    const float f = acc[1][j][r] + o.bias[mf];
                    const float zv = cmtts_gate(g, f);
                    const int t = n;
                    if (t < o.Tout)
                        o.Y[zq * o.y_zs0 + zr * o.y_zs1 + (long)(zrow0 + row) * o.ldy + t] = zv;
                }
            }
        }
    } else {
        const ConvOut& o = (m0 >= a.split) ? a.out[1] : a.out[0];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                epi_tile(o, acc[i][j], m0 + (wm * MT + i) * 32, rbase, n0 + (wn * NT + j) * 32 + col, a.M, a.N, zq, zr);
    }
    dbg_out();
}

template <int BM, int BN, int WM, int WN, int EPI, int KC = 16, int TB = 1>
int launch_cfg(const ConvArgs& a, int nbatch, hipStream_t stream) {
    const int adil = a.dil < 0 ? -a.dil : a.dil;
    const int halo = (a.taps - 1) * adil;
    if (halo > 64) return -2;
    const int XW = BN + halo;
    const size_t lds = (size_t)(2 * TB * KC * BM + 2 * KC * XW) * sizeof(float);
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nbatch);
    hipLaunchKernelGGL((conv1d_mfma_kernel<BM, BN, WM, WN, EPI, KC, TB>), grid, dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

extern "C" int cmtts_launch_conv(const ConvArgs* ap, int epi, int nbatch, void* stream_) {
    const ConvArgs& a = *ap;
    hipStream_t stream = (hipStream_t)stream_;
    if (a.M <= 0 || a.N <= 0 || nbatch <= 0) return 0;
    if ((a.a_ld & 3) || (a.a_cols & 3)) return -2;
    if (epi == EPI_GATED) {
        if (a.M % 64) return -2;
        return launch_cfg<128, 128, 2, 2, EPI_GATED>(a, nbatch, stream);
    }
    if (a.M > 64) {
        if (a.split != INT_MAX && (a.split % 128)) return -2;
        // small launches (text-side convs: N = phonemes; per-step projections) cannot fill 256 CUs with
        // 128x128 tiles: use 64x64
        const long big = (long)((a.N + 127) / 128) * ((a.M + 127) / 128) * nbatch;
        // (measured on cfg2: below two 128x128 workgroups per CU the 64x64 tiling wins, 18.2 -> 17.9 ms/step)
        if ((big < 512 || a.small_tiles) && a.split == INT_MAX) {
            // k=1 contractions only: there a 64-channel chunk accumulates in the same order as four 16-channel
            // chunks, so an utterance's result does not depend on which configuration its batch size selects
            // (tests: every utterance bit-identical to synthesising it alone)
            if (a.taps == 1 && a.K >= 128) return launch_cfg<64, 64, 2, 2, EPI_PLAIN, 64>(a, nbatch, stream);
            // k > 1: up to five taps per barrier (same accumulation order)
            if (a.taps > 1) return launch_cfg<64, 64, 2, 2, EPI_PLAIN, 16, 5>(a, nbatch, stream);
            return launch_cfg<64, 64, 2, 2, EPI_PLAIN>(a, nbatch, stream);
        }
        return launch_cfg<128, 128, 2, 2, EPI_PLAIN>(a, nbatch, stream);
    }
    if (a.split != INT_MAX) return -2;
    // few rows AND few columns (attention products of short utterances: V^T = h^T Wv^T and K^T Q with M = phonemes <= 64):
    // the 256-column tiles would stage mostly padding; 64x64 tiles with 64-channel chunks, same accumulation order
    if (a.taps == 1 && a.K >= 128 && (long)((a.N + 255) / 256) * nbatch < 256)
        return launch_cfg<64, 64, 2, 2, EPI_PLAIN, 64>(a, nbatch, stream);
    if (a.M > 32) return launch_cfg<64, 256, 1, 4, EPI_PLAIN>(a, nbatch, stream);
    return launch_cfg<32, 256, 1, 4, EPI_PLAIN>(a, nbatch, stream);
}

// stamps buffer (device) + the (M, K) of the launches to stamp; nullptr = off
extern "C" void cmtts_conv_set_debug(long long* dbg, int M, int K) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d_conv_dbg), &dbg, sizeof(dbg));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d_conv_dbg_m), &M, sizeof(M));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d_conv_dbg_k), &K, sizeof(K));
}
