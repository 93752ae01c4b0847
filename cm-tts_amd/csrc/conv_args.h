// Argument block of the fp32 MFMA implicit-GEMM Conv1D kernel (conv_mfma.hip).
//
// One kernel covers every dense contraction on the CM-TTS inference path:
//   Y[z][m][t_out] = epilogue( sum_{tap, k} A[z][tap][k][m] * pre(X[z][k][n + tap*dil - pad]) )
// with m = output channel (GEMM M), n = output position (GEMM N), k = input channel (GEMM K).
// All activations are channel-major ([.., C, T], T contiguous) so the frame axis is the coalesced
// axis for loads, LDS staging (with the Conv1D halo) and stores.
#pragma once
#include <stdint.h>

enum ConvAct { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_ERF = 2, ACT_TANH = 3 };
enum ConvEpi { EPI_PLAIN = 0, EPI_GATED = 1 };

struct ConvOut {
    float* Y;             // output base
    long y_zs0, y_zs1;    // batch strides (z / zdiv, z % zdiv)
    int ldy;              // row stride
    int row_off;          // output row = m - row_off
    int Tout;             // valid output width (store guard: 0 <= t_out < Tout)
    int ostride;          // t_out = n * ostride + ooff_base + (z % zdiv) * ooff_mul
    int ooff_base, ooff_mul;
    const float* bias;    // [M] indexed by m (may be null)
    const float* bvec;    // per-(z/zdiv) row vector: bvec[(z/zdiv)*bvec_zs + m] (may be null)
    long bvec_zs;
    const float* res;     // residual, indexed like Y with its own strides (may be null)
    long r_zs0, r_zs1;
    int ldr;
    const int64_t* lens;  // columns t_out >= lens[z / zdiv] are written as 0 (may be null)
    float alpha;          // v = (acc + bias) * alpha
    int act;              // ConvAct
    float div;            // v = v / div   (1 = skip)
    float rmul;           // v = v * rmul  (0 = skip): the residual blocks' 1 / sqrt(2) as a multiplication (round 5: the IEEE division is a
                          // ten-instruction sequence per element, and next to fp32 MFMAs every VALU instruction costs its own issue time)
    int accum;            // Y += v instead of Y = v
};

struct ConvArgs {
    const float* A;       // k-major operand: A[tap][k][m], m contiguous (packed weights, or an activation)
    const float* X;       // X[k][t], t contiguous
    int M, N, K;          // valid rows of Y / GEMM columns to compute / valid rows of X and A
    int taps, dil, pad;   // t_in = n + tap*dil - pad   (dil may be negative: transposed conv phases)
    int Tin;              // valid input width (zero padding outside [0, Tin))
    int a_ld;             // A row stride in floats (multiple of 4, 16-B aligned rows)
    int a_cols;           // valid columns per A row (multiple of 4): float4 guard
    long a_tap_stride;
    int ldx;
    int zdiv;             // z -> (z / zdiv, z % zdiv)
    long a_zs0, a_zs1, x_zs0, x_zs1;
    float pre_div;        // X is divided by this on load (1 = skip)
    float pre_slope;      // then leaky_relu with this slope (1 = identity)
    int split;            // rows m >= split use out[1]
    ConvOut out[2];
    // 16-bit kernel only (conv_mfma16.hip): activations exchanged between two convs of a ResBlock in 16 bits
    int x16;              // X holds 16-bit values that are already activated (no pre_div / pre_slope applied)
    int y16;              // Y is written as 16-bit leaky_relu(v, y16_slope) instead of fp32 v
    float y16_slope;
    int small_tiles;      // fp32 kernel: prefer 64x64 tiles even where the launch would fill the chip with 128x128 ones (N far from a multiple of 128)
    // 16-bit kernel only: the weights once more in ITERATION order [K/32][taps][2][M/32][64][8] for the deep-ring variant of
    // the wide convs (null = none; the launcher then runs the one-step-ahead form on `wfrag`)
    const void* wfrag_iter;
    // X-resident kernel only (conv_xres.hip): LayerNorm over the K rows of X as a prologue on the staged tile (null = none)
    const float* ln_g;
    const float* ln_b;
    float ln_eps;
    int xres_nt;              // 0 = the launcher chooses 96- or 32-column tiles by how full the chip gets; 1 / 3 = forced (tests, tools)
    const int64_t* ln_lens;   // LayerNorm prologue: columns t >= ln_lens[z] become 0 (layernorm_ct_kernel's optional mask); nullptr = none
    int ln_skip_tiles;        // conv_xres only, with ln_lens: a column tile WHOLLY beyond ln_lens[z] is not computed and its outputs are NOT WRITTEN
                              // (stale workspace contents stay).  Set only by callers whose every consumer of those columns masks them by select with
                              // the same lengths: the ragged FFT blocks (pad_lens) and the predictor convs (the next LayerNorm's mask)
    // conv_xres.hip, FFN fusion: the k = 1 linear that follows (W2: [M2][M], M2 = 256) applied to this workgroup's 128 activated output rows while
    // they are on chip — its K-segment partial sum [M2][N] goes to part + z * part_zs0 + (m-block) * part_zs1 (row stride part_ld) instead of
    // the activated rows going to out[0].Y; w2frag = W2 as A fragments in iteration order [M/16][2][M2/32][64][4] (to_fragment_iter_order)
    const float* w2frag;
    float* part;
    long part_zs0, part_zs1;
    int part_ld, M2;
    // 16-bit kernel only (round 3, the opt-in 16-bit text side): out[0] through conv_epilogue.h's epi_tile_simple (bias, alpha, none / GELU,
    // residual, length mask) instead of the ResBlock epilogue
    int text_epi;
};

extern int g_conv_xt16;   // conv_xt16.hip: X-resident kernel for the text16 convs (internal switch "text_xt16")
#ifdef __cplusplus
extern "C" {
#endif
// Chooses the tile configuration from (M, N, taps*|dil|) and launches on `stream`.  epi = ConvEpi.
// Returns 0 or a negative cmtts status.
int cmtts_launch_conv(const ConvArgs* a, int epi, int nbatch, void* stream);
// 16-bit-operand variant (conv_mfma16.hip): wfrag = fragment-order weights, mode 1 = bf16, 2 = fp16.
int cmtts_launch_conv16(const ConvArgs* a, const void* wfrag, int mode, int nbatch, void* stream);
// X-resident form for the text side's K = 256 convs (conv_xt16.hip; tried first by cmtts_launch_conv16 when a->text_epi; same bits); -2 = not covered
int cmtts_launch_conv_xt16(const ConvArgs* a, const void* wfrag, int mode, int nbatch, void* stream);
// X-resident variant for short sequences (conv_xres.hip): wfrag = fp32 fragment-order weights; -2 = unsupported.
int cmtts_launch_conv_xres(const ConvArgs* a, const float* wfrag, int nbatch, void* stream);
void cmtts_conv_set_debug(long long* dbg, int M, int K);   // generic kernel: cycle counters of the launches with this (M, K)
int cmtts_xres_set_nt(int nt);               // internal switch "xres_nt" (0 = launcher's rule, 1 / 3 = tile width of launches that do not choose); returns the previous value
void cmtts_xres_set_debug(long long* dbg);   // cycle stamps [workgroup][wave][8] (tools/xres_phases.py); nullptr = off
#ifdef __cplusplus
}
#endif
