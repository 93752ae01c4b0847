// Arguments of the fused HiFi-GAN ResBlock pair kernels (resblock_pair.hip: fp32; resblock_pair16.hip: 16-bit operands).
#pragma once

struct PairArgs {
    const float* x;       // [B][C][ld] input of the pair = residual operand (batch stride bstride)
    float* y;             // [B][C][ld] output (must not alias x: neighbouring tiles read x's halo columns)
    const void* w1f;      // conv1 (k taps, dilation dil) weights in MFMA A-fragment order (fp32 or 16-bit)
    const float* b1;
    const void* w2f;      // conv2 (k taps, dilation 1)
    const float* b2;
    long bstride;
    int B, C, T, ld;
    int k, dil;
    int accum;            // y += result (the MRF sum) instead of y = result
    float slope;          // LeakyReLU slope of both activations (0.1)
    long long* dbg;       // filled by the launcher: cycle stamps (PAIR_DBG == 5 builds only)
};

// One conv of a wide (C = 128 / 256) ResBlock, X-resident (conv_xl_kernel in resblock_pair.hip)
struct ConvXlArgs {
    const float* x;       // [B][C][ld] input (LeakyReLU(slope) applied while staging)
    float* y;             // [B][C][ld] output (must not alias x)
    const float* wf;      // weights, MFMA A fragments in iteration order
    const float* bias;
    const float* res;     // residual, indexed like y (may be null; may alias nothing written by this launch)
    long bstride;
    int B, C, T, ld;
    int k, dil;
    int accum;
    float slope;          // 1 = no activation on the input
    int relu;             // ReLU after the bias (variance-predictor convs, model/modules.py:470-499); 0 for HiFi-GAN
    int cin;              // input channels when they differ from C (0 = C); x then has its own batch stride:
    long xbstride;
    int wino_force;       // cmtts_launch_conv_xlw: take the Winograd form whatever the launch size (tests: goldens are small)
    // cmtts_launch_conv_k5q only (conv_k5q.hip, round 6; zero / null elsewhere):
    const float* ln_g;    // LayerNorm over the 256 input channels of every frame, applied while the tile is staged (null: none)
    const float* ln_b;
    float ln_eps;
    int row_split;        // 0 = the launcher's rule; 1 / 2 / 4 workgroups per frame tile (tests: the result does not depend on it)
};

// ---- Winograd form of a k-tap (dilated) Conv1d for the fp32 X-resident kernels (round 4; conv_xlw_kernel in resblock_pair.hip, weights packed by
// cmtts_api.hip: to_wino_iter_fragments).  Outputs are computed in PAIRS (t, t + dil); with X(m) = act(x)[t + (m - (k-1)/2) dil] the k taps split
// into groups of three consecutive taps done as F(2,3) (4 products per pair instead of 6), a leftover pair of taps as F(2,2) (3 instead of
// 4) and a leftover single tap directly (2):  k = 3: 4 products per pair instead of 6, k = 7: 10 instead of 14, k = 11: 15 instead of 22.
// Every product is one entry of this table: accumulator M_acc += W_kind(tau) * (X(a) + sgn X(b))   (sgn = 0: X(a) alone), and
//   y(t) = (M0 + M1) + M2,   y(t + dil) = (M1 - M2) - M3.
// wkind: 0 g[tau] | 1 (g[tau] + g[tau+1] + g[tau+2]) / 2 | 2 (g[tau] - g[tau+1] + g[tau+2]) / 2 | 3 g[tau+2] | 4 -g[tau] | 5 g[tau] + g[tau+1] | 6 g[tau+1]
struct WinoEntry { signed char acc, a, b, sgn, wkind, tau; };
#define WINO_F23(t) {0, t, (t) + 2, -1, 0, t}, {1, (t) + 1, (t) + 2, 1, 1, t}, {2, (t) + 2, (t) + 1, -1, 2, t}, {3, (t) + 1, (t) + 3, -1, 3, t}
#define WINO_F22(t) {0, t, (t) + 1, -1, 0, t}, {1, (t) + 1, 0, 0, 5, t}, {3, (t) + 1, (t) + 2, -1, 6, t}
#define WINO_ONE(t) {0, t, 0, 0, 0, t}, {3, (t) + 1, 0, 0, 4, t}
template <int KT> struct WinoTab;
template <> struct WinoTab<3> { static constexpr int N = 4; static constexpr WinoEntry e[N] = {WINO_F23(0)}; };
template <> struct WinoTab<7> { static constexpr int N = 10; static constexpr WinoEntry e[N] = {WINO_F23(0), WINO_F23(3), WINO_ONE(6)}; };
template <> struct WinoTab<11> { static constexpr int N = 15; static constexpr WinoEntry e[N] = {WINO_F23(0), WINO_F23(3), WINO_F23(6), WINO_F22(9)}; };

#ifdef __cplusplus
extern "C" {
#endif
int cmtts_launch_conv_xl(const ConvXlArgs* a, void* stream);
// the same conv in its Winograd form (a->wf = to_wino_iter_fragments of the same weights); -2 = shape not covered (C = 128 / 256, k = 3 / 7 / 11, dilation 1 / 3 / 5)
int cmtts_launch_conv_xlw(const ConvXlArgs* a, void* stream);
// the dilation-1 conv in its F(4,3) form (conv_xlq.hip; a->wf = to_wino43_iter_fragments of the same weights); -2 = shape not covered (C = 64 / 128 / 256, k = 3 / 7 / 11)
// or a launch of fewer than 1024 column tiles without a->wino_force
int cmtts_launch_conv_xlq(const ConvXlArgs* a, void* stream);
// a k = 3 pair (conv1 at dilation 1 / 3 / 5, LeakyReLU, conv2, + x [+ the MRF sum]) with both convs in that form and xt kept on the CU (conv_xlq_pair.hip, round 6;
// a->w1f / a->w2f = to_wino43_iter_fragments; within fp32 Winograd rounding of the two conv_xlq launches: its conv1 quads start one frame earlier); -2 = shape not covered (C = 64 / 128, k = 3)
int cmtts_launch_conv_xlq_pair(const PairArgs* a, void* stream);
// Conv1d(cin = 128 | 256 -> 256, k = 5) + bias [+ ReLU] as two F(4,3) tap groups, optional LayerNorm prologue (conv_k5q.hip: the frame-level pitch predictor; a->wf = to_wino43_iter_fragments, taps = 5)
int cmtts_launch_conv_k5q(const ConvXlArgs* a, void* stream);
// HiFi-GAN upsampler (ConvTranspose1d, kernel 2 s, stride s, padding s / 2), all phases in one X-resident launch (resblock_pair.hip)
int cmtts_launch_convT(const float* x, float* y, const float* wf, const float* bias, long xbstride, long ybstride, int B, int cin,
                       int co, int Ti, int To, int ldx, int ldy, int s, float pre_div, float slope, void* stream);
// 16-bit twin (resblock_pair16.hip): io 1 = fp32 in / 16-bit activated out (conv1 of a pair), io 2 = 16-bit in / fp32 out (conv2)
int cmtts_launch_conv_xl16(const ConvXlArgs* a, int mode, int io, void* stream);
// 0 = launched, -2 = shape not covered (the caller runs the two layer-granular launches), -3 = HIP error
int cmtts_launch_resblock_pair(const PairArgs* a, void* stream);
int cmtts_launch_resblock_pair16(const PairArgs* a, int mode, void* stream);   // mode 1 = bf16, 2 = fp16
// the upsampler with 16-bit operands (resblock_pair16.hip): wf16 = 16-bit fragments of the same two-tap stacked weights
int cmtts_launch_convT16(const float* x, float* y, const void* wf16, const float* bias, long xbstride, long ybstride, int B, int cin,
                         int co, int Ti, int To, int ldx, int ldy, int s, float pre_div, float slope, int mode, void* stream);
// a whole ResBlock (three pairs, dilations 1 / 3 / 5) of a C = 64 / 32 stage in one launch, 16-bit operands; -2 = not covered
int cmtts_launch_resblock16(const float* x, float* y, const void* const* w1f, const void* const* w2f, const float* const* b1,
                            const float* const* b2, long bstride, int B, int C, int T, int ld, int k, int accum, float slope,
                            int mode, void* stream);
// wide stages (C = 128): the pair in one launch with ONE in-place 16-bit image, two workgroups per CU (resblock_pairw16.hip)
int cmtts_launch_resblock_pairw16(const PairArgs* a, int mode, void* stream);
// the pair with fp16x3 operands (resblock_pair16x3.hip): w1f / w2f = hi fragment set followed by the lo set; C = 32 / 64 / 128
int cmtts_launch_resblock_pair16x3(const PairArgs* a, void* stream);
// one conv of the C = 256 stage, fp16x3 operands, fp32 in / out (same file)
int cmtts_launch_conv_xl16x3(const ConvXlArgs* a, void* stream);
void cmtts_pair_set_debug(long long* dbg);
int cmtts_xl_set_split(int on);      // conv_xl: m-tiles over several workgroups for launches of a few column tiles (same bits); returns the previous value
#ifdef __cplusplus
}
#endif
