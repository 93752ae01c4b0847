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
};

#ifdef __cplusplus
extern "C" {
#endif
int cmtts_launch_conv_xl(const ConvXlArgs* a, void* stream);
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
