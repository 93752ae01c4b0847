// Arguments of the fused denoiser residual-block kernel (resblock_fused.hip).
#pragma once

struct ResArgs {
    const float* x_in;    // [B][256][T]
    const float* cp;      // this layer's slice of the precomputed conditioner projection, batch stride cp_bstride
    const float* dp;      // [B][vec_stride]: diffusion (+ speaker) projection of this layer, d + p
    const float* d;       // [B][vec_stride]: diffusion projection alone (residual = x + d)
    float* x_out;         // [B][256][T]  (must not alias x_in: neighbouring tiles read its halo)
    float* skip;          // [B][256][T]
    const float *W3f, *b3; // conv_layer, MFMA A-fragment order [3][32 k-groups][16 m-tiles][64 lanes][4], gate-permuted rows
    const float *Wof, *bo; // output_projection, fragment order [32][16][64][4]
    long cp_bstride;
    long vec_stride;
    int B, T;
    int accum_skip;       // skip += o[C:] (layers > 0) or skip = o[C:] (layer 0)
    float* z;             // split form only (resblock_split.hip): scratch [B][256][T] for the gated activations
    long long* dbg;       // optional [grid][8] cycle stamps written by wave 0 (phase timing, tools/phase_timing.py)
    unsigned* flag;       // optional pinned error word (16-bit kernel, fp16 mode): set to 3 when a conv input leaves the fp16 range
};

#ifdef __cplusplus
extern "C" {
#endif
int cmtts_launch_resblock(const ResArgs* a, void* stream);
int cmtts_launch_resblock_split(const ResArgs* a, void* stream);         // fp32, two launches, small batches (needs a->z)
int cmtts_launch_resblock_lp(const ResArgs* a, int mode, void* stream);   // mode 1 = bf16, 2 = fp16 operands
void cmtts_resblock_set_tile(int frames);   // 0 = automatic, 32 or 64 = forced frames per workgroup
void cmtts_resblock_set_debug(long long* dbg);
#ifdef __cplusplus
}
#endif
