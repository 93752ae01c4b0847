// Arguments of the fused denoiser residual-block kernel (resblock_fused.hip).
#pragma once

struct ResArgs {
    const float* x_in;    // [B][256][T]
    const float* cp;      // this layer's slice of the precomputed conditioner projection, batch stride cp_bstride
    const float* dp;      // [B][vec_stride]: diffusion (+ speaker) projection of this layer, d + p
    const float* d;       // [B][vec_stride]: diffusion projection alone (residual = x + d)
    float* x_out;         // [B][256][T]  (must not alias x_in: neighbouring tiles read its halo)
    float* skip;          // [B][256][T]
    const float *W3, *b3; // conv_layer              k-major [3][256][512], gate-permuted rows
    const float *Wo, *bo; // output_projection       k-major [256][512]
    long cp_bstride;
    long vec_stride;
    int B, T;
    int accum_skip;       // skip += o[C:] (layers > 0) or skip = o[C:] (layer 0)
    int stagger_mode;     // 0 none; 1: second half of the grid; 2: odd workgroups — delayed start (experiment)
    int stagger_sleeps;
    unsigned* cu_arrivals; // [2048] monotonically increasing per-CU arrival counters (stagger mode 3)
    long long* dbg;       // optional [grid][8] s_memtime stamps written by wave 0 (phase timing)   // number of s_sleep 127 (~3.4 us each) for the delayed workgroups
};

#ifdef __cplusplus
extern "C" {
#endif
int cmtts_launch_resblock(const ResArgs* a, void* stream);
void cmtts_resblock_set_stagger(int mode, int sleeps);
void cmtts_resblock_set_debug(long long* dbg);
#ifdef __cplusplus
}
#endif
