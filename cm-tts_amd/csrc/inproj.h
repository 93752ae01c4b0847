// Arguments of the fused denoiser-input kernel (inproj.hip): c_in scaling + transpose + input projection + halo clearing.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct InProjArgs {
    const float* x;        // [B][T][M] time-major mel (sampler state)
    const float* scale_b;  // per-utterance scale (may be null: `scale`)
    float scale;           // c_in
    const float* wf;       // input_projection weight as MFMA A fragments [M/8][C/32][64][4] (to_fragment_order)
    const float* bias;     // [C]
    float* h;              // [B][C][T] out: relu(W (scale x) + b)
    int B, T, M, C;
    void* zero;            // region to clear (16-byte aligned), or null
    long zero_f4;          // its size in 16-byte units
};

#ifdef __cplusplus
extern "C" {
#endif
int cmtts_launch_inproj(const InProjArgs* a, void* stream);
#ifdef __cplusplus
}
#endif
