// Arguments of the fused self-attention kernel (attention.hip).
#pragma once
#include <stdint.h>

struct AttnArgs {
    const float* qkv;     // [B][3*H*dh][ld] channel-major: rows [0, H*dh) = Q, [H*dh, 2*H*dh) = K, [2*H*dh, 3*H*dh) = V
    float* out;           // [B][H*dh][ld] channel-major
    const int64_t* lens;  // [B] valid keys per utterance (key padding mask)
    long bstride, obstride;
    int B, H, dh, L, ld;
    float scale;          // 1 / sqrt(dh), applied to the scores before the softmax
};

#ifdef __cplusplus
extern "C" {
#endif
// 0 = launched, -2 = shape not covered (L > 192, dh != 128), -3 = HIP error
int cmtts_launch_attention(const AttnArgs* a, void* stream);
int cmtts_attention_set_qb(int on);         // internal switch "attn_qb": 1 (default) = L <= 128 with the queries split over workgroups (attention_qb_kernel; same bits); returns the previous value
#ifdef __cplusplus
}
#endif
