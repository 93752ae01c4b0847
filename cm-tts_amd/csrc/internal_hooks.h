// NOT part of the C ABI (include/cmtts_hip.h): A/B switches between a fused kernel and the path it replaces, for the bitwise
// cross-check tests (tests/test_gpu_parity.py) and the measurement tools (tools/).  Every pair produces identical bits EXCEPT:
// attn_fused for L > 192 (the key-chunked online softmax against the two-pass softmax of the three-launch path: ~1e-7),
// attn_fused for L <= 192 against the three-launch path (different fp32 summation order: <= 2e-5 on the encoder output),
// cond_gemm16 (16-bit against fp32 operands of the conditioner GEMM in bf16 / fp16 / fp16x3 models: a numerics switch — since
// round 3 the 16-bit form is the default for EVERY shape of a 16-bit model, which changed those models' default numerics against
// round 2), persist_wino (the fp32 persistent stack's k = 3 conv in a Winograd form — the process-wide A/B twin of the model option "winograd":
// 3 (default since round 5) F(4,3), <= 1.8e-5 per network evaluation against the direct form; 1 F(2,3) (round 4), <= 1e-5; 2 runs as 1 (round 5's
// one-wave-per-SIMD F(2,3) stack left the build in round 6: tools/attic/); 0 direct), ffn_wino (the FFT blocks' k = 9 FFN conv as Winograd tap groups in the fused launch: 1 (default
// since round 6) F(2,3) over output pairs, ~2.5e-6 on the encoder output against the direct form; 2 F(4,3) over quads (round 5), ~4.5e-6; 0 direct), pred_wino (round 6: the frame-level
// pitch predictor's k = 5 convs as F(4,3) tap groups, conv_k5q.hip, at every launch size: ~7e-6 on the cwt output; 0 = the direct kernels), voc_wino43 (round 5: the Winograd path's convs as F(4,3) tap groups over output quads, conv_xlq_kernel: 1 (default)
// dilation 1 and 3 everywhere + dilation 5 at C = 256 or k = 3, 2 only dilation 1, 3 every conv, 0 none), voc_wino (round 4: the fp32 generator's C >= 128 ResBlock
// convs in their Winograd form — 0 never, 1 (default) launches of >= 1024 column tiles, 2 always; <= 1.2e-6 on the waveform), voc_qpair (round 6: the k = 3 pairs of the
// C = 64 / 128 stages of that path as ONE fused F(4,3) launch, conv_xlq_pair.hip — 0 the forms it replaces (C = 128: two conv_xlq launches; C = 64: the direct pair kernel), 1 (default)
// with voc_wino's launch-size rule, 2 always; <= 1e-6 on the waveform between the arms).  The switches are process-wide and unsynchronised.  Exported from libcmtts_hip.so so that ctypes can reach it
// (cmtts_amd/_lib.py: internal_set).
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
// Returns the previous value (a value outside the switch's range only queries) or CMTTS_E_INVALID for an unknown name.
// Names: cond_gemm, persist_tail, inproj_fused, ffn_xres, ffn_fused, text_xres, attn_fused, pred_xl, pred_head, voc_pair,
// voc_pair3, voc_pairw, voc_pair128, voc_rb16, voc_xl, voc_xl16, voc_upsT, post_v4, cond_gemm16, cond_factored, cond_inkernel, xres_small, pred_xres, cwt_in_phoneme, voc_xl_split, text_xt16,
// persist_wino, ffn_wino, pred_wino, voc_wino, voc_wino43, voc_wino64, voc_wino64_k, voc_qpair, xres_nt, and (round 6, same bits on / off) attn_qb (attention with the queries split over workgroups),
// stats_mlp (cwt_stats_layers as one launch), energy_head (energy bucketize + embedding add inside the energy predictor's head launch)   (cmtts_api.hip: cmtts_internal_set).
int cmtts_internal_set(const char* name, int value);
// Test hook: the stacked conditioner projections alone, with the model's current precision mode.  cond_ct [B][hidden][T] -> cp [B][NL * C][T]
// (device pointers).  Returns a cmtts_status.
struct cmtts_model;
int cmtts_internal_cond_projections(struct cmtts_model* m, const float* cond_ct, int B, int T, float* cp, void* stream);
// Test hook: the same tensor expanded from the factors cmtts_frame_forward_sub returns (cond_p1 [B][NL*C][p1_ld], mel2ph, p_idx [B][T])
int cmtts_internal_cond_factored(struct cmtts_model* m, const float* p1, int p1_ld, int L, const int64_t* mel2ph, const int64_t* p_idx, int B, int T,
                                 float* cp, void* stream);
#ifdef __cplusplus
}
#endif
