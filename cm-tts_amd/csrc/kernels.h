// Launch wrappers of the small HBM-bound kernels (kernels.hip).  All activations are channel-major
// [B][C][ld] with the frame / phoneme axis contiguous; integer tensors are int64 like the
// reference's (torch.long) unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum DenseAct { DENSE_NONE = 0, DENSE_RELU = 1, DENSE_MISH = 2 };

void k_embed_tokens(const int64_t* texts, const int64_t* lens, const float* E, const float* omega, const float* tab,
                    int tab_rows, float* x, int B, int L, int ld, int C, float scale, hipStream_t s);
void k_layernorm_ct(const float* in, float* out, const float* gamma, const float* beta, float eps,
                    const int64_t* lens, int B, int T, int ld, hipStream_t s);          // C = 256
void k_softmax_cols(float* st, const int64_t* lens, int nz, int H, int L, int ld, long zs, hipStream_t s);
void k_add_rowvec(float* x, const float* vec, int B, int C, int L, int ld, hipStream_t s,
                  const int64_t* lens = nullptr);      // lens: columns l >= lens[b] are left untouched
void k_pos_embed_add(const float* x, float* out, const float* alpha, const float* omega, const float* tab,
                     int tab_rows, int B, int C, int T, int ld, hipStream_t s,
                     const int64_t* lens = nullptr);      // lens: columns t >= lens[b] are written as 0
// pos_embed_add on the length-regulated input, gathered on the fly: x[c][t] = mel2ph[t] > 0 ? src[c][mel2ph[t] - 1] : padv[c]
void k_pos_embed_add_lr(const float* src, int ldl, const int64_t* mel2ph, const float* padv, float* out, const float* alpha, const float* omega,
                        const float* tab, int tab_rows, int B, int C, int T, hipStream_t s);
void k_chan_linear(const float* x, const float* W, const float* bias, float* out, const int64_t* lens,
                   int B, int C, int T, int ld, int O, hipStream_t s);
// cwt_stats_layers as one launch (three dense_small<4> layers, same bits); false = shape not covered (the caller runs the three launches)
bool k_stats_mlp(const float* in, long in_bs, long in_ks, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                 const float* b2, float* out, int B, int K0, int N0, int N1, int N2, hipStream_t s);
void k_dense_small(const float* in, long in_bs, long in_ks, const float* Wt, const float* bias,
                   const float* add, float* out, int B, int K, int N, int act, hipStream_t s);
void k_energy_embed(const float* x, const float* e_pred, float* e_scaled, const float* e_target, float e_control,
                    const float* bins, int nbins, const float* E, float* out1, int64_t* e_idx, int B, int C, int L, int ld,
                    hipStream_t s);
void k_durations(const float* logd, float d_control, float* d_rounded, int* cum, int64_t* mel_len,
                 int B, int L, hipStream_t s);
void k_durations_serial(const float* logd, float d_control, float* d_rounded, int* cum, int64_t* mel_len,
                 int B, int L, hipStream_t s);
void k_broadcast_row(const float* row, float* out, int B, int n, hipStream_t s);      // out[b][:] = row[:]
void k_add_rows(const float* a, const float* b, float* out, long n, hipStream_t s);     // out = a + b
// out = mask(sum_s part[b][s] (ascending) + bias + res); part: [B][nseg][C][ld]
void k_reduce_partials(const float* part, int nseg, const float* bias, const float* res, const int64_t* lens, float* out, int B, int C,
                       int L, int ld, hipStream_t s);
// LayerNorm(256 channels) + Linear(256 -> O) in one launch, O in {1, 10, 11} (false: not covered); out is time-major [B][T][O]
// ln_linear (O = 1) + bucketize + out1 = xin + energy_embedding[bucket] in one launch (the energy predictor's head; same bits as k_ln_linear + k_energy_embed)
void k_ln_linear_energy(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias, float* out,
                        const int64_t* ln_lens, const int64_t* out_lens, int B, int T, int ld, const float* xin, const float* e_target, float e_control,
                        const float* bins, int nbins, const float* E, float* out1, int64_t* e_idx, float* e_scaled, hipStream_t s);
bool k_ln_linear(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias, float* out,
                 const int64_t* ln_lens, const int64_t* out_lens, int B, int T, int ld, int O, hipStream_t s);
void k_cumsum_durations(const float* dur, int* cum, int64_t* mel_len, int B, int L, hipStream_t s);
void k_mel2ph(const int* cum, int64_t* mel2ph, int B, int L, int T, hipStream_t s);
void k_length_regulate(const float* out1, const int64_t* mel2ph, float* xlr, int B, int C, int ldl,
                       int T, hipStream_t s, const float* padv = nullptr);     // padv [C]: value of padding frames (default 0)
void k_pitch_index(const float* cwt, int O, const float* mean_p, const float* std_p, int stat_ld, float std_scale,
                   const float* uv_logit, int uv_ld, const uint8_t* uv_mask, float eps, float* r_ws, int64_t* p_idx,
                   float* f0_denorm, int B, int T, hipStream_t s);
void k_gather_add(const float* x, const int64_t* idx, const float* E, float* out, int B, int C, int T,
                  hipStream_t s);
// gather_add with the length regulator's gather inside: out = (mel2ph > 0 ? out1[..][mel2ph - 1] : 0) + E[idx]
void k_lr_gather_add(const float* out1, const int64_t* mel2ph, int ldl, const int64_t* idx, const float* E, float* out, int B, int C, int T,
                     hipStream_t s);
void k_mel_prep(const float* x, const float* scale_b, float scale, float* hin, int B, int T, int M, hipStream_t s);
void k_mel_post(const float* F, const float* xold, const float* noise, float c_out, float c_skip,
                float nstd, float* out, int B, int T, int M, unsigned* flag, hipStream_t s);
void k_diff_embed(const float* t, const float* omega, float* emb, int B, int C, hipStream_t s);
extern int g_post_v4;            // conv_post with 16-byte loads (kernels.hip; same bits; internal switch "post_v4")
void k_conv_post(const float* x, const float* w, const float* bias, float pre_div, float slope, float* wav, int B,
                 int C, int T, int ld, int KW, hipStream_t s);
void k_wav_to_int16(const float* wav, int16_t* pcm, long n, float max_wav, hipStream_t s);
void k_length_mask(const int64_t* lens, uint8_t* mask, int B, int W, hipStream_t s);    // mask[b][t] = t >= lens[b]
void k_transpose(const float* in, float* out, int B, int R, int Cn, hipStream_t s);   // [B][R][Cn] -> [B][Cn][R]
void k_fill_lens(int64_t* lens, int64_t v, int B, hipStream_t s);
void k_gather_rows(const float* table, const int64_t* idx, float* out, int B, int C, int n_rows, hipStream_t s);
void k_copy_rows(float* dst, int dst_ld, const float* src, int src_ld, int width, long rows, hipStream_t s);
void k_scale(const float* in, float* out, long n, float sc, hipStream_t s);
void k_fill_float(float* p, float v, int n, hipStream_t s);
