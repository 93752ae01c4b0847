/* cmtts_hip.h — C ABI of the MI355X-native CM-TTS inference hot path (libcmtts_hip.so).
 *
 * The reference (XiangLi2022/CM-TTS) is pure Python/PyTorch and has no FFI; the path sits behind
 * Python callables.  Each entry point below replaces one of those callables and
 * cites it (paths relative to the reference root).  Conventions:
 *   - plain pointers and sizes only, no torch types; every tensor pointer is a DEVICE pointer unless
 *     the parameter is documented as host;
 *   - caller-allocated outputs, caller-provided workspaces (size queries below), no hidden
 *     allocation and no host synchronisation on the hot path;
 *   - `stream` is a hipStream_t passed as void*; the result is ordered on it.  Independent branches of a call may run
 *     on library-owned side streams that are forked from and joined back into `stream` with events before the call
 *     returns control of the data (cmtts_set_option("branch_streams", 0) keeps everything on `stream`);
 *   - like the reference (one Python thread, synthesize.py), the library is NOT thread-safe: process-wide options,
 *     the error string and the persistent-launch admission table are unsynchronised — one host thread per process
 *     (one process per GPU), any number of streams;
 *   - return 0 on success, negative on error (CMTTS_E_*), message via cmtts_last_error();
 *   - integer tensors are int64 (torch.long) like the reference's; activations fp32;
 *   - frame-level activations cross the boundary channel-major ("_ct": [B, C, T], T contiguous) —
 *     the layout the reference itself transposes to before Conv1d (tts_net.py:31-32);
 *     cmtts_transpose() converts to/from the reference's [B, T, C].
 */
#ifndef CMTTS_HIP_H
#define CMTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMTTS_OK 0
#define CMTTS_E_INVALID (-1)      /* bad argument / missing tensor */
#define CMTTS_E_UNSUPPORTED (-2)  /* shape outside what the kernels support */
#define CMTTS_E_HIP (-3)          /* HIP runtime error */
#define CMTTS_E_WORKSPACE (-4)    /* workspace too small */

/* Supported shapes: the kernels are specialised for what the reference's three configs use — hidden = res_channels = 256,
 * 2 attention heads of 128 channels, 80 mel bins (<= 96), <= 32 residual layers, HiFi-GAN V1 (hifigan/config.json).  Anything
 * else returns CMTTS_E_UNSUPPORTED from cmtts_create / cmtts_finalize / the forward calls — never a silent fallback. */
typedef struct cmtts_model cmtts_model;      /* CMTotalTTS weights, packed for the kernels */
typedef struct cmtts_vocoder cmtts_vocoder;  /* hifigan.Generator weights */

/* Hyper-parameters: config/<dataset>/{model,preprocess,train}.yaml of the reference. */
typedef struct cmtts_config {
    int32_t n_symbols, hidden, enc_layers, enc_heads, ffn_kernel;
    int32_t pred_filter, pred_layers, pred_kernel, dur_layers, dur_kernel, cwt_hidden;
    int32_t pitch_bins, energy_bins, use_uv, multi_speaker, external_speaker_dim;
    int32_t n_speaker;   /* > 0: preprocess.yaml speaker_embedder "none" — speaker_emb is nn.Embedding(n_speaker, hidden) indexed by
                            `speakers` (model/cmtts.py:26-38,77-78); 0: nn.Linear(external_speaker_dim, hidden) of spker_embeds */
    int32_t n_mels, res_layers, res_channels;
    float cwt_std_scale, pitch_norm_eps;
    float sigma_min, sigma_max, sigma_data, rho;
} cmtts_config;

const char* cmtts_last_error(void);
const char* cmtts_version(void);
/* Binary-interface revision of this header: bumped whenever a struct layout or an existing signature changes (round 2
 * inserted cmtts_config.n_speaker and the `speakers` parameter of cmtts_text_forward: revision 2; round 3 adds entry
 * points only but starts the counter: 3; round 4 appends the conditioner factors to cmtts_sample_group: 4;
 * round 6: 5 — no layout change, but the MEANING of an existing option value changed in round 5 without a bump: cmtts_model_set_option(m, "winograd", 1) selects
 * the F(4,3) form since then and 2 the F(2,3) form that 1 used to select, and cmtts_poll_error's codes 2 / 3 no longer fail the next launch).  A host compares cmtts_abi_version() with the CMTTS_ABI_VERSION it was built
 * against before it passes a struct (cmtts_amd/_lib.py does at load time). */
#define CMTTS_ABI_VERSION 5
int cmtts_abi_version(void);

/* ---- weight import: replaces torch.load + load_state_dict (synthesize.py:79-83).
 * cmtts_set_tensor takes one entry of CMTotalTTS.state_dict() under its original key and in its
 * original layout (HOST pointer); cmtts_finalize re-packs (k-major / transposed / gate-permuted)
 * and uploads.  Unknown keys (e.g. *_float_tensor buffers) are ignored. */
int cmtts_create(const cmtts_config* cfg, cmtts_model** out);
int cmtts_set_tensor(cmtts_model* m, const char* name, const float* host_data, const int64_t* shape, int ndim);
int cmtts_finalize(cmtts_model* m);
void cmtts_destroy(cmtts_model* m);

/* ---- DurationPitchSpeakerNet.forward, phoneme-level half (model/cmtts.py:44-122 up to the
 * duration rounding, model/modules.py:331-372): text encoder, speaker projection, duration and
 * energy predictors, durations, cumulative sums.  The phoneme-level state needed by
 * cmtts_frame_forward stays in `text_ws`.  Optional outputs may be NULL.
 *   texts int64 [B,L] (0 = pad), src_lens int64 [B], spker_embeds fp32 [B,external_speaker_dim]
 *   (multi-speaker with an external embedder, else NULL), speakers int64 [B] (multi-speaker with n_speaker > 0:
 *   rows of the speaker_emb table, else NULL / ignored).
 *   out: log_d fp32 [B,L], d_rounded fp32 [B,L], mel_len int64 [B], e_pred fp32 [B,L],
 *        e_idx int64 [B,L], enc_out_ct fp32 [B,hidden,L], speaker_emb fp32 [B,hidden]. */
size_t cmtts_text_workspace_bytes(const cmtts_model* m, int B, int L);
int cmtts_text_forward(cmtts_model* m, const int64_t* texts, const int64_t* src_lens, const float* spker_embeds,
                       const int64_t* speakers, int B, int L, float d_control,
                       float* log_d, float* d_rounded, int64_t* mel_len, float* e_pred, int64_t* e_idx,
                       float* enc_out_ct, float* speaker_emb,
                       void* text_ws, size_t text_ws_bytes, void* stream);
/* The same for a RAGGED batch (round 4; BASELINE.json configs[3]): utterances of several padded groups — the bucket groups of a shard —
 * in one call, padded to the longest group's L.  pad_lens int64 [B] (device): the padded phoneme count of each utterance's OWN group;
 * columns l >= pad_lens[b] do not exist for utterance b.  The padded length enters the reference's arithmetic in three places:
 * LayerNorm2 of an FFT block turns a masked (zero) column into its bias vector, which the k = 9 FFN conv reads up to four columns beyond
 * src_len (model/blocks.py:612-615, 539-546); the speaker vector is added to every column of the padded batch
 * (model/modules.py:349-352); the energy predictor runs unmasked over them (:520-554).  All three stop at pad_lens[b]; everything else
 * is column-local or masked by src_lens, so every utterance gets the bits of running its group alone through cmtts_text_forward.  Outputs are [B, L] with L the call's; a group's [Bg, Lg] block is the
 * sub-array rows b0.., columns < Lg (columns >= pad_lens[b] hold unspecified values).  pad_lens NULL = cmtts_text_forward. */
int cmtts_text_forward_ragged(cmtts_model* m, const int64_t* texts, const int64_t* src_lens, const int64_t* pad_lens,
                              const float* spker_embeds, const int64_t* speakers, int B, int L, float d_control,
                              float* log_d, float* d_rounded, int64_t* mel_len, float* e_pred, int64_t* e_idx,
                              float* enc_out_ct, float* speaker_emb,
                              void* text_ws, size_t text_ws_bytes, void* stream);

/* ---- VarianceAdaptor controls and teacher-forced targets (model/modules.py:331-343: p_control, e_control,
 * pitch_target, energy_target, duration_target; d_control is an argument of cmtts_text_forward).  The
 * settings stay on the model until replaced; NULL restores the inference defaults (controls 1, no
 * targets).  All pointers are DEVICE pointers that must stay valid while the forward calls run.
 *   d_target fp32 [B,L]: durations used instead of the predicted ones (:365-367)
 *   e_target fp32 [B,L]: energies bucketized instead of prediction * e_control (:318-328)
 *   cwt_spec fp32 [B,T,10] + f0_mean, f0_std fp32 [B] + uv u8 [B,T]: pitch target (:379-390; f0 from
 *   cwt2f0_norm utils/pitch_tools.py:268-273 with the target statistics; uv only read when use_uv)
 * The predictors still run and their outputs are returned, like the reference. */
typedef struct cmtts_variance_controls {
    float p_control, e_control;
    const float* d_target;
    const float* e_target;
    const float* cwt_spec;
    const float* f0_mean;
    const float* f0_std;
    const uint8_t* uv;
} cmtts_variance_controls;
int cmtts_set_variance_controls(cmtts_model* m, const cmtts_variance_controls* vc);

/* ---- frame-level half (model/modules.py:373-412; LengthRegulator :415-448; dur_to_mel2ph
 * utils/tools.py:768-798; get_pitch_embedding cwt branch :259-317).  T = padded frame count chosen
 * by the host (max(mel_len) like the reference, or a static bucket).
 *   out: cond_ct fp32 [B,hidden,T], mel2ph int64 [B,T], cwt_out fp32 [B,T,10|11] (opt),
 *        f0_denorm fp32 [B,T] (opt), p_idx int64 [B,T] (opt), f0_stats fp32 [B,2] (opt). */
size_t cmtts_frame_workspace_bytes(const cmtts_model* m, int B, int T);
int cmtts_frame_forward(cmtts_model* m, const void* text_ws, int B, int L, int T,
                        float* cond_ct, int64_t* mel2ph, float* cwt_out, float* f0_denorm, int64_t* p_idx,
                        float* f0_stats, void* frame_ws, size_t frame_ws_bytes, void* stream);
/* The same for the sub-batch [b0, b0 + B) of a text workspace filled for B_all utterances padded to L_all phonemes (a bucket group of a
 * ragged shard with its own padded frame count T), and — optionally — with the phoneme-level factor of the conditioner projections:
 *   cond_p1 fp32 [B, res_layers * res_channels, L_all rounded up to 4] (or NULL) = conditioner_projection weights of all residual
 *   layers (model/blocks.py:663,676) applied to the phoneme-level conditioning, no bias.  Since
 *   cond[:, t] = out1[:, mel2ph[t] - 1] + pitch_embed[p_idx[t]] (model/modules.py:373-395), the projections of the frames are
 *   W out1 [:, mel2ph - 1] + (W pitch_embed^T + b)[:, p_idx]: cmtts_sample_factored / cmtts_sample_group expand them from cond_p1,
 *   mel2ph and p_idx instead of running the stacked GEMM over the frames (7 instead of 43 GFLOP per 32 x 512-frame batch; fp32 rounding
 *   differs from the dense product at the 1e-7 level). */
int cmtts_frame_forward_sub(cmtts_model* m, const void* text_ws, int B_all, int L_all, int b0, int B, int T,
                            float* cond_ct, int64_t* mel2ph, float* cwt_out, float* f0_denorm, int64_t* p_idx,
                            float* f0_stats, float* cond_p1, void* frame_ws, size_t frame_ws_bytes, void* stream);
/* cmtts_frame_forward_sub that also returns the phoneme-level factor with the CHANNELS contiguous (round 6):
 *   cond_p1t fp32 [B, res_layers, p1_ld = L_all rounded up to 4, res_channels] (or NULL; needs cond_p1) — the layout the persistent denoiser's
 *   factored instances gather from (16-byte loads of four consecutive channels).  Given here it is written on the library's branch
 *   stream beside the frame-level predictors (a bandwidth-bound copy under MFMA-bound convs); cmtts_sample_factored_t then takes it as it
 *   is, where cmtts_sample_factored transposes cond_p1 at the sampler's entry, in front of the first evaluation (38 us per call at
 *   B = 32 x 88 phonemes).  The same values either way: bit-identical mels. */
int cmtts_frame_forward_sub_t(cmtts_model* m, const void* text_ws, int B_all, int L_all, int b0, int B, int T,
                              float* cond_ct, int64_t* mel2ph, float* cwt_out, float* f0_denorm, int64_t* p_idx,
                              float* f0_stats, float* cond_p1, float* cond_p1t, void* frame_ws, size_t frame_ws_bytes, void* stream);

/* ---- length regulator alone (LengthRegulator.forward, model/modules.py:446-448): bit-exact
 * gather x_ct [B,C,L] -> out_ct [B,C,T] given fp32 durations [B,L]; also emits mel2ph and mel_len.
 * scratch_cum: int32 [B,L]. */
int cmtts_length_regulate(const float* x_ct, const float* durations, int B, int C, int L, int T,
                          float* out_ct, int64_t* mel2ph, int64_t* mel_len, int32_t* scratch_cum, void* stream);

/* ---- CMDenoiserTTS.forward(x, timesteps, conditioner, speaker_emb, mask) (tts_net.py:29-37;
 * Denoiser.forward model/modules.py:600-639).  x fp32 [B,1,T,80] (already scaled by c_in),
 * timesteps fp32 [B] (= 250 ln sigma), cond_ct fp32 [B,hidden,T], speaker_emb [B,hidden] or NULL,
 * out fp32 [B,1,T,80].  `mask` is ignored by the reference and is not a parameter. */
size_t cmtts_denoiser_workspace_bytes(const cmtts_model* m, int B, int T);
int cmtts_denoiser_forward(cmtts_model* m, const float* x, const float* timesteps, const float* cond_ct,
                           const float* speaker_emb, int B, int T, float* out,
                           void* ws, size_t ws_bytes, void* stream);

/* ---- karras_sample_tts (karras_diffusion.py:480-577) with sampler onestep (:801-811) for
 * n_steps = 1 and stochastic_iterative_sampler (:830-854) as synthesize.py:111-147 calls it for
 * n_steps = 2, 4; KarrasDenoiser.denoise (:392-407) with boundary-condition scalings fused.
 * The noise is an input (random_util.py:17-25 draws it with th.randn on the reference device):
 *   noise fp32 [n_noise,B,1,T,80] N(0,1): noise[0] -> x_T, noise[1+i] -> re-noise after eval i
 *   (n_noise = 1 for n_steps = 1, else n_steps + 1).
 *   sigmas, renoise_std: HOST fp32 [n_steps] (cmtts_schedule fills them).
 *   out mel fp32 [B,T,80]. */
int cmtts_schedule(const cmtts_model* m, int n_steps, float* sigmas_host, float* renoise_std_host);
int cmtts_sample(cmtts_model* m, const float* noise, const float* cond_ct, const float* speaker_emb,
                 int B, int T, int n_steps, const float* sigmas_host, const float* renoise_std_host,
                 float* mel, void* ws, size_t ws_bytes, void* stream);
/* cmtts_sample for conditioning that cmtts_frame_forward_sub produced together with its phoneme-level factor: cond_p1 / p1_ld / L /
 * mel2ph / p_idx as described there.  cond_ct stays required (16-bit models and the unfused path read it); cond_p1 NULL = cmtts_sample. */
int cmtts_sample_factored(cmtts_model* m, const float* noise, const float* cond_ct, const float* speaker_emb,
                          int B, int T, int n_steps, const float* sigmas_host, const float* renoise_std_host,
                          float* mel, void* ws, size_t ws_bytes, void* stream,
                          const float* cond_p1, int p1_ld, int L, const int64_t* mel2ph, const int64_t* p_idx);
/* ... with cmtts_frame_forward_sub_t's channel-contiguous copy of the factor (cond_p1t NULL = cmtts_sample_factored). */
int cmtts_sample_factored_t(cmtts_model* m, const float* noise, const float* cond_ct, const float* speaker_emb,
                            int B, int T, int n_steps, const float* sigmas_host, const float* renoise_std_host,
                            float* mel, void* ws, size_t ws_bytes, void* stream,
                            const float* cond_p1, const float* cond_p1t, int p1_ld, int L, const int64_t* mel2ph, const int64_t* p_idx);

/* ---- the same sampler for a RAGGED shard (BASELINE.json configs[3]: variable-length utterances dealt into static frame buckets;
 * new work — the reference synthesizes one padded batch at a time, synthesize.py:195-227).  Every group is one padded (B, T) batch
 * with its own noise / conditioning / output / workspace (cmtts_denoiser_workspace_bytes(m, B, T)) exactly as cmtts_sample takes
 * them, and its results are those of cmtts_sample on that batch (outputs are defined per padded bucket: model/modules.py:429-430
 * via model/cmtts.py:61-62); but the residual layers of ALL groups run in ONE persistent launch per evaluation, so that
 * buckets too small to fill 256 CUs fill them together.
 *   active_frames: optional HOST int64 [B] (NULL = every frame of the padded batch): the frames of each utterance the caller
 *   will use (its mel_len).  With it the utterance is only computed on its first
 *   ceil((active_frames + tail_frames + res_layers * (n_steps - i)) / 64) 64-frame tiles in evaluation i — an output frame
 *   depends on res_layers frames of input to either side, so every frame below active_frames + tail_frames comes out
 *   BIT-IDENTICAL to the untrimmed result in the direct and F(2,3) forms of the stack, and within fp32 rounding of it (<= 1.5e-5) in the
 *   default F(4,3) form (model option "winograd" below: a quad of frames is rounded from all six inputs of the quad); frames beyond the computed range are unspecified padding — zeros in the one-launch
 *   form, the untrimmed values when the call falls back to cmtts_sample per group (the reference fills them with denoised
 *   padding that every caller slices off: utils/tools.py:575-576, utils/model.py:199-203).
 *   tail_frames: frames beyond active_frames that must still be exact (the receptive field of whatever consumes the padded
 *   mel: 16 covers hifigan.Generator; 0 for mel-only use).
 * Falls back to one cmtts_sample per group when the one-launch form does not apply (16-bit precision modes, tiny shards). */
typedef struct cmtts_sample_group {
    const float* noise;          /* [n_noise,B,1,T,80] */
    const float* cond_ct;        /* [B,hidden,T] */
    const float* speaker_emb;    /* [B,hidden] or NULL */
    int32_t B, T;
    const int64_t* active_frames;/* HOST [B] or NULL */
    float* mel;                  /* out [B,T,80] */
    void* ws;
    size_t ws_bytes;
    /* round 4 (ABI 4), all optional (NULL / 0 = the dense conditioner GEMM on cond_ct): the factors cmtts_frame_forward_sub returned */
    const float* cond_p1;        /* [B, res_layers * res_channels, p1_ld] */
    int32_t p1_ld, L;            /* row stride of cond_p1 (L_all rounded up to 4) and the phoneme count mel2ph indexes into */
    const int64_t* mel2ph;       /* [B,T] */
    const int64_t* p_idx;        /* [B,T] */
} cmtts_sample_group;
int cmtts_sample_ragged(cmtts_model* m, const cmtts_sample_group* groups, int n_groups, int n_steps,
                        const float* sigmas_host, const float* renoise_std_host, int tail_frames, void* stream);

/* ---- hifigan.Generator (hifigan/models.py:112-174) + get_vocoder weight handling
 * (utils/model.py:155-184; weights with weight-norm already folded). */
int cmtts_vocoder_create(cmtts_vocoder** out);
int cmtts_vocoder_set_tensor(cmtts_vocoder* v, const char* name, const float* host_data, const int64_t* shape, int ndim);
int cmtts_vocoder_finalize(cmtts_vocoder* v);
void cmtts_vocoder_destroy(cmtts_vocoder* v);
size_t cmtts_vocoder_workspace_bytes(const cmtts_vocoder* v, int B, int T);
/* Generator.forward: mel_ct fp32 [B,80,T] -> wav fp32 [B,1,256*T] */
int cmtts_vocoder_forward(cmtts_vocoder* v, const float* mel_ct, int B, int T, float* wav,
                          void* ws, size_t ws_bytes, void* stream);
/* vocoder_infer's cast (utils/model.py:195-198): pcm = (wav * max_wav_value).astype(int16) */
int cmtts_wav_to_int16(const float* wav, int16_t* pcm, int64_t n, float max_wav_value, void* stream);

/* ---- measurement hook (no reference counterpart; the reference's only perf tooling is the
 * wall-clock Timer of p_rtf_cm.py:64-108): HIP events recorded on the launch stream around every
 * stride-th launch of the dominant kernel (the fused residual block of the denoiser) between begin and
 * end, at most max_launches of them.  end() synchronises on the events and returns the summed duration
 * and the number of launches measured.  (Two event records per launch cost ~4 % of a cfg2 step when every
 * launch is bracketed; bench.py samples one launch in seven, which visits every layer.) */
int cmtts_profile_begin(int max_launches, int stride);
/* Selects the fused one-kernel-per-layer residual block (default, 1) or the three-launch form (0) of
 * the denoiser; both are bitwise identical (tests) — the switch exists for A/B measurement.
 * Returns the previous setting. */
int cmtts_set_fused_resblock(int on);
/* Residual layers of the denoiser as ONE persistent launch per utterance chunk (denoiser_persist.hip: the tile's
 * residual stream and skip sum stay in registers for all layers, neighbouring tiles exchange their edge columns
 * in-kernel) instead of one launch per layer.  mode 0 = never, 1 = when it pays (default: more 64-frame
 * tiles than half the CUs, i.e. the per-layer kernels would need a second round of 32-frame tiles), 2 = whenever the shape is supported (an utterance has at most as many 64-frame tiles as the
 * GPU has CUs).  Bitwise identical to the per-layer kernels (tests); fp32 operands only.  Any other value only
 * queries.  Returns the previous mode. */
int cmtts_set_persistent_denoiser(int mode);
/* Asynchronous failures: one pinned host word that kernels set, read and cleared (one atomic exchange) by cmtts_poll_error(): 0, or
 * CMTTS_E_HIP with the message in cmtts_last_error().  Codes:
 *   1  the persistent launch bounds every wait for a neighbouring tile (~2 s); one expired: the kernel poisoned the affected utterance
 *      (its mel comes out NaN, never plausible-but-wrong).  The chip did not hold the whole grid: the NEXT denoiser call checks for
 *      this code before it launches and fails instead.
 *   2  a denoiser evaluation produced a non-finite mel value (any precision mode: the sampler's post-scaling sees every output element).
 *   3  a conv input of the fp16 / fp16x3 residual blocks left the fp16 range (|u| > 65504): that mel is finite and wrong.
 * Codes 2 / 3 describe ONE earlier request's numerics; the word is process-wide (not per model or stream), so they are reported by
 * cmtts_poll_error() only and never fail an unrelated later call (round 6; round 5 refused the next launch for them as well).
 * The word reflects launches that have COMPLETED, so call cmtts_poll_error() after synchronising the stream the mel was produced on
 * (host.py does, at every point where it hands host-visible data back: vocoder_infer, synthesize(sync=True), host.synchronize()).
 * There is no counterpart in the reference (its errors are Python exceptions). */
int cmtts_poll_error(void);
/* Process-wide scheduling options.  None of them changes a result bit (tested); they are per process, not per model or
 * stream.  Returns the previous value (a value outside the option's range only queries) or CMTTS_E_INVALID for an
 * unknown name.
 *   "branch_streams"      1 (default) / 0: independent branches of a call on library-owned side streams / all on `stream`
 *   "resblock_split"      fp32 residual block as two launches over 4x the CUs: 0 never, 1 (default) small batches, 2 always
 *   "step_cache"          1 (default) / 0: cmtts_sample keeps the timestep-only rows of the step embedding on the device
 *   "cooperative_launch"  persistent denoiser through hipLaunchCooperativeKernel: 0 never, 1 every launch, 2 (default)
 *                         automatic — once "process_group" is set the first launch of every (variant, grid) is cooperative
 *                         (the runtime validates that the grid is co-resident and fails the launch otherwise), later launches
 *                         of a validated shape are plain (every cooperative launch drains all queues of the device: +24 % per
 *                         step next to RCCL, profiles/r03_cooperative.md)
 *   "process_group"       1: a communicator / process group exists in this process (set by cmtts_comm_init_rank and by the
 *                         Python host once torch.distributed is initialised)
 * (The switches between a fused kernel and the path it replaces that the bitwise tests flip are not part of the ABI:
 * csrc/internal_hooks.h.) */
int cmtts_set_option(const char* name, int value);
/* Per-handle numerics options: choices that change low-order bits belong to a model, not to the process.
 *   cmtts_model_set_option(m, "ffn2_split", 1 (default) | 0): the FFN linear of the FFT blocks (model/blocks.py:547-551) as
 *       eight K-segment partial GEMMs added in ascending order, or as one launch — two fp32 summation orders.
 *   cmtts_model_set_option(m, "text16", 0 (default) | 1): bf16 / fp16 models only — the four weight contractions of every FFT block (encoder,
 *       decoder: in- / out-projection, FFN conv, FFN linear) and the conv layers of the variance predictors with 16-bit MFMA operands too (LayerNorm, attention scores / softmax / P V, bias,
 *       scale, GELU, residual, mask and accumulation stay fp32).  Off by default because
 *       the text side feeds the integer stages: with it, durations / pitch buckets / lengths may differ from the fp32 model's by one unit.
 *   cmtts_model_set_option(m, "winograd", 1 (default) | 2 | 0): fp32 models, large batches (the persistent denoiser stack, csrc/denoiser_persist.hip):
 *       the gated k = 3 convolution of every residual layer (model/blocks.py:672-679) as a Winograd convolution along the frame axis —
 *       1: F(4,3), 6 products per quad of output frames instead of 12 (round 5); 2: F(2,3), 4 per pair instead of 6 (round 4); transformed
 *       weights formed in double and rounded once, every product and sum in fp32 — or (0) as the direct 3-tap contraction, which is bit for
 *       bit what the per-layer kernels of small batches compute.  The forms differ by fp32 rounding only: F(4,3) <= 1.8e-5 / F(2,3) <= 1e-5 on
 *       one network evaluation, ~8e-6 / ~4e-6 on a T = 4 mel, all equally far from a float64 evaluation at ordinary activation scales
 *       (tests/test_gpu_precision.py; F(4,3) loses a decimal digit on conv inputs of ~6e4); with 1 or 2 an utterance's low-order bits depend on
 *       whether its batch takes the persistent stack.
 *   cmtts_vocoder_set_option(v, "winograd", 1 (default) | 0): fp32 generator, launches of >= 1024 column tiles (large batches): the k = 3 / 7 / 11
 *       ResBlock convs of the C = 256 and C = 128 stages and the k = 7 / 11 ResBlock convs of the C = 64 stage (hifigan/models.py:96-103) as Winograd convolutions — since round 5
 *       over QUADS of outputs one dilation apart, groups of three taps as F(4,3): 6 / 16 / 24 fp32 products per quad instead of 12 / 28 / 44 (csrc/conv_xlq.hip; dilation 1 and 3, dilation 5 at
 *       C = 256 or k = 3); the remaining dilation-5 convs over PAIRS of outputs, groups of three taps as F(2,3), a remainder of two taps as F(2,2): 4 / 10 / 15 products per pair instead of
 *       6 / 14 / 22 (round 4) — or (0) in the direct form, which small batches always take.  fp32 rounding differences only: <= 1.4e-6 on the waveform, both forms
 *       1.0e-6 from a float64 evaluation (tests/test_gpu_precision.py); with 1 an utterance's int16 samples may differ by one LSB between a
 *       large and a small batch.
 *   cmtts_vocoder_set_option(v, "ups16", 1 (default) | 0): in the 16-bit precision modes the ConvTranspose1d upsamplers take
 *       16-bit operands as well, or stay fp32.
 * Same return convention as cmtts_set_option. */
int cmtts_model_set_option(cmtts_model* m, const char* name, int value);
int cmtts_vocoder_set_option(cmtts_vocoder* v, const char* name, int value);
/* Operand precision of the denoiser's residual-block contractions (93 % of its FLOPs): 0 = fp32 (default,
 * the reference's only inference precision), 1 = bf16, 2 = fp16 MFMA operands with fp32 accumulation;
 * activations in HBM, biases, gate and residual arithmetic stay fp32.  BASELINE.json configs[2]/[4].  In modes 1 / 2 the blocks'
 * conditioner projections take 16-bit operands as well (since round 3, every batch shape; mode 3: (hi, lo) pairs).
 * 3 = "fp16x3": every operand carried as hi = fp16(v), lo = fp16(v - hi) and every product as three fp16 MFMAs
 * (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulate): fp32-class accuracy (error vs float64 within 2x of the fp32
 * kernels', tests/test_gpu_precision.py) at 3/16 of the fp32 matrix cost; used by the persistent stack (large
 * batches), other shapes run the exact fp32 kernels. */
int cmtts_set_precision(cmtts_model* m, int mode);
/* Same switch for the HiFi-GAN generator.  Modes 1 / 2 (bf16 / fp16 operands, fp32 accumulate): the 72 ResBlock convs (94 %
 * of the generator's FLOPs) AND, by default, the four ConvTranspose1d upsamplers take 16-bit operands
 * (cmtts_vocoder_set_option(v, "ups16", 0) keeps the upsamplers fp32); inside a ResBlock pair the intermediate xt crosses HBM
 * (or stays in LDS) as convert(leaky_relu(xt)) in 16 bits; the stage tensors (residual stream, MRF sum), conv_pre and conv_post
 * stay fp32.  Mode 3 = fp16x3 as above (fp32-class): ResBlock convs and ("ups16", default) upsamplers with (hi, lo) operand pairs,
 * every HBM tensor fp32. */
int cmtts_vocoder_set_precision(cmtts_vocoder* v, int mode);

/* Tuning knob of the fused residual block: frames per workgroup (0 = automatic: 64 when that still
 * yields >= 512 workgroups, else 32).  Affects speed only, never results. */
int cmtts_set_resblock_tile(int frames);
/* Debug/tuning: when non-NULL, every workgroup of the fused residual block writes 8 int64 s_memtime
 * stamps (phase boundaries) to dev_buf[workgroup*8 ...]; NULL switches it off. */
int cmtts_set_debug_stamps(void* dev_buf);
int cmtts_profile_end(double* total_ms, int* n_launches);

/* ---- FastspeechDecoder.forward (model/modules.py:154-165 = FFTBlocks.forward :80-105 with use_pos_embed=True,
 * learnable pos_embed_alpha): available when the state dict held "decoder.layers.N.op.*", "decoder.layer_norm.*" and
 * "decoder.pos_embed_alpha" (what `self.decoder = FastspeechDecoder(model_config)` registers; CM-TTS defines the class
 * but never instantiates it).  x_ct, out_ct fp32 [B,hidden,T] channel-major; the padding mask is `t >= lens[b]`. */
size_t cmtts_decoder_workspace_bytes(const cmtts_model* m, int B, int T);
int cmtts_decoder_forward(cmtts_model* m, const float* x_ct, const int64_t* lens, int B, int T, float* out_ct,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- the path's one collective (new work: the reference's inference is single-process, synthesize.py:32,43):
 * every rank contributes its padded mel block [Bl,T,M] and mel_len [Bl] and receives all ranks' blocks in rank order.
 * RCCL (librccl.so, the library torch.distributed's "nccl" backend uses on ROCm) is opened at run time with dlopen — the
 * library does not link against it and every other entry point works without it.  A host that is not Python builds its
 * communicator here: rank 0 calls cmtts_comm_unique_id and ships the 128 bytes to the other ranks by any means, every rank
 * calls cmtts_comm_init_rank (one process per GPU, device already selected with hipSetDevice).
 *   cmtts_allgather_mels: mel fp32 [Bl,T,M], mel_len int64 [Bl] (device) -> out_mel fp32 [world*Bl,T,M], out_len int64
 *   [world*Bl] (device); ws = device scratch of cmtts_allgather_workspace_bytes(world, Bl, T, M) bytes (the packed send /
 *   receive buffers: mel_len rides in the same buffer as a trailing column, so it is ONE ncclAllGather); everything is
 *   enqueued on `stream`.  The persistent denoiser needs every CU: issue it after the gather has completed on the stream
 *   (same stream: automatic). */
int cmtts_comm_unique_id(void* id128_host);
int cmtts_comm_init_rank(void** comm, int world, int rank, const void* id128_host);
int cmtts_comm_destroy(void* comm);
size_t cmtts_allgather_workspace_bytes(int world, int Bl, int T, int M);
int cmtts_allgather_mels(void* comm, int world, const float* mel, const int64_t* mel_len, int Bl, int T, int M,
                         float* out_mel, int64_t* out_len, void* ws, size_t ws_bytes, void* stream);
/* The same exchange for end-to-end wav jobs (BASELINE.json configs[4]; SURVEY.md §8e: "cfg5 gathers int16 wav [16, 1024*256]"):
 * what it collates is vocoder_infer's output (utils/model.py:187-205) — pcm int16 [Bl,N] (N = padded frames * hop) and the
 * valid sample count wav_len int64 [Bl] = mel_len * hop (device) -> out_pcm int16 [world*Bl,N], out_len int64 [world*Bl] in
 * rank order.  The count rides in the same buffer (four int16 slots after each row), so it is ONE ncclAllGather of bytes.
 * ws = device scratch of cmtts_allgather_pcm_workspace_bytes(world, Bl, N) bytes. */
size_t cmtts_allgather_pcm_workspace_bytes(int world, int Bl, int64_t N);
int cmtts_allgather_pcm(void* comm, int world, const int16_t* pcm, const int64_t* wav_len, int Bl, int64_t N,
                        int16_t* out_pcm, int64_t* out_len, void* ws, size_t ws_bytes, void* stream);

/* ---- get_mask_from_lengths (utils/tools.py:275-283): mask[b][t] = (t >= lens[b]) as one byte per element
 * (True = padding), lens int64 [B], mask [B,W]. */
int cmtts_length_mask(const int64_t* lens, uint8_t* mask, int B, int W, void* stream);

/* ---- layout helper: in [B,R,C] -> out [B,C,R] */
int cmtts_transpose(const float* in, float* out, int B, int R, int C, void* stream);

/* ---- generic Conv1d on the MFMA kernel, exposed for kernel-level parity tests and roofline
 * measurement: y[B,Cout,T] = act(conv1d(x[B,Cin,T], w, bias, padding, dilation)).
 * `packed_w` comes from cmtts_pack_conv_weight (HOST in [Cout,Cin,K] -> DEVICE packed). */
int cmtts_pack_conv_weight(const float* host_w, int Cout, int Cin, int K, float** dev_packed, int* ld);
void cmtts_free_device(void* p);
int cmtts_conv1d(const float* x, const float* packed_w, int ld, const float* bias, int B, int Cin, int Cout,
                 int T, int K, int dilation, int padding, int act, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CMTTS_HIP_H */
