// Arguments of the persistent denoiser-stack kernel (denoiser_persist.hip).
#pragma once
#include <stddef.h>

#define PERSIST_MAX_LAYERS 32
#define PERSIST_MAX_GROUPS 8      // utterance groups (frame buckets) one ragged launch may mix
#define PERSIST_MAX_WG 256        // workgroups of a ragged launch (= the CU count of an MI355X; larger shards run in rounds)

// One utterance group of a ragged launch (denoiser_persist.hip, RAGGED instance): the buffers cmtts_sample carves for a (B, T) batch
struct PersistGroup {
    const float* x0;      // [B][256][T]
    const float* cp;      // [B][NL*256][T]
    const float* dp;      // [B][vec_stride]
    const float* d;       // [B][vec_stride]
    float* skip;          // [B][256][T] (tail == 0 only)
    unsigned long long* halo;   // [2][B][tiles][2][256] granules of THIS group
    const float* xold;    // tail: [B][T][n_mels] or null
    const float* noise;   // tail: [B][T][n_mels] or null
    float* out;           // tail: [B][T][n_mels]
    long cp_bstride;
    int B, T, tiles, pad_;
    // FACT instances (round 4): the conditioner projections gathered from their factors instead of read from cp (PersistArgs.p2 is shared)
    const float* p1;      // [B][NL*256][ldp]
    const long long* mel2ph;    // [B][T] int64
    const long long* pidx;      // [B][T] int64
    int ldp, Lph;
    const float* p1t;     // [B][NL][ldp][256]: p1 with the channels contiguous (round 5: the publish phase's 16-byte gathers; PersistArgs.p2t)
    float* xst;           // WINO instances: [B][tiles][16384] kernel-private state (the residual stream x of every tile between layers)
};

struct PersistArgs {
    const float* x0;      // [B][256][T] input of layer 0 (input projection output)
    const float* cp;      // [B][NL*256][T] conditioner projections of all layers (batch stride cp_bstride)
    const float* dp;      // [B][vec_stride]: per layer diffusion (+ speaker) projection, layer l at + l*256
    const float* d;       // [B][vec_stride]: diffusion projection alone (residual = x + d)
    float* skip;          // [B][256][T] out: sum of the layers' skip halves
    const float* W3f[PERSIST_MAX_LAYERS];   // per layer: conv_layer / output_projection in MFMA A-fragment order
    const float* b3[PERSIST_MAX_LAYERS];    // (same packing as ResArgs, resblock_args.h)
    const float* Wof[PERSIST_MAX_LAYERS];
    const float* bo[PERSIST_MAX_LAYERS];
    unsigned long long* halo;   // [2][B][tiles][2][256] {tag, value} granules, zeroed by the launcher
    unsigned* tmo;              // set to 1 when a bounded neighbour wait expires
    long cp_bstride;
    long vec_stride;
    int B, T, NL;
    int tiles;            // filled by the launcher
    // optional in-kernel tail (fp32 kernel): skip head of Denoiser.forward (model/modules.py:634-637) + the sampler's
    // post-scaling (karras_diffusion.py:406,852); when tail != 0 the skip sum is not written to `skip`
    int tail;
    const float* Wsf;     // skip_projection, fragment order [32][8][64][4]
    const float* bs;      // [256]
    const float* Wpf;     // output_projection, fragment order [32][3][64][4] (rows >= n_mels zero)
    const float* bp;      // [n_mels]
    float skip_div;       // sqrt(NL)
    int n_mels;
    const float* xold;    // [B][T][n_mels] or null
    const float* noise;   // [B][T][n_mels] or null
    float c_out, c_skip, nstd;
    float* out;           // [B][T][n_mels]
    // FACT instances (round 4, fp32 kernel): cp[r][t] = (mel2ph[t] > 0 ? p1[r][mel2ph[t] - 1] : 0) + p2[r][pidx[t]] is formed where cp would be
    // read — the cp tensor is neither written (cond_expand_kernel) nor read (335 MB per launch at the bench shape); the same bits as
    // expanding first.  fact = 0: read cp.
    int fact;
    const float* p1;      // [B][NL*256][ldp]: conditioner weights applied to the phoneme-level conditioning, no bias
    const float* p2;      // [NL*256][ld2]: the same weights applied to the pitch-embedding table, + bias
    const long long* mel2ph;    // [B][T] int64: 1-based phoneme of a frame, 0 = padding
    const long long* pidx;      // [B][T] int64: pitch bucket of a frame
    int ldp, Lph, ld2;
    // Round 5: the same factors with the CHANNELS contiguous — p1t [B][NL][ldp][256], p2t [NL][ld2][256] — for the 8-wave FACT instances'
    // publish phase: an element's 16 rows are four runs of four consecutive channels, i.e. four 16-byte loads per factor and n-tile where
    // the row-major tables needed 16 scattered dwords each (the phase was bound by the cache lines its gathers touch).  Layer-0 staging and the
    // halo entries (one element per lane) keep reading p1 / p2.
    const float* p1t;
    const float* p2t;
    int wino;             // 3 (round 5, the default): the 8-wave F(4,3) instances (denoiser_persist.hip, WINO == 2) — W3f = six transformed weight sets as
                          // v_mfma_f32_16x16x4_f32 fragments (cmtts_api.hip: to_wino43_fragments), state as for 1.
                          // (2 was round 5's one-wave-per-SIMD stack: tools/attic/denoiser_persist4.hip, not built.)  1: fp32 kernel, round 4: W3f holds the Winograd F(2,3) transformed conv weights (cmtts_api.hip: to_wino_fragments) and `skip` is the
                          // kernel's between-layers storage of the skip sum (denoiser_persist.hip, WINO instances); NOT bitwise the direct form
    float* xst;           // WINO: [B][tiles][16384] kernel-private state — the residual stream x of every 64-frame tile between layers
                          // (cmtts_persist_state_floats(B, T) floats)
    int halo_zeroed;      // the caller has already cleared `halo` on this stream (inproj.hip): the launcher skips its memset
    long long* dbg;       // optional [grid][16 waves][8] cycle stamps of layer NL/2 (phase timing, tools/persist_timing.py)
    // ---- ragged launches (round 3; fp32 kernel): a 1-D grid of n_wg workgroups, workgroup i works on tile (desc >> 13 & 127) of
    // utterance (desc >> 3 & 1023) of group (desc & 7); (desc >> 20 & 255) = ACTIVE tiles of that utterance: frames at and beyond
    // active * 64 are treated like frames beyond T (never computed, never read).  x0 / cp / dp / d / skip / halo / xold / noise / out /
    // cp_bstride / B / T / tiles above are ignored; everything else (weights, NL, tail constants, tmo) is shared by all groups.
    int n_groups;         // 0 = the uniform (tile, b) grid
    int n_wg;
    PersistGroup grp[PERSIST_MAX_GROUPS];
    unsigned desc[PERSIST_MAX_WG];
};

#ifdef __cplusplus
extern "C" {
#endif
size_t cmtts_persist_halo_bytes(int B, int T);
size_t cmtts_persist_state_floats(int B, int T);   // PersistArgs.xst / PersistGroup.xst of a (B, T) batch
// max_blocks = workgroups that are certainly co-resident (the CU count).  0 = launched, -2 = shape not
// supported or (unless force) too small to pay off: the caller uses the per-layer kernels, -3 = HIP error.
int cmtts_launch_denoiser_persist(const PersistArgs* a, int max_blocks, int force, void* stream);
// 16-bit operand variant (denoiser_persist_lp.hip): mode 1 = bf16, 2 = fp16; W3f / Wof = 16-bit fragment-order weights.
int cmtts_launch_denoiser_persist_lp(const PersistArgs* a, int mode, int max_blocks, int force, void* stream);
// Ragged form: a->n_groups, a->grp[], a->n_wg, a->desc[] filled by the caller (cmtts_api.hip: sample_ragged); the halo granules of every
// group must already be cleared on `stream`.  0 = launched, -2 = not supported, -3 = HIP error.
int cmtts_launch_denoiser_persist_ragged(const PersistArgs* a, void* stream);
int cmtts_persist_plan(int B, int T, int NL, int max_blocks, int force);   // resident workgroups of the largest launch (0 = path not taken)
int cmtts_persist_chunks(int B, int T, int max_blocks);   // launches one call makes (0 = not supported)
void cmtts_persist_set_debug(long long* dbg);
long long* cmtts_persist_get_debug(void);
// 1: launch through hipLaunchCooperativeKernel (the runtime refuses a grid that cannot be co-resident and dispatches it
// with the cooperative-queue guarantee); 0: plain launch, grid <= CU count by construction.  Returns the previous value.
// -1 (default) = automatic: once a process group / communicator exists in the process (cmtts_persist_note_process_group) the first
// launch of every (variant, grid) is cooperative — the runtime validates co-residency — and later ones are plain (denoiser_persist.hip).
int cmtts_persist_set_cooperative(int on);
void cmtts_persist_validated(int variant, int gx, int gy);    // the runtime accepted a cooperative launch of this grid
int cmtts_persist_cooperative(int variant, int gx, int gy);   // should THIS launch be cooperative? (variant = kernel instance, denoiser_persist.hip)
int cmtts_persist_note_process_group(int on);   // cmtts_comm_init_rank and the Python host (torch.distributed initialised) call this
#ifdef __cplusplus
}
#endif
