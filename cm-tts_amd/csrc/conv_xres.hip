// X-resident Conv1D for short sequences on gfx950: the k=9 FFN conv of the FFT blocks (model/blocks.py:539-546,
// Conv1d(256 -> 1024, k=9) + scale + GELU over L <= ~100 phonemes).  The generic kernel (conv_mfma.hip) runs an
// 85-phoneme utterance as two 64-column tiles (a third of the MFMAs multiply padding) and re-stages X per m-tile.
// Here a workgroup (4 waves) stages the whole X tile [K <= 256][96 + halo] of one utterance in LDS once, each wave
// owns one 32-row m-tile over three 32-column n-tiles, and the weights stream L2 -> VGPR in MFMA A-fragment order
// (iteration order [chunk][tap][half][m-tile][lane][4], one dwordx4 per 12 MFMAs) through a register ring — no weight staging,
// no barrier in the K loop.  Same (16-channel chunk, tap, k) accumulation order and the same epilogue arithmetic as
// the generic kernel: BITWISE equal (tests/test_gpu_parity.py::test_xres_conv_bitwise).
//
// Round 2: the same kernel runs the other K = 256 contractions of an FFT block (QKV projection, k = 1), optionally with the
// block's LayerNorm as a PROLOGUE on the staged tile (a.ln_g != nullptr: mean / variance over the K rows of every column,
// two lane halves x 128 rows each, partial sums exchanged with __shfl_xor — no LDS scratch, no extra launch, no normalised
// copy in HBM; layernorm_ct_kernel's summation order => the same bits); the tile is staged with 16-byte loads, 13 in flight per lane, when rows are 16-byte aligned; the epilogue is
// compiled per activation (the run-time switch of the generic epilogue cost 32 k of this kernel's 315 k cycles per wave).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "conv_args.h"
#include "conv_epilogue.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

#ifndef XRES_OCC1
#define XRES_OCC1 2         // 32-column instances: waves per SIMD the register budget allows (41 KB of LDS per workgroup: up to three per CU)
#endif
constexpr int KMAX = 256;
constexpr int HALO = 8;         // (taps - 1) * dil <= 8
constexpr int RING = 4;

// NT = MFMA n-tiles (32 columns each) per wave = per workgroup: 3 (96-column tiles: a full chip, one utterance of <= 96 phonemes per
// tile) or 1 (round 4: 32-column tiles for launches that cannot fill the chip — a single request, a few utterances — where one
// accumulation chain per wave and three times the workgroups finish sooner than three chains per wave; same chains, same bits)
// WQ (round 5): the k = 9 FFN conv of the fused launch as three F(4,3) tap groups over output QUADS (points 0, +-1, +-2, inf: 18 products per quad instead of 36) —
// denoiser_persist.hip's WINO == 2 products on v_mfma_f32_16x16x4_f32: lane (q = l & 15, k = l >> 4) owns quad q of an n-tile of 16 quads in channel 4 ks + k; a wave's
// 32 rows = two 16-row m-tiles x six transforms x NTQ n-tiles (96-column tiles: 24 quads in two n-tiles, the last eight quad lanes idle; 32-column tiles: eight quads in
// one).  wfrag = cmtts_api.hip: to_wino43_xres_fragments ([K/4][M/32][9][64 lanes][4]: element (pt & 1) * 2 + i of vector pt / 2 = transform pt, m-tile i).  Every
// output element sees the same products in the same order whatever the tile width, so the 96- and 32-column instances agree bit for bit; against the direct form the
// difference is fp32 rounding (tests/test_gpu_parity.py).  Everything around the K loop — staging, LayerNorm prologue, GELU, the FFN linear's partial product — is unchanged.
// WQ == 2 (round 6, the default for fp32 models): the same three tap groups as F(2,3) over output PAIRS (points 0, +-1, inf: m0 = (d0 - d2) g0, m1 = (d1 + d2) (g0 + g1 + g2) / 2,
// m2 = (d2 - d1) (g0 - g1 + g2) / 2, m3 = (d1 - d3) g2; y(2p) = m0 + m1 + m2, y(2p + 1) = m1 - m2 - m3).  A 96-column tile is 48 pairs = THREE full n-tiles of 16 pair lanes where it
// is 24 quads = one and a half n-tiles of quad lanes: 4 transforms x 3 n-tiles = the 6 x 2 MFMAs of the F(4,3) form per (k-step, tap group, m-tile) — the same matrix work — with 3
// instead of 16 VALU operations per transformed n-tile, four transformed weight sets instead of six, and F(2,3)'s smaller rounding error (an output depends on its own taps only); a
// 32-column tile is ONE full n-tile of pairs (F(4,3): six transforms on half an n-tile of quads: 1.5 x the MFMAs).  wfrag = cmtts_api.hip: to_wino23_xres_fragments
// ([K/4][M/32][3 groups][2][64 lanes][4]: element (tr & 1) * 2 + i of vector tr / 2 = transform tr, m-tile half i).
template <bool LN, int NT, int WQ = 0>
__global__ __launch_bounds__(256, NT == 1 ? XRES_OCC1 : 1) void conv_xres_kernel(const ConvArgs a, const float* __restrict__ wfrag, long long* dbg) {
    constexpr int BN = 32 * NT;          // columns per workgroup
    constexpr int X_LD = BN + HALO;      // 104 / 40
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [K][X_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // cycle stamps per wave (tools/xres_phases.py; dbg == nullptr in normal operation)
    auto stamp = [&](int slot) {
        if (dbg && lane == 0)
            dbg[((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + w) * 8 + slot] =
                (long long)__builtin_readcyclecounter();
    };
    stamp(0);
    // workgroup -> (m-block, utterance): consecutive workgroup ids go to consecutive XCDs (8 of them, each with its own L2);
    // with one column tile per utterance and a multiple of 8 utterances, all m-blocks of an utterance are put on ONE XCD, so
    // the burst that stages X at the start of the launch reads every utterance's tile from HBM once instead of once per XCD
    // (the weights, whose stream is spread over the whole K loop, are then read by every XCD)
    int by = blockIdx.y, bz = blockIdx.z;
    if (gridDim.x == 1 && (gridDim.z & 7) == 0) {
        const int lin = blockIdx.y + gridDim.y * blockIdx.z;
        const int slot = lin >> 3;
        by = slot % gridDim.y;
        bz = (lin & 7) + 8 * (slot / gridDim.y);
    }
    const int n0 = blockIdx.x * BN;
    const int mt = by * 4 + w;             // this wave's m-tile
    const int z = bz;
    // a ragged batch (ln_lens = each utterance's own padded length): a column tile wholly beyond it computes nothing anyone reads —
    // the normalised input is zero there and every consumer masks those columns by select (reduce_partials, the k = 1 linear)
    if (LN && a.ln_skip_tiles && a.ln_lens && n0 > 0 && (int64_t)n0 >= a.ln_lens[z]) return;
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* Xb = a.X + z * a.x_zs0;
    const int MTn = (a.M + 31) / 32;
    const int total = (a.K / 16) * a.taps * 2;     // k-groups: (16-channel chunk, tap, 8-channel half)

    // weights: A fragments in ITERATION order ([chunk][tap][half][m-tile][lane][4], cmtts_finalize): the stream of k-group
    // it = chunk * nq + tap * 2 + half is one linear walk (clamped m-tile: idle waves load valid memory)
    const int mtc = min(mt, MTn - 1);
    const float* wl = wfrag + ((long)mtc * 64 + lane) * 4;
    auto load_a = [&](f32x4& dst, int it) {
        dst = *reinterpret_cast<const f32x4*>(wl + (long)min(it, total - 1) * MTn * 256);
    };
    f32x4 A[RING];
    if constexpr (WQ == 0) {
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) load_a(A[s], s);
    }

    const int xw = BN + (a.taps - 1) * a.dil;
    float* gs = xs + a.K * X_LD;                   // LayerNorm weight / bias [2][256] behind the tile
    if (LN) { gs[tid] = a.ln_g[tid]; gs[256 + tid] = a.ln_b[tid]; }
    if (((n0 - a.pad) & 3) == 0 && (a.ldx & 3) == 0 && ((uintptr_t)Xb & 15) == 0) {
        // 16-byte loads: tile row = X_LD / 4 = 26 float4; a float4 that starts outside [0, ldx - 4] lies wholly outside [0, Tin)
        constexpr int V = X_LD / 4, U = NT == 3 ? 26 : 10;
        const int nvec = a.K * V;
        const int t00 = n0 - a.pad;
#pragma unroll 1
        for (int base = tid; base < nvec; base += 256 * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = min(base + 256 * u, nvec - 1);
                const int row = idx / V, c4 = idx - row * V;
                const int t4 = min(max(t00 + 4 * c4, 0), a.ldx - 4);
                v[u] = *reinterpret_cast<const f32x4*>(Xb + (long)row * a.ldx + t4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + 256 * u;
                if (idx < nvec) {
                    const int row = idx / V, c4 = idx - row * V;
                    const int c = 4 * c4, t = t00 + c;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (c + e < xw && t + e >= 0 && t + e < a.Tin) ? v[u][e] : 0.f;
                    *reinterpret_cast<f32x4*>(xs + row * X_LD + c) = o;
                }
            }
        }
    } else {   // stage X[k][n0 - pad + c], c in [0, X_LD): zero outside [0, Tin); lane = column (two passes), a quarter of the
        // rows per wave, 32 unconditional clamped loads in flight per lane
        const int rows = a.K / 4;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int c = lane + 64 * cp;
            const int t = n0 - a.pad + c;
            const bool ok = c < xw && t >= 0 && t < a.Tin;
            const int tc = min(max(t, 0), a.Tin - 1);
            if (c < X_LD) {
#pragma unroll 1
                for (int k0 = w * rows; k0 < (w + 1) * rows; k0 += 32) {
                    float v[32];
#pragma unroll
                    for (int q = 0; q < 32; ++q) v[q] = Xb[(long)min(k0 + q, a.K - 1) * a.ldx + tc];
#pragma unroll
                    for (int q = 0; q < 32; ++q)
                        if (k0 + q < (w + 1) * rows) xs[(k0 + q) * X_LD + c] = ok ? v[q] : 0.f;
                }
            }
        }
    }
    stamp(1);
    __syncthreads();
    if (LN) {
        // LayerNorm over the K = 256 rows of every staged column that lies inside [0, Tin) (model/blocks.py:88-107; padding
        // columns stay 0: the conv pads the NORMALISED sequence).  Column = 32 w + (lane & 31); the summation order is
        // layernorm_ct_kernel's (kernels.hip: eight partial sums over rows y + 8 i, added in the order y = 0..7), four partials
        // per lane half, exchanged with __shfl_xor => the same bits as the separate launch, whatever path a batch takes.
        const int c = 32 * w + l31;
        const int t = n0 - a.pad + c;
        const bool live = c < xw && t >= 0 && t < a.Tin;
        const bool on = live && !(a.ln_lens && (int64_t)t >= a.ln_lens[z]);      // masked columns: 0, as layernorm_ct_kernel writes them
        const float* col = xs + min(c, X_LD - 1);
        float p[4] = {0.f, 0.f, 0.f, 0.f}, q[4];
        const float* cr = col + 4 * khalf * X_LD;
#pragma unroll 8
        for (int i2 = 0; i2 < 32; ++i2)          // four independent chains, rows y + 8 i2: loads of a whole unrolled block in flight
#pragma unroll
            for (int y = 0; y < 4; ++y) p[y] += cr[(y + 8 * i2) * X_LD];
#pragma unroll
        for (int y = 0; y < 4; ++y) q[y] = __shfl_xor(p[y], 32);
        float tot = 0.f;
#pragma unroll
        for (int y = 0; y < 4; ++y) tot += khalf ? q[y] : p[y];
#pragma unroll
        for (int y = 0; y < 4; ++y) tot += khalf ? p[y] : q[y];
        const float mean = tot / 256.0f;
#pragma unroll
        for (int y = 0; y < 4; ++y) p[y] = 0.f;
#pragma unroll 8
        for (int i2 = 0; i2 < 32; ++i2)
#pragma unroll
            for (int y = 0; y < 4; ++y) { const float d = cr[(y + 8 * i2) * X_LD] - mean; p[y] = __fmaf_rn(d, d, p[y]); }
#pragma unroll
        for (int y = 0; y < 4; ++y) q[y] = __shfl_xor(p[y], 32);
        float var = 0.f;
#pragma unroll
        for (int y = 0; y < 4; ++y) var += khalf ? q[y] : p[y];
#pragma unroll
        for (int y = 0; y < 4; ++y) var += khalf ? p[y] : q[y];
        var = var / 256.0f;
        const float rstd = 1.0f / sqrtf(var + a.ln_eps);
        if (on) {
            float* wc = xs + c + 4 * khalf * X_LD;
            const float* g = gs + 4 * khalf;
            const float* be = gs + 256 + 4 * khalf;
#pragma unroll 1
            for (int i0 = 0; i0 < 32; i0 += 4) {      // 16 rows per batch: all reads issued before the first write (the compiler
                float xv[16], gv[16], bv[16];         // cannot reorder LDS reads over LDS writes on its own)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = (e & 3) + 8 * (i0 + (e >> 2));
                    xv[e] = wc[k * X_LD]; gv[e] = g[k]; bv[e] = be[k];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = (e & 3) + 8 * (i0 + (e >> 2));
                    wc[k * X_LD] = __fmaf_rn((xv[e] - mean) * rstd, gv[e], bv[e]);
                }
            }
        } else if (live) {
            float* wc = xs + c + 4 * khalf * X_LD;
#pragma unroll 8
            for (int i2 = 0; i2 < 32; ++i2)
#pragma unroll
                for (int y = 0; y < 4; ++y) wc[(y + 8 * i2) * X_LD] = 0.f;
        }
        __syncthreads();
    }
    stamp(2);

    constexpr int NTQ = NT == 3 ? 2 : 1;
    f32x4 Mq[WQ == 1 ? 2 : 1][WQ == 1 ? NTQ : 1][6];
    f32x4 Mp[WQ == 2 ? 2 : 1][WQ == 2 ? NT : 1][4];      // F(2,3): [m-tile half][n-tile of 16 pairs][transform]
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* bl = xs + khalf * X_LD + l31;
    if constexpr (WQ == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int p = 0; p < 4; ++p) Mp[i][nt][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int R = 3, U = 2;      // ring of three entries (tap groups); two k-steps = six entries per unrolled round
        const int NKS = a.K / 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, NKS * MTn * 6 * 1024, 0x00020000);
        const int voff = lane * 16;
        const int mtu = __builtin_amdgcn_readfirstlane(mtc);
        f32x4 Ap[R][2];
        auto load_ap = [&](f32x4 (&dst)[2], int ks0, int n) {      // entry n of the round starting at k-step ks0: k-step ks0 + n / 3 (clamped: the last one again, never used), tap group n % 3
            const int soff = ((min(ks0 + n / 3, NKS - 1) * MTn + mtu) * 3 + (n % 3)) * 2048;
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + j * 1024, soff, 0));
        };
        // raw inputs: pair P = p + 16 nt reads columns 2 P + 3 g .. + 3 of row 4 ks + (lane >> 4) (column c of the tile <-> t = n0 - pad + c, pad = 4): 8-byte aligned reads
        const float* xl = xs + (lane >> 4) * X_LD + 2 * (lane & 15);
        float D[2][NT][4];
        auto load_dp = [&](float (&d)[NT][4], int ks0, int n) {
            const int g = n % 3;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float* r = xl + min(ks0 + n / 3, NKS - 1) * (4 * X_LD) + nt * 32;
                if (g == 1) {          // columns 3 .. 6
                    const f32x2 q = *reinterpret_cast<const f32x2*>(r + 4);
                    d[nt][0] = r[3]; d[nt][1] = q.x; d[nt][2] = q.y; d[nt][3] = r[6];
                } else {               // columns 0 .. 3 / 6 .. 9
                    const f32x2 p = *reinterpret_cast<const f32x2*>(r + 3 * g), q = *reinterpret_cast<const f32x2*>(r + 3 * g + 2);
                    d[nt][0] = p.x; d[nt][1] = p.y; d[nt][2] = q.x; d[nt][3] = q.y;
                }
            }
        };
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_ap(Ap[s], 0, s);
        load_dp(D[0], 0, 0);
#pragma unroll 1
        for (int ks0 = 0; ks0 < NKS; ks0 += U) {
#pragma unroll
            for (int n = 0; n < U * 3; ++n) {
                const int slot = n % R;
                float V[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float (&d)[4] = D[n & 1][nt];
                    const f32x2 P01 = {d[0], d[1]}, P23 = {d[2], d[3]};
                    const f32x2 V03 = P01 - P23;                   // d0 - d2, d1 - d3
                    V[nt][0] = V03.x; V[nt][1] = d[1] + d[2]; V[nt][2] = d[2] - d[1]; V[nt][3] = V03.y;
                }
                __builtin_amdgcn_sched_barrier(0);
                load_ap(Ap[(slot + R - 1) % R], ks0, n + R - 1);
                load_dp(D[(n + 1) & 1], ks0, n + 1);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            Mp[i][nt][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[slot][p >> 1][(p & 1) * 2 + i], V[nt][p], Mp[i][nt][p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if constexpr (WQ == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
                for (int p = 0; p < 6; ++p) Mq[i][nt][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int R = 3, U = 2;      // ring of three entries (tap groups); two k-steps = six entries per unrolled round (ring slots and input buffers compile-time)
        const int NKS = a.K / 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, NKS * MTn * 9 * 1024, 0x00020000);
        const int voff = lane * 16;
        const int mtu = __builtin_amdgcn_readfirstlane(mtc);
        f32x4 Aq[R][3];
        auto load_aq = [&](f32x4 (&dst)[3], int ks0, int n) {      // entry n of the round starting at k-step ks0: k-step ks0 + n / 3, tap group n % 3 (past the end: out of range, zeros)
            const int soff = (((ks0 + n / 3) * MTn + mtu) * 9 + 3 * (n % 3)) * 1024;
#pragma unroll
            for (int j = 0; j < 3; ++j) dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + j * 1024, soff, 0));
        };
        // raw inputs: quad Q = q + 16 nt reads columns 4 Q + 3 g .. + 5 of row 4 ks + (lane >> 4) (column c of the tile <-> t = n0 - pad + c, pad = 4)
        const float* xl = xs + (lane >> 4) * X_LD + 4 * (lane & 15);
        float D[2][NTQ][6];
        auto load_dq = [&](float (&d)[NTQ][6], int ks0, int n) {
            const int g = n % 3;
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) {
                const float* r = xl + (ks0 + n / 3) * (4 * X_LD) + nt * 64;
                if (g == 0) {
                    const f32x4 p = *reinterpret_cast<const f32x4*>(r);
                    const f32x2 q = *reinterpret_cast<const f32x2*>(r + 4);
                    d[nt][0] = p[0]; d[nt][1] = p[1]; d[nt][2] = p[2]; d[nt][3] = p[3]; d[nt][4] = q.x; d[nt][5] = q.y;
                } else if (g == 1) {
                    const f32x4 p = *reinterpret_cast<const f32x4*>(r + 4);
                    d[nt][0] = r[3]; d[nt][1] = p[0]; d[nt][2] = p[1]; d[nt][3] = p[2]; d[nt][4] = p[3]; d[nt][5] = r[8];
                } else {
                    const f32x2 q = *reinterpret_cast<const f32x2*>(r + 6);
                    const f32x4 p = *reinterpret_cast<const f32x4*>(r + 8);
                    d[nt][0] = q.x; d[nt][1] = q.y; d[nt][2] = p[0]; d[nt][3] = p[1]; d[nt][4] = p[2]; d[nt][5] = p[3];
                }
            }
        };
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_aq(Aq[s], 0, s);
        load_dq(D[0], 0, 0);
#pragma unroll 1
        for (int ks0 = 0; ks0 < NKS; ks0 += U) {
#pragma unroll
            for (int n = 0; n < U * 3; ++n) {
                const int slot = n % R;
                float V[NTQ][6];
#pragma unroll
                for (int nt = 0; nt < NTQ; ++nt) {
                    const float (&d)[6] = D[n & 1][nt];
                    const f32x2 P01 = {d[0], d[1]}, P23 = {d[2], d[3]}, P45 = {d[4], d[5]};
                    const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, c2 = {2.f, -2.f};
                    const f32x2 V05 = __builtin_elementwise_fma(c4, P01, __builtin_elementwise_fma(cm5, P23, P45));
                    const float u0 = __builtin_fmaf(-4.f, d[2], d[4]), u1 = __builtin_fmaf(-4.f, d[1], d[3]);
                    const float u2 = d[4] - d[2], u3 = d[3] - d[1];
                    const f32x2 a0 = {u0, u0}, a1 = {u1, -u1}, b0 = {u2, u2}, b1 = {u3, u3};
                    const f32x2 V12 = a0 + a1;
                    const f32x2 V34 = __builtin_elementwise_fma(c2, b1, b0);
                    V[nt][0] = V05.x; V[nt][1] = V12.x; V[nt][2] = V12.y; V[nt][3] = V34.x; V[nt][4] = V34.y; V[nt][5] = V05.y;
                }
                __builtin_amdgcn_sched_barrier(0);
                load_aq(Aq[(slot + R - 1) % R], ks0, n + R - 1);
                load_dq(D[(n + 1) & 1], ks0, n + 1);
                if (ks0 + n / 3 < NKS) {
#pragma unroll
                    for (int p = 0; p < 6; ++p)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int nt = 0; nt < NTQ; ++nt)
                                Mq[i][nt][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aq[slot][p >> 1][(p & 1) * 2 + i], V[nt][p], Mq[i][nt][p], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // scalar walk over (chunk, tap, half): boff = float offset of the group's first row / tap column in the X tile
        int boff = 0, tap = 0, half = 0;
        const int boff_max = (a.K - 8) * X_LD + (a.taps - 1) * a.dil;
        auto advance = [&]() {
            if (half == 0) { half = 1; boff += 8 * X_LD; }
            else {
                half = 0; boff -= 8 * X_LD; ++tap; boff += a.dil;
                if (tap == a.taps) { tap = 0; boff += 16 * X_LD - a.taps * a.dil; }
            }
        };
        float Bv[2][4][NT];
    #pragma unroll
        for (int kk = 0; kk < 4; ++kk)
    #pragma unroll
            for (int j = 0; j < NT; ++j) Bv[0][kk][j] = bl[2 * kk * X_LD + j * 32];
        // loads are threaded between the MFMAs (one k-step's three ds_reads + the A load of a later group per k-step): a wave
        // issues in order, so loads bunched between two groups of MFMAs leave the matrix pipe idle while they issue
    #pragma unroll 1
        for (int it = 0; it < total; it += RING) {
    #pragma unroll
            for (int s = 0; s < RING; ++s) {
                advance();
                const float* bs = bl + min(boff, boff_max);
                load_a(A[(s + RING - 1) % RING], it + s + RING - 1);
    #pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
    #pragma unroll
                    for (int j = 0; j < NT; ++j) Bv[(s + 1) & 1][kk][j] = bs[2 * kk * X_LD + j * 32];
    #pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][kk], Bv[s & 1][kk][j], acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
                    if (kk == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                }
            }
        }
    }
    stamp(3);
    const ConvOut& o = a.out[0];
    if (a.w2frag) {
        // FFN fusion (model/blocks.py:539-551: w_2(gelu(w_1(x) * k^-0.5))): this workgroup's 128 activated rows are K segment `by` of the
        // linear that follows.  They go to LDS (over the X tile) instead of HBM, and the segment's partial product W2[:, 128 by : 128 by + 128] h
        // is formed here — two m-tiles per wave, the generic kernel's (16-row chunk, k) order over the segment => the bits of the
        // K-segment launch it replaces (cmtts_api.hip: FFN2_SEG); reduce_partials_kernel adds the segments, bias, residual and mask as before.
        const int MT2 = a.M2 >> 5;                       // 8
        const float* w2 = a.w2frag + (((long)by * 16) * MT2 + 2 * w) * 256 + lane * 4;
        f32x4 A2[16][2];                                 // the wave's whole weight slice (32 KB), requested before the tile is even written
#pragma unroll
        for (int it = 0; it < 16; ++it)
#pragma unroll
            for (int i = 0; i < 2; ++i) A2[it][i] = *reinterpret_cast<const f32x4*>(w2 + ((long)it * MT2 + i) * 256);
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = o.bias ? o.bias[(unsigned)(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf)] : 0.f;
        __syncthreads();                                 // every wave is done with the X tile
        if constexpr (WQ == 2) {
            // output transform (y(2p) = (m0 + m1) + m2, y(2p + 1) = (m1 - m2) - m3), then the direct form's epilogue per element; the lane's pair of a row as one 8-byte LDS store
            const int p16 = lane & 15, rq = lane >> 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowl = w * 32 + 16 * i + 4 * rq + r;
                    const float bq = o.bias ? o.bias[(unsigned)(mt * 32 + 16 * i + 4 * rq + r)] : 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float m0 = Mp[i][nt][0][r], m1 = Mp[i][nt][1][r], m2 = Mp[i][nt][2][r], m3 = Mp[i][nt][3][r];
                        f32x2 y;
                        y.x = (m0 + m1) + m2;
                        y.y = (m1 - m2) - m3;
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float v = y[c];
                            if (o.bias) v += bq;
                            v *= o.alpha;
                            y[c] = act_apply(v, ACT_GELU_ERF);
                        }
                        *reinterpret_cast<f32x2*>(xs + rowl * X_LD + 2 * (p16 + 16 * nt)) = y;
                    }
                }
        } else if constexpr (WQ == 1) {
            // output transform (y0 = m0 + (m1 + m2) + (m3 + m4), y1 = (m1 - m2) + 2 (m3 - m4), y2 = (m1 + m2) + 4 (m3 + m4), y3 = (m1 - m2) + 8 (m3 - m4) + m5), then the
            // direct form's epilogue per element; the lane's quad of a row as one 16-byte LDS store
            const int q4 = lane & 15, rq = lane >> 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowl = w * 32 + 16 * i + 4 * rq + r;
                    const float bq = o.bias ? o.bias[(unsigned)(mt * 32 + 16 * i + 4 * rq + r)] : 0.f;
#pragma unroll
                    for (int nt = 0; nt < NTQ; ++nt) {
                        const float m0 = Mq[i][nt][0][r], m1 = Mq[i][nt][1][r], m2 = Mq[i][nt][2][r], m3 = Mq[i][nt][3][r], m4 = Mq[i][nt][4][r], m5 = Mq[i][nt][5][r];
                        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                        f32x4 y;
                        y[0] = (m0 + s12) + s34;
                        y[1] = __builtin_fmaf(2.f, d34, d12);
                        y[2] = __builtin_fmaf(4.f, s34, s12);
                        y[3] = __builtin_fmaf(8.f, d34, d12) + m5;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = y[c];
                            if (o.bias) v += bq;
                            v *= o.alpha;
                            y[c] = act_apply(v, ACT_GELU_ERF);
                        }
                        const int col = 4 * (q4 + 16 * nt);
                        if (col < BN) *reinterpret_cast<f32x4*>(xs + rowl * X_LD + col) = y;
                    }
                }
        } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {               // epi_tile_simple<ACT_GELU_ERF>'s arithmetic (no residual, no mask)
                float v = acc[j][r];
                if (o.bias) v += bi[r];
                v *= o.alpha;
                v = act_apply(v, ACT_GELU_ERF);
                xs[(w * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * X_LD + j * 32 + l31] = v;
            }
        }
        __syncthreads();
        f32x16 acc2[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
#pragma unroll
        for (int it = 0; it < 16; ++it)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float b2[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) b2[j] = bl[(it * 8 + 2 * kk) * X_LD + j * 32];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[it][i][kk], b2[j], acc2[i][j], 0, 0, 0);
            }
        float* pb = a.part + z * a.part_zs0 + by * a.part_zs1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + j * 32 + l31;
                if (n < a.N && n < o.Tout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        pb[(unsigned)((2 * w + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * (unsigned)a.part_ld + (unsigned)n] = acc2[i][j][r] * 1.0f;
                }
            }
        stamp(4);
        return;
    }
    if (mt >= MTn) return;
    if (epi_simple(o) && o.act == ACT_GELU_ERF) {
#pragma unroll
        for (int j = 0; j < NT; ++j) epi_tile_simple<ACT_GELU_ERF>(o, acc[j], mt * 32, 4 * khalf, n0 + j * 32 + l31, a.M, a.N, z);
    } else if (epi_simple(o) && o.act == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < NT; ++j) epi_tile_simple<ACT_RELU>(o, acc[j], mt * 32, 4 * khalf, n0 + j * 32 + l31, a.M, a.N, z);
    } else if (epi_simple(o) && o.act == ACT_NONE) {
#pragma unroll
        for (int j = 0; j < NT; ++j) epi_tile_simple<ACT_NONE>(o, acc[j], mt * 32, 4 * khalf, n0 + j * 32 + l31, a.M, a.N, z);
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) epi_tile(o, acc[j], mt * 32, 4 * khalf, n0 + j * 32 + l31, a.M, a.N, z, 0);
    }
    stamp(4);
}

long long* g_xres_dbg = nullptr;
int g_xres_nt = 0;          // internal switch "xres_nt": 0 = the launcher's rule, 1 / 3 = every launch that does not ask for one itself (measurements)

}  // namespace

extern "C" void cmtts_xres_set_debug(long long* dbg) { g_xres_dbg = dbg; }
extern "C" int cmtts_xres_set_nt(int nt) { const int p = g_xres_nt; if (nt == 0 || nt == 1 || nt == 3) g_xres_nt = nt; return p; }

// Conv1d with fp32 weights as MFMA A fragments in iteration order [K/16][taps][2][ceil(M/32)][64][4]; zdiv == 1, split == INT_MAX,
// dil > 0, K % 32 == 0, K <= 256, (taps-1)*dil <= 8, no pre-activation; a.ln_g / ln_b / ln_eps = LayerNorm prologue over the K rows.  Meant for short sequences (N of a few 96-column tiles)
// with many output rows.  Returns 0, -2 (unsupported: use cmtts_launch_conv) or -3.
extern "C" int cmtts_launch_conv_xres(const ConvArgs* ap, const float* wfrag, int nbatch, void* stream_) {
    const ConvArgs& a = *ap;
    if (a.M <= 0 || a.N <= 0 || nbatch <= 0) return 0;
    if (!wfrag || a.zdiv != 1 || a.split != INT_MAX || a.dil <= 0 || a.K % 32 != 0 || a.K > KMAX || (a.taps - 1) * a.dil > HALO ||
        a.pre_div != 1.0f || a.pre_slope != 1.0f)
        return -2;
    if (a.ln_g && (!a.ln_b || a.K != 256)) return -2;
    if (a.w2frag) {      // FFN fusion: 128-row m-blocks = K segments of a 256-row linear, plain GELU epilogue, an LDS tile of at least 128 rows
        const ConvOut& o = a.out[0];
        if (!a.part || a.M2 != 256 || a.M % 128 != 0 || a.K < 128 || o.ostride != 1 || o.ooff_base || o.ooff_mul || o.row_off || o.div != 1.0f || o.accum || o.bvec || o.act != ACT_GELU_ERF || o.res || o.lens) return -2;
    }
    // column tiles: 96 (three n-tiles per wave) when that already gives every CU a workgroup, else 32 (one n-tile per wave: three times the
    // workgroups, a third of the MFMAs per accumulation chain's wave); a.xres_nt forces one (tests, tools)
    const int mblocks = ((a.M + 31) / 32 + 3) / 4;
    const long wg96 = (long)((a.N + 95) / 96) * mblocks * nbatch;
    const int nt = a.xres_nt == 1 || a.xres_nt == 3 ? a.xres_nt : (g_xres_nt == 1 || g_xres_nt == 3 ? g_xres_nt : (wg96 >= 128 ? 3 : 1));
    const int bn = 32 * nt, x_ld = bn + HALO;
    static bool attr_set = false;
    const size_t lds = (size_t)a.K * x_ld * sizeof(float) + (a.ln_g ? 2 * 256 * sizeof(float) : 0);
    if (!attr_set) {
        const int mx = (int)((size_t)KMAX * (96 + HALO) * sizeof(float) + 2 * 256 * sizeof(float));
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.N + bn - 1) / bn, mblocks, nbatch);
    long long* dbg = g_xres_dbg;
    if (dbg) {    // CMTTS_XRES_DBG_M=<rows>: stamps of the launches with that many output rows only (tools/xres_phases.py)
        static const char* want = getenv("CMTTS_XRES_DBG_M");
        if (want && atoi(want) != a.M) dbg = nullptr;
    }
    hipStream_t st = (hipStream_t)stream_;
    if (dbg) {      // CMTTS_XRES_DBG_TWICE=1 (tools/xres_phases.py): the stamped launch runs twice back to back — the second one with the kernel's code warm in the caches (idempotent launches only: no in-place residual)
        static const char* twice = getenv("CMTTS_XRES_DBG_TWICE");
        if (twice && atoi(twice) == 1 && a.out[0].res != a.out[0].Y && !a.w2frag) {
            if (nt == 3) { if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 3>), grid, dim3(256), lds, st, a, wfrag, nullptr); else hipLaunchKernelGGL((conv_xres_kernel<false, 3>), grid, dim3(256), lds, st, a, wfrag, nullptr); }
            else { if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 1>), grid, dim3(256), lds, st, a, wfrag, nullptr); else hipLaunchKernelGGL((conv_xres_kernel<false, 1>), grid, dim3(256), lds, st, a, wfrag, nullptr); }
        }
    }
    if (nt == 3) {
        if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 3>), grid, dim3(256), lds, st, a, wfrag, dbg);
        else hipLaunchKernelGGL((conv_xres_kernel<false, 3>), grid, dim3(256), lds, st, a, wfrag, dbg);
    } else {
        if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 1>), grid, dim3(256), lds, st, a, wfrag, dbg);
        else hipLaunchKernelGGL((conv_xres_kernel<false, 1>), grid, dim3(256), lds, st, a, wfrag, dbg);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// The fused FFN launch (a.w2frag set: k = 9 conv + GELU + the FFN linear's K-segment partial product) with the conv as three F(4,3) tap groups (template parameter WQ);
// wfrag = to_wino43_xres_fragments of the same weights.  Returns 0, -2 (not that launch: use cmtts_launch_conv_xres) or -3.
extern "C" int cmtts_launch_conv_xresq(const ConvArgs* ap, const float* wfrag, int nbatch, void* stream_, int form) {      // form: 1 = F(4,3) quads (to_wino43_xres_fragments), 2 = F(2,3) pairs (to_wino23_xres_fragments)
    const ConvArgs& a = *ap;
    if (a.M <= 0 || a.N <= 0 || nbatch <= 0) return 0;
    if (!wfrag || a.zdiv != 1 || a.split != INT_MAX || a.dil <= 0 || a.K % 32 != 0 || a.K > KMAX || (a.taps - 1) * a.dil > HALO ||
        a.pre_div != 1.0f || a.pre_slope != 1.0f)
        return -2;
    if (a.ln_g && (!a.ln_b || a.K != 256)) return -2;
    if (!a.w2frag || a.taps != 9 || a.dil != 1 || a.pad != 4 || a.K != 256) return -2;      // the fused FFN launch of an FFT block only
    if (a.w2frag) {      // FFN fusion: 128-row m-blocks = K segments of a 256-row linear, plain GELU epilogue, an LDS tile of at least 128 rows
        const ConvOut& o = a.out[0];
        if (!a.part || a.M2 != 256 || a.M % 128 != 0 || a.K < 128 || o.ostride != 1 || o.ooff_base || o.ooff_mul || o.row_off || o.div != 1.0f || o.accum || o.bvec || o.act != ACT_GELU_ERF || o.res || o.lens) return -2;
    }
    // column tiles: 96 (three n-tiles per wave) when that already gives every CU a workgroup, else 32 (one n-tile per wave: three times the
    // workgroups, a third of the MFMAs per accumulation chain's wave); a.xres_nt forces one (tests, tools)
    const int mblocks = ((a.M + 31) / 32 + 3) / 4;
    const long wg96 = (long)((a.N + 95) / 96) * mblocks * nbatch;
    const int nt = a.xres_nt == 1 || a.xres_nt == 3 ? a.xres_nt : (g_xres_nt == 1 || g_xres_nt == 3 ? g_xres_nt : (wg96 >= 128 ? 3 : 1));
    const int bn = 32 * nt, x_ld = bn + HALO;
    static bool attr_set = false;
    const size_t lds = (size_t)a.K * x_ld * sizeof(float) + (a.ln_g ? 2 * 256 * sizeof(float) : 0);
    if (!attr_set) {
        const int mx = (int)((size_t)KMAX * (96 + HALO) * sizeof(float) + 2 * 256 * sizeof(float));
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<false, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<true, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<false, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<true, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<false, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<true, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<false, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel<true, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.N + bn - 1) / bn, mblocks, nbatch);
    long long* dbg = g_xres_dbg;
    if (dbg) {    // CMTTS_XRES_DBG_M=<rows>: stamps of the launches with that many output rows only (tools/xres_phases.py)
        static const char* want = getenv("CMTTS_XRES_DBG_M");
        if (want && atoi(want) != a.M) dbg = nullptr;
    }
    hipStream_t st = (hipStream_t)stream_;
    if (form == 2) {
        if (nt == 3) {
            if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 3, 2>), grid, dim3(256), lds, st, a, wfrag, dbg);
            else hipLaunchKernelGGL((conv_xres_kernel<false, 3, 2>), grid, dim3(256), lds, st, a, wfrag, dbg);
        } else {
            if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 1, 2>), grid, dim3(256), lds, st, a, wfrag, dbg);
            else hipLaunchKernelGGL((conv_xres_kernel<false, 1, 2>), grid, dim3(256), lds, st, a, wfrag, dbg);
        }
    } else if (nt == 3) {
        if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 3, 1>), grid, dim3(256), lds, st, a, wfrag, dbg);
        else hipLaunchKernelGGL((conv_xres_kernel<false, 3, 1>), grid, dim3(256), lds, st, a, wfrag, dbg);
    } else {
        if (a.ln_g) hipLaunchKernelGGL((conv_xres_kernel<true, 1, 1>), grid, dim3(256), lds, st, a, wfrag, dbg);
        else hipLaunchKernelGGL((conv_xres_kernel<false, 1, 1>), grid, dim3(256), lds, st, a, wfrag, dbg);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
