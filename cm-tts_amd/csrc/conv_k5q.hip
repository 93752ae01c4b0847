// Frame-level k = 5 convs of the pitch (cwt) predictor, fp32, as Winograd F(4,3) tap groups over output QUADS (round 6) — model/modules.py:470-499
// (`Conv1d(C_in, 256, 5, padding=2) -> ReLU -> LayerNorm`, two blocks: 128 -> 256 and 256 -> 256) over the B x T length-regulated frames:
// 16.1 GFLOP per 32 x 512-frame batch in the direct form, the largest contraction of the text + frame side after the FFN convs.
//
// The products are conv_xlq.hip's (HiFi-GAN, round 5): five taps = two groups of three with a zero sixth tap; a group at tap offset o reads
// d0..d5 = X[4q + o .. 4q + o + 5] of the quad's input row (column j of the staged tile = frame t0 - 2 + j) and forms
//   V0 = 4 d0 - 5 d2 + d4, V1 = (d4 - 4 d2) + (d3 - 4 d1), V2 = (d4 - 4 d2) - (d3 - 4 d1), V3 = (d4 - d2) + 2 (d3 - d1), V4 = (d4 - d2) - 2 (d3 - d1),
//   V5 = 4 d1 - 5 d3 + d5;   M_p += U_p V_p  (U = cmtts_api.hip: to_wino43_iter_fragments, formed in double, rounded once);
//   y0 = M0 + (M1 + M2) + (M3 + M4), y1 = (M1 - M2) + 2 (M3 - M4), y2 = (M1 + M2) + 4 (M3 + M4), y3 = (M1 - M2) + 8 (M3 - M4) + M5
// — 12 products per quad instead of 20.  One n-tile of v_mfma_f32_16x16x4_f32 = one transform of the tile's 16 quads = 64 output frames; lane
// (q = l & 15, k = l >> 4) owns quad q in input channel 4 ks + k, so all six transforms of a quad are in-lane.  A wave owns 64 output rows (four
// 16-row m-tiles x six transforms = 24 accumulators of 4 registers); weights stream L2 -> VGPR in iteration order through a buffer descriptor
// and a four-entry register ring, no barrier in the K loop.
//
// What this kernel adds to conv_xlq's scheme:
//   * C_in != C_out (the first block reads the 128-channel input projection);
//   * the PREVIOUS block's LayerNorm (over the 256 channels of every frame, eps 1e-12) as a prologue on the staged tile — layernorm_ct_kernel's
//     summation order, conv_xres.hip's code: no LayerNorm launch, no normalised copy in HBM;
//   * a row split: gridDim.z workgroups of 4 / gridDim.z waves share a frame tile (a single request has three tiles: one workgroup per tile
//     would walk all 256 rows alone).  An output element's products and their order do not depend on the split, so the form is taken at
//     EVERY batch size and an utterance's pitch contour does not depend on the batch it is in (as for the FFN convs, conv_xres.hip WQ).
// NOT bitwise the direct form (conv_xl_kernel / conv_xres_kernel): fp32 Winograd rounding, ~2e-6 of the activations per conv
// (tests/test_gpu_parity.py::test_conv_k5q_kernel_vs_oracle; restated in oracle/winograd_ref.py: conv1d_f43_taps, k = 5).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "resblock_pair.h"
#include "xcd_map.h"
#ifndef XCD_MAP
#define XCD_MAP 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CO = 256;          // output rows
constexpr int KT = 5, PAD = 2;
constexpr int BN = 64;           // output frames per tile = 16 quads
constexpr int XIN = BN + KT;     // staged columns: frames t0 - 2 .. t0 + 66 (the last one under the zero sixth tap)
constexpr int XW = 72;           // row pitch (16-byte aligned quads)
constexpr int NPT = 12;          // transformed weight sets: two groups x six points
constexpr int NV = XW / 4;       // 16-byte items per staged row
constexpr int R = 4;             // weight ring: three entries in flight + the one in use
constexpr int U = 2;             // k-steps per unrolled round (four entries: ring slots and input buffers are compile-time)

template <int CIN, bool LN>
__global__ __launch_bounds__(256, 2) void conv_k5q_kernel(const ConvXlArgs a) {
    constexpr int NKS = CIN / 4;
    static_assert(NKS % U == 0, "k-steps per round");
    extern __shared__ __attribute__((aligned(16))) float Xs[];      // [CIN][XW] (+ LayerNorm weight / bias [2][256])
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wl = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nthr >> 6;
    const int w = blockIdx.z * nwv + wl;          // this wave's block of 64 output rows
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * BN;
    const int T = a.T;
    const float* xb = a.x + (long)b * (a.cin ? a.xbstride : a.bstride);
    float* gs = Xs + CIN * XW;
    if (LN) {
        for (int i = tid; i < 256; i += nthr) { gs[i] = a.ln_g[i]; gs[256 + i] = a.ln_b[i]; }
    }
    {   // stage the tile: items (row, quad of columns) on consecutive threads, up to NV 16-byte loads in flight per lane before the first LDS
        // write (4-byte aligned addresses: global_load_dwordx4 takes them); zeros outside [0, T)
        const int tbase = t0 - PAD;
        constexpr int total = CIN * NV;
#pragma unroll 1
        for (int base = tid; base < total; base += nthr * NV) {
            f32x4 v[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int idx = min(base + u * nthr, total - 1), r = idx / NV, q = idx - r * NV;
                const int t = tbase + 4 * q;
                const float* src = xb + (long)r * a.ld;
                if (t >= 0 && t + 3 < T) v[u] = *reinterpret_cast<const f32x4*>(src + t);
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[u][c] = src[min(max(t + c, 0), T - 1)];
                }
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int idx = base + u * nthr;
                if (idx < total) {
                    const int r = idx / NV, q = idx - r * NV;
                    const int t = tbase + 4 * q;
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = (4 * q + c < XIN && t + c >= 0 && t + c < T) ? v[u][c] : 0.f;
                    *reinterpret_cast<f32x4*>(Xs + r * XW + 4 * q) = o;
                }
            }
        }
    }
    __syncthreads();
    if constexpr (LN) {
        // LayerNorm over the 256 rows of every staged column inside [0, T) (padding columns stay 0: the conv pads the NORMALISED sequence);
        // conv_xres.hip's prologue: column = 32 x (wave slot) + (lane & 31), layernorm_ct_kernel's summation order (eight partial sums over
        // rows y + 8 i, four per lane half, exchanged with __shfl_xor)
        static_assert(!LN || CIN == 256, "LayerNorm prologue: 256 channels");
        const int l31 = lane & 31, khalf = lane >> 5;
        for (int cb = 32 * wl; cb < XW; cb += 32 * nwv) {
            const int c = cb + l31;
            const int t = t0 - PAD + c;
            const bool live = c < XIN && t >= 0 && t < T;
            const float* cr = Xs + min(c, XW - 1) + 4 * khalf * XW;
            float p[4] = {0.f, 0.f, 0.f, 0.f}, q[4];
#pragma unroll 8
            for (int i2 = 0; i2 < 32; ++i2)
#pragma unroll
                for (int y = 0; y < 4; ++y) p[y] += cr[(y + 8 * i2) * XW];
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
                for (int y = 0; y < 4; ++y) { const float d = cr[(y + 8 * i2) * XW] - mean; p[y] = __fmaf_rn(d, d, p[y]); }
#pragma unroll
            for (int y = 0; y < 4; ++y) q[y] = __shfl_xor(p[y], 32);
            float var = 0.f;
#pragma unroll
            for (int y = 0; y < 4; ++y) var += khalf ? q[y] : p[y];
#pragma unroll
            for (int y = 0; y < 4; ++y) var += khalf ? p[y] : q[y];
            var = var / 256.0f;
            const float rstd = 1.0f / sqrtf(var + a.ln_eps);
            if (live) {
                float* wc = Xs + c + 4 * khalf * XW;
                const float* g = gs + 4 * khalf;
                const float* be = gs + 256 + 4 * khalf;
#pragma unroll 1
                for (int i0 = 0; i0 < 32; i0 += 4) {      // 16 rows per batch: all reads issued before the first write
                    float xv[16], gv[16], bv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int k = (e & 3) + 8 * (i0 + (e >> 2));
                        xv[e] = wc[k * XW]; gv[e] = g[k]; bv[e] = be[k];
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int k = (e & 3) + 8 * (i0 + (e >> 2));
                        wc[k * XW] = __fmaf_rn((xv[e] - mean) * rstd, gv[e], bv[e]);
                    }
                }
            }
        }
        __syncthreads();
    }

    f32x4 M[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 6; ++p) M[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        constexpr int NWVT = CO / 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wf), 0, NKS * NWVT * NPT * 1024, 0x00020000);
        const int voff = lane * 16;
        f32x4 A[R][6];
        // entry n of the round that starts at k-step ks0: (k-step ks0 + n / 2, tap group n % 2); past the last k-step the last one is requested again (never used)
        auto load_a = [&](f32x4 (&dst)[6], int ks0, int n) {
            const int soff = ((min(ks0 + n / 2, NKS - 1) * NWVT + w) * NPT + 6 * (n % 2)) * 1024;
#pragma unroll
            for (int p = 0; p < 6; ++p) dst[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + p * 1024, soff, 0));
        };
        // raw inputs of an entry: the lane's quad at column 4 q of row 4 ks + (lane >> 4); group 0 = columns 0..5, group 1 = columns 3..8.  Past the
        // last k-step the row is clamped (the values are never used)
        const float* xl = Xs + (lane >> 4) * XW + 4 * (lane & 15);
        float D[2][6];
        auto load_d = [&](float (&d)[6], int ks0, int n) {
            const float* r = xl + min(ks0 + n / 2, NKS - 1) * (4 * XW);
            if (n % 2 == 0) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(r);
                const f32x2 q = *reinterpret_cast<const f32x2*>(r + 4);
                d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3]; d[4] = q.x; d[5] = q.y;
            } else {
                const f32x4 p = *reinterpret_cast<const f32x4*>(r + 4);
                d[0] = r[3]; d[1] = p[0]; d[2] = p[1]; d[3] = p[2]; d[4] = p[3]; d[5] = r[8];
            }
        };
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_a(A[s], 0, s);
        load_d(D[0], 0, 0);
#pragma unroll 1
        for (int ks0 = 0; ks0 < NKS; ks0 += U) {
#pragma unroll
            for (int n = 0; n < U * 2; ++n) {
                const int slot = n % R;
                float V[6];
                {
                    const float (&d)[6] = D[n & 1];
                    const f32x2 P01 = {d[0], d[1]}, P23 = {d[2], d[3]}, P45 = {d[4], d[5]};
                    const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, c2 = {2.f, -2.f};
                    const f32x2 V05 = __builtin_elementwise_fma(c4, P01, __builtin_elementwise_fma(cm5, P23, P45));
                    const float u0 = __builtin_fmaf(-4.f, d[2], d[4]), u1 = __builtin_fmaf(-4.f, d[1], d[3]);
                    const float u2 = d[4] - d[2], u3 = d[3] - d[1];
                    const f32x2 a0 = {u0, u0}, a1 = {u1, -u1}, b0 = {u2, u2}, b1 = {u3, u3};
                    const f32x2 V12 = a0 + a1;
                    const f32x2 V34 = __builtin_elementwise_fma(c2, b1, b0);
                    V[0] = V05.x; V[1] = V12.x; V[2] = V12.y; V[3] = V34.x; V[4] = V34.y; V[5] = V05.y;
                }
                __builtin_amdgcn_sched_barrier(0);
                load_a(A[(slot + R - 1) % R], ks0, n + R - 1);
                load_d(D[(n + 1) & 1], ks0, n + 1);
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) M[i][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[slot][p][i], V[p], M[i][p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- output transform + epilogue: (acc + bias) [ReLU], the lane's quad of a row as one 16-byte store where rows are aligned
    float* yb = a.y + (long)b * a.bstride;
    const int q4 = lane & 15, rq = lane >> 4;
    const int tq = t0 + 4 * q4;
    const bool vec = ((a.ld & 3) == 0) && ((reinterpret_cast<size_t>(yb) & 15) == 0) && tq + 3 < T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float bi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bi[r] = a.bias[w * 64 + 16 * i + 4 * rq + r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = w * 64 + 16 * i + 4 * rq + r;
            const float m0 = M[i][0][r], m1 = M[i][1][r], m2 = M[i][2][r], m3 = M[i][3][r], m4 = M[i][4][r], m5 = M[i][5][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            f32x4 y;
            y[0] = ((m0 + s12) + s34) + bi[r];
            y[1] = __builtin_fmaf(2.f, d34, d12) + bi[r];
            y[2] = __builtin_fmaf(4.f, s34, s12) + bi[r];
            y[3] = (__builtin_fmaf(8.f, d34, d12) + m5) + bi[r];
            if (a.relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) y[c] = y[c] > 0.f ? y[c] : 0.f;
            }
            const long o = (long)row * a.ld + tq;
            if (vec) *reinterpret_cast<f32x4*>(yb + o) = y;
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (tq + c < T) yb[o + c] = y[c];
            }
        }
    }
}

template <int CIN, bool LN>
int launch_k5q(const ConvXlArgs& a, hipStream_t stream) {
    const size_t lds = ((size_t)CIN * XW + (LN ? 512 : 0)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k5q_kernel<CIN, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
        attr_set = true;
    }
    // row split: 1 (four waves walk all 256 rows of a tile) once every CU has a tile, else 2 / 4 workgroups of 2 / 1 waves per tile
    const long tiles = (long)((a.T + BN - 1) / BN) * a.B;
    const int split = a.row_split == 1 || a.row_split == 2 || a.row_split == 4 ? a.row_split : (tiles >= 128 ? 1 : tiles >= 64 ? 2 : 4);
    dim3 grid((a.T + BN - 1) / BN, a.B, split);
    hipLaunchKernelGGL((conv_k5q_kernel<CIN, LN>), grid, dim3(256 / split), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// Conv1d(cin -> 256, k = 5, padding 2) + bias [+ ReLU] in its F(4,3) form (a->wf = to_wino43_iter_fragments(taps = 5) of the same weights), optionally
// with LayerNorm(256) (a->ln_g / ln_b / ln_eps) applied to the input while it is staged.  Returns 0, -2 (shape not covered) or -3 (HIP error).
extern "C" int cmtts_launch_conv_k5q(const ConvXlArgs* a, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (a->C != CO || a->k != KT || a->dil != 1 || a->T < 1 || a->B < 1 || a->res || a->accum || a->slope != 1.0f || !a->wf || !a->bias) return -2;
    const int cin = a->cin ? a->cin : a->C;
    if (a->ln_g && (!a->ln_b || cin != 256)) return -2;
    if (cin == 256) return a->ln_g ? launch_k5q<256, true>(*a, s) : launch_k5q<256, false>(*a, s);
    if (cin == 128 && !a->ln_g) return launch_k5q<128, false>(*a, s);
    return -2;
}
