// HiFi-GAN ResBlock convs (dilation 1; 3 / 5 by residue classes, below), fp32, in a Winograd F(4,3) form (round 5) — the X-resident single-conv kernel of resblock_pair.hip
// (conv_xl_kernel / conv_xlw_kernel; hifigan/models.py:96-103: xt = c1(leaky_relu(x)); x = c2(leaky_relu(xt)) + x) with the products of
// denoiser_persist.hip's WINO == 2 instances: outputs in QUADS (4q .. 4q + 3), every group of three consecutive taps as F(4,3) over the
// points 0, +-1, +-2, inf — six products per quad and group instead of twelve (conv_xlw_kernel's F(2,3) groups: eight) — all groups of a conv
// into the same six transform-domain accumulators (the output transform is linear):
//
//   k = 3:  one group                                   6 products per quad (direct 12, F(2,3) 8)
//   k = 7:  two groups + the seventh tap on its own      16                  (direct 28, F(2,3) tap groups 20)
//   k = 11: four groups, the twelfth tap zero            24                  (direct 44, F(2,3) tap groups 30)
//
// A group at tap offset o reads d0..d5 = X[4q + o .. 4q + o + 5] (X = the activated input tile, column j = frame t0 - (k-1)/2 + j) and forms
//   V0 = 4 d0 - 5 d2 + d4, V1 = (d4 - 4 d2) + (d3 - 4 d1), V2 = (d4 - 4 d2) - (d3 - 4 d1), V3 = (d4 - d2) + 2 (d3 - d1), V4 = (d4 - d2) - 2 (d3 - d1),
//   V5 = 4 d1 - 5 d3 + d5;   M_p += U_p V_p with U0 = g0/4, U1 = -(g0+g1+g2)/6, U2 = -(g0-g1+g2)/6, U3 = g0/24 + g1/12 + g2/6, U4 = g0/24 - g1/12 + g2/6, U5 = g2;
// a single tap g at offset o (x0..x3 = X[4q + o ..]) enters the same accumulators as M0 += g (x0 - x2), M1 += g/2 (x1 + x2), M2 += g/2 (x2 - x1),
// M5 += g (x3 - x1);   y0 = M0 + (M1 + M2) + (M3 + M4), y1 = (M1 - M2) + 2 (M3 - M4), y2 = (M1 + M2) + 4 (M3 + M4), y3 = (M1 - M2) + 8 (M3 - M4) + M5.
//
// One n-tile of v_mfma_f32_16x16x4_f32 = one transform of the tile's 16 quads = 64 output columns: lane (q = l & 15, k = l >> 4) owns quad q in
// channel 4 ks + k, so the six transforms of a quad sit in one lane and both transforms are in-lane.  A wave owns 64 output rows (four 16-row
// m-tiles x six transforms = 24 accumulators of 4 registers); a workgroup = C / 64 waves, all C rows of a 64-column tile; the activated tile
// [C][64 + k - 1] in LDS in natural column order (two aligned LDS reads per group and k-step).  Weights (cmtts_api.hip: to_wino43_iter_fragments,
// formed in double, rounded once) stream L2 -> VGPR in iteration order [k-step][wave][point][64 lanes][4 m-tiles] through a buffer descriptor
// (lane offset constant, step offset scalar) and a register ring; no barrier in the K loop.
// NOT bitwise the direct form (fp32 Winograd; measured on the waveform next to the F(2,3) form in tests/test_gpu_parity.py); restated in
// oracle/winograd_ref.py (conv1d_f43_taps) and checked against the plain conv on the CPU.
#include <hip/hip_runtime.h>
#include "resblock_pair.h"
#include "xcd_map.h"

#ifndef XCD_MAP
#define XCD_MAP 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int KT> struct QTab;      // entries of a k-step: kind (0 = F(4,3) group, 1 = single tap), tap offset, first point in the weight stream
template <> struct QTab<3> { static constexpr int NE = 1, NPT = 6; static constexpr int kind[1] = {0}, off[1] = {0}, pt0[1] = {0}; };
template <> struct QTab<7> { static constexpr int NE = 3, NPT = 16; static constexpr int kind[3] = {0, 0, 1}, off[3] = {0, 3, 6}, pt0[3] = {0, 6, 12}; };
template <> struct QTab<11> { static constexpr int NE = 4, NPT = 24; static constexpr int kind[4] = {0, 0, 0, 0}, off[4] = {0, 3, 6, 9}, pt0[4] = {0, 6, 12, 18}; };

// Dilation DIL = 3 / 5 (conv1 of a ResBlock's second / third pair): the outputs of a residue class t = r (mod DIL) form an UNDILATED conv on that
// class's subsequence, so the tile is 60 columns = DIL classes x (5 | 3) quads (15 of the 16 quad lanes), the activated tile lives in LDS class by
// class ([C][DIL][CP], CP = the class's entries rounded up to 4) and lane (class r, quad m) reads, transforms and accumulates exactly as at dilation 1
// from base r CP + 4 m; only the staging scatter and the strided stores differ.
template <int C, int KT, int DIL>
__global__ __launch_bounds__(64 * (C / 64), 2) void conv_xlq_kernel(const ConvXlArgs a) {
    using TAB = QTab<KT>;
    constexpr int NWV = C / 64;
    constexpr int NE = TAB::NE, NPT = TAB::NPT;
    constexpr int PAD = DIL * ((KT - 1) / 2);
    constexpr int BN = DIL == 1 ? 64 : 60;                  // output columns per tile
    constexpr int QPC = BN / (4 * DIL);                     // quads per residue class: 16 / 5 / 3
    constexpr int XIN = BN + DIL * (KT == 11 ? KT : KT - 1);   // staged columns (k = 11: one more per class, under the zero twelfth tap)
    constexpr int CP = (XIN / DIL + 3) / 4 * 4;             // entries per class, 16-byte aligned quads: 68 / 72 / 76 at dilation 1
    constexpr int XW = DIL * CP;                            // row pitch
    constexpr int NKS = C / 4;                              // k-steps of four channels
    extern __shared__ __attribute__((aligned(16))) float Xs[];      // [C][DIL][CP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * BN;
    const int T = a.T;
    const float* xb = a.x + (long)b * a.bstride;
    if constexpr (DIL > 1 && C == 256) {      // (measured: at C = 256 the dword form, lanes along the frame axis, beats the 16-byte form's four scattered LDS writes per item by 4 %)
        const int tbase = t0 - PAD;
        constexpr int XBLK = (XIN + 63) / 64;
#pragma unroll
        for (int h = 0; h < 64; h += 16) {
            float v[XBLK][16];
#pragma unroll
            for (int jb = 0; jb < XBLK; ++jb) {
                const int t_c = min(max(tbase + jb * 64 + lane, 0), T - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[jb][r] = xb[(long)(w * 64 + h + r) * a.ld + t_c];
            }
#pragma unroll
            for (int jb = 0; jb < XBLK; ++jb) {
                const int j = jb * 64 + lane, t = tbase + j;
                const int jc = (j % DIL) * CP + j / DIL;      // class-major position
                if (j < XIN) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Xs[(w * 64 + h + r) * XW + jc] = (t >= 0 && t < T) ? leaky(v[jb][r], a.slope) : 0.f;
                }
            }
        }
    } else {
        // stage the activated tile: wave w its 64 rows as 16-byte loads (NV per row, item = (row, quad of columns) on consecutive lanes), ALL of them in
        // flight before the first LDS write — one round trip per tile where the dword form took four of 32 loads each (the tile's overhead, not its
        // K loop, is what the C = 128 / 64 instances lose to the ideal); zeros outside [0, T).  Dilation 3 / 5: the four columns of an item go to
        // their class-major positions one by one.  (Requesting the weight ring's first stages and the first residual operands in front of this
        // staging changed nothing: 64.3 vs 64.2 ms per batch — the other workgroups of the CU already cover those round trips.)
        const int tbase = t0 - PAD;
        constexpr int NV = (XIN + 3) / 4;
        f32x4 v[NV];
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = it * 64 + lane, r = idx / NV, q = idx - r * NV;
            const int t = tbase + 4 * q;
            const float* src = xb + (long)(w * 64 + r) * a.ld;
            if (t >= 0 && t + 3 < T) v[it] = *reinterpret_cast<const f32x4*>(src + t);      // (4-byte aligned: global_load_dwordx4 takes it)
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[it][c] = src[min(max(t + c, 0), T - 1)];
            }
        }
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int idx = it * 64 + lane, r = idx / NV, q = idx - r * NV;
            const int t = tbase + 4 * q;
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (t + c >= 0 && t + c < T) ? leaky(v[it][c], a.slope) : 0.f;
            if constexpr (DIL == 1) *reinterpret_cast<f32x4*>(Xs + (w * 64 + r) * XW + 4 * q) = o;
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 4 * q + c;
                    if (j < XIN) Xs[(w * 64 + r) * XW + (j % DIL) * CP + j / DIL] = o[c];
                }
            }
        }
    }
    __syncthreads();

    f32x4 M[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 6; ++p) M[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        constexpr int R = NE == 4 ? 4 : 3;                 // ring: entries in flight + the one in use
        constexpr int U = NE == 1 ? 6 : NE == 3 ? 2 : 1;   // k-steps per unrolled round: its U NE entries are a multiple of the ring depth AND even (the raw
                                                           // inputs alternate between two buffers), so ring slots and buffers are compile-time
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wf), 0, NKS * NWV * NPT * 1024, 0x00020000);
        const int voff = lane * 16;
        f32x4 A[R][6];
        // entry n of the round that starts at k-step ks0: (k-step ks0 + n / NE, entry n % NE); past the last k-step the loads are out of range (zeros)
        auto load_a = [&](f32x4 (&dst)[6], int ks0, int n) {
            const int e = n % NE, ks = ks0 + n / NE;
            const int soff = ((ks * NWV + w) * NPT + TAB::pt0[e]) * 1024;
#pragma unroll
            for (int p = 0; p < 6; ++p)
                if (p < (TAB::kind[e] ? 4 : 6)) dst[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + p * 1024, soff, 0));
        };
        // raw inputs of an entry: two aligned LDS reads (three at tap offset 3); lane's quad at column 4 q of row 4 ks + (lane >> 4)
        const int lq = min(lane & 15, DIL * QPC - 1);      // (dilation 3 / 5: the sixteenth quad lane repeats the fifteenth; it stores nothing)
        const float* xl = Xs + (lane >> 4) * XW + (lq / QPC) * CP + 4 * (lq % QPC);
        float D[2][6];
        auto load_d = [&](float (&d)[6], int ks0, int n) {
            const int e = n % NE, ks = ks0 + n / NE;
            const float* r = xl + ks * (4 * XW);
            const int o = TAB::off[e];
            if (TAB::kind[e]) {             // single tap at offset 6: x0..x3 = columns 6 .. 9
                const f32x2 p = *reinterpret_cast<const f32x2*>(r + o), q = *reinterpret_cast<const f32x2*>(r + o + 2);
                d[0] = p.x; d[1] = p.y; d[2] = q.x; d[3] = q.y; d[4] = 0.f; d[5] = 0.f;
            } else if (o == 0) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(r);
                const f32x2 q = *reinterpret_cast<const f32x2*>(r + 4);
                d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3]; d[4] = q.x; d[5] = q.y;
            } else if (o == 3) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(r + 4);
                d[0] = r[3]; d[1] = p[0]; d[2] = p[1]; d[3] = p[2]; d[4] = p[3]; d[5] = r[8];
            } else if (o == 6) {
                const f32x2 q = *reinterpret_cast<const f32x2*>(r + 6);
                const f32x4 p = *reinterpret_cast<const f32x4*>(r + 8);
                d[0] = q.x; d[1] = q.y; d[2] = p[0]; d[3] = p[1]; d[4] = p[2]; d[5] = p[3];
            } else {                        // offset 9: columns 9 .. 14 (14 under the zero tap)
                const f32x4 p = *reinterpret_cast<const f32x4*>(r + 8), q = *reinterpret_cast<const f32x4*>(r + 12);
                d[0] = p[1]; d[1] = p[2]; d[2] = p[3]; d[3] = q[0]; d[4] = q[1]; d[5] = q[2];
            }
        };
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_a(A[s], 0, s);
        load_d(D[0], 0, 0);
#pragma unroll 1
        for (int ks0 = 0; ks0 < NKS; ks0 += U) {
#pragma unroll
            for (int n = 0; n < U * NE; ++n) {
                const int e = n % NE, slot = n % R;
                float V[6];
                {
                    const float (&d)[6] = D[n & 1];
                    if (TAB::kind[e]) {
                        V[0] = d[0] - d[2]; V[1] = d[1] + d[2]; V[2] = d[2] - d[1]; V[3] = d[3] - d[1]; V[4] = 0.f; V[5] = 0.f;      // (V[3] feeds M5)
                    } else {
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
                }
                __builtin_amdgcn_sched_barrier(0);
                load_a(A[(slot + R - 1) % R], ks0, n + R - 1);
                load_d(D[(n + 1) & 1], ks0, n + 1);
                if (ks0 + n / NE < NKS) {
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        if (p < (TAB::kind[e] ? 4 : 6)) {
                            const int mp = TAB::kind[e] && p == 3 ? 5 : p;      // a single tap's fourth product goes to M5
#pragma unroll
                            for (int i = 0; i < 4; ++i) M[i][mp] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[slot][p][i], V[p], M[i][mp], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- output transform + epilogue: ((acc + bias) [relu]) + residual [+ y_old], the expressions of conv_xl_kernel
    float* yb = a.y + (long)b * a.bstride;
    const float* rb = a.res ? a.res + (long)b * a.bstride : nullptr;
    const int q4 = lane & 15, rq = lane >> 4;
    // this lane's quad: columns tq, tq + DIL, tq + 2 DIL, tq + 3 DIL
    const int tq = t0 + (DIL == 1 ? 4 * q4 : (q4 / QPC) + 4 * DIL * (q4 % QPC));
    const bool qv = q4 < DIL * QPC;
    const bool vec = DIL == 1 && ((a.ld & 3) == 0) && ((reinterpret_cast<size_t>(yb) & 15) == 0) && (!rb || (reinterpret_cast<size_t>(rb) & 15) == 0) && tq + 3 < T;
    // (two m-tiles per round trip: all residual / old-y loads of a pair of m-tiles in flight before the first use)
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += 2) {
        float bi[2][4];
        f32x4 xr[2][4], yo[2][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = w * 64 + 16 * (i0 + ii) + 4 * rq + r;
                bi[ii][r] = a.bias[row];
                const long o = (long)row * a.ld + tq;
                if (vec) {
                    xr[ii][r] = rb ? *reinterpret_cast<const f32x4*>(rb + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                    yo[ii][r] = a.accum ? *reinterpret_cast<const f32x4*>(yb + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const long oc = (long)row * a.ld + min(tq + c * DIL, T - 1);
                        xr[ii][r][c] = rb ? rb[oc] : 0.f;
                        yo[ii][r][c] = a.accum ? yb[oc] : 0.f;
                    }
                }
            }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + ii;
                const int row = w * 64 + 16 * i + 4 * rq + r;
                const float m0 = M[i][0][r], m1 = M[i][1][r], m2 = M[i][2][r], m3 = M[i][3][r], m4 = M[i][4][r], m5 = M[i][5][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                f32x4 y;
                y[0] = ((m0 + s12) + s34) + bi[ii][r];
                y[1] = __builtin_fmaf(2.f, d34, d12) + bi[ii][r];
                y[2] = __builtin_fmaf(4.f, s34, s12) + bi[ii][r];
                y[3] = (__builtin_fmaf(8.f, d34, d12) + m5) + bi[ii][r];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = y[c];
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    if (rb) v += xr[ii][r][c];
                    if (a.accum) v += yo[ii][r][c];
                    y[c] = v;
                }
                const long o = (long)row * a.ld + tq;
                if (vec) *reinterpret_cast<f32x4*>(yb + o) = y;
                else if (qv) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (tq + c * DIL < T) yb[o + c * DIL] = y[c];
                }
            }
    }
}

template <int C, int KT, int DIL>
int launch_xlq(const ConvXlArgs& a, hipStream_t stream) {
    constexpr int BN = DIL == 1 ? 64 : 60;
    constexpr int XIN = BN + DIL * (KT == 11 ? KT : KT - 1), CP = (XIN / DIL + 3) / 4 * 4;
    const size_t lds = (size_t)C * DIL * CP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xlq_kernel<C, KT, DIL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
        attr_set = true;
    }
    dim3 grid((a.T + BN - 1) / BN, a.B);
    hipLaunchKernelGGL((conv_xlq_kernel<C, KT, DIL>), grid, dim3(64 * (C / 64)), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int C, int KT>
int launch_xlq_d(const ConvXlArgs& a, hipStream_t s) {
    if (a.dil == 1) return launch_xlq<C, KT, 1>(a, s);
    if (a.dil == 3) return launch_xlq<C, KT, 3>(a, s);
    if (a.dil == 5) return launch_xlq<C, KT, 5>(a, s);
    return -2;
}

template <int C>
int launch_xlq_k(const ConvXlArgs& a, hipStream_t s) {
    if (a.k == 3) return launch_xlq_d<C, 3>(a, s);
    if (a.k == 7) return launch_xlq_d<C, 7>(a, s);
    if (a.k == 11) return launch_xlq_d<C, 11>(a, s);
    return -2;
}

}  // namespace

// The conv in its F(4,3) form (a->wf = to_wino43_iter_fragments of the same weights).  Returns 0, -2 (shape not covered: C = 64 / 128 / 256,
// k = 3 / 7 / 11, dilation 1 / 3 / 5, C_in = C; or a launch too small to pay for the transforms unless a->wino_force) or -3 (HIP error).
extern "C" int cmtts_launch_conv_xlq(const ConvXlArgs* a, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if ((a->cin && a->cin != a->C) || a->T < 1 || a->B < 1) return -2;
    if (!a->wino_force && (long)((a->T + 63) / 64) * a->B < 1024) return -2;
    if (a->C == 64) return launch_xlq_k<64>(*a, s);
    if (a->C == 128) return launch_xlq_k<128>(*a, s);
    if (a->C == 256) return launch_xlq_k<256>(*a, s);
    return -2;
}
