// HiFi-GAN ResBlock PAIR with k = 3, fp32, both convs in the F(4,3) form of conv_xlq.hip and `xt` kept on the CU (round 6, VERDICT r05 #4 ii):
//   xt = conv1(leaky_relu(x, 0.1))  [3 taps, dilation d in {1, 3, 5}];   y = conv2(leaky_relu(xt, 0.1)) + x  [3 taps, dilation 1]   (+= the MRF sum)
// (hifigan/models.py:96-103, one iteration of ResBlock.forward's loop).  As two conv_xlq launches the k = 3 pairs of the C = 128 stage are bound by the
// five tensor passes they move (x in, xt out, xt in, x again as the residual, y out: 1.6 GB per 0.69 ms launch, 47-56 % MFMA-busy) and at C = 64 the
// fused DIRECT pair kernel (resblock_pair.hip) is bound by its MFMAs (12 products per quad and conv where F(4,3) takes 6).  Here one workgroup does both:
//
//   * ONE LDS image, used in place: the activated x tile (class-major at dilation 3 / 5, exactly conv_xlq's layout) is consumed by conv1's K loop; behind a
//     barrier the same memory takes the activated xt tile in natural column order, which conv2's K loop reads.  C = 128: 35-41 KB per workgroup, as many
//     workgroups per CU as the two-launch form had (an image per conv would halve them: one wave per SIMD, nothing to overlap the staging with);
//   * conv2's one-frame halo is RECOMPUTED: conv1 produces xt on 16 quads = 64 frames [t0 - 1, t0 + 63) (dilation 1) or on 15 quads = 60 frames (dilation
//     3 / 5: DIL residue classes x (5 | 3) quads — the outputs of a class are an undilated conv on the class's subsequence), conv2 its 15 / 14 quads = 60 / 56
//     output frames [t0, t0 + BN): 7-11 % more MFMAs than the two launches' tiles, three tensor passes fewer.  xt outside [0, T) is ZERO (conv2 pads xt, it
//     does not see conv1 evaluated beyond the sequence);
//   * the K loops, transforms, weight streams (to_wino43_iter_fragments, k = 3: six points per k-step) and epilogue expressions are conv_xlq_kernel<C, 3, DIL>'s
//     and <C, 3, 1>'s.  NOT bitwise the two-launch form: an F(4,3) output is rounded from the six inputs of ITS quad, and this tile's conv1 quads start at
//     frame t0 - 1 where the single conv's start at a multiple of its own tile — the same products on quads one frame apart (measured <= 4.3e-6 on outputs of
//     a few units; against the direct form fp32 Winograd rounding, as for conv_xlq: tests/test_gpu_parity.py::test_conv_xlq_pair_vs_oracle).  Every tile is
//     computed alone, so an utterance's result does not depend on the batch it is in.
//
// Lane (q = l & 15, k = l >> 4) owns quad q of the tile in channel 4 ks + k; a wave owns 64 output rows (four 16-row m-tiles x six transforms); C / 64 waves.
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

// The k = 3 K loop of conv_xlq_kernel (one F(4,3) group per k-step of four channels): M[i][p] += U_p (rows 16 i ..) x V_p(d0..d5), d = six consecutive
// entries at xl + ks * 4 * XW.  Weight ring three deep through the buffer descriptor rs, raw inputs double-buffered, six k-steps per unrolled round.
template <int C, int XW>
__device__ __forceinline__ void kloop3(const float* xl, const __amdgpu_buffer_rsrc_t rs, const int w, const int lane, f32x4 (&M)[4][6]) {
    constexpr int NWV = C / 64, NKS = C / 4, R = 3, U = 6;
    const int voff = lane * 16;
    f32x4 A[R][6];
    auto load_a = [&](f32x4 (&dst)[6], int ks) {       // past the last k-step the loads are out of range (zeros, never multiplied)
        const int soff = ((ks * NWV + w) * 6) * 1024;
#pragma unroll
        for (int p = 0; p < 6; ++p) dst[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + p * 1024, soff, 0));
    };
    float D[2][6];
    auto load_d = [&](float (&d)[6], int ks) {
        const float* r = xl + min(ks, NKS - 1) * (4 * XW);
        const f32x4 p = *reinterpret_cast<const f32x4*>(r);
        const f32x2 q = *reinterpret_cast<const f32x2*>(r + 4);
        d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3]; d[4] = q.x; d[5] = q.y;
    };
#pragma unroll
    for (int s = 0; s < R - 1; ++s) load_a(A[s], s);
    load_d(D[0], 0);
#pragma unroll 1
    for (int ks0 = 0; ks0 < NKS; ks0 += U) {
#pragma unroll
        for (int n = 0; n < U; ++n) {
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
            load_a(A[(slot + R - 1) % R], ks0 + n + R - 1);
            load_d(D[(n + 1) & 1], ks0 + n + 1);
            if (ks0 + n < NKS) {
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) M[i][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[slot][p][i], V[p], M[i][p], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// output transform of one (m-tile, row) of a quad + bias: conv_xlq_kernel's expressions
__device__ __forceinline__ f32x4 out_quad(const f32x4 (&Mi)[6], int r, float bias) {
    const float m0 = Mi[0][r], m1 = Mi[1][r], m2 = Mi[2][r], m3 = Mi[3][r], m4 = Mi[4][r], m5 = Mi[5][r];
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    f32x4 y;
    y[0] = ((m0 + s12) + s34) + bias;
    y[1] = __builtin_fmaf(2.f, d34, d12) + bias;
    y[2] = __builtin_fmaf(4.f, s34, s12) + bias;
    y[3] = (__builtin_fmaf(8.f, d34, d12) + m5) + bias;
    return y;
}

template <int C, int DIL>
struct PairGeom {
    static constexpr int NWV = C / 64;
    static constexpr int QPC = DIL == 1 ? 16 : (DIL == 3 ? 5 : 3);   // conv1: quads per residue class
    static constexpr int NQ1 = DIL * QPC;                            // conv1 quad lanes: 16 / 15 / 15
    static constexpr int NX1 = 4 * NQ1;                              // xt frames computed: [t0 - 1, t0 - 1 + NX1)
    static constexpr int NQ2 = (NX1 - 2) / 4;                        // conv2 quads: 15 / 14 / 14
    static constexpr int BN = 4 * NQ2;                               // output frames per tile: 60 / 56 / 56
    static constexpr int XIN = NX1 + 2 * DIL;                        // staged x frames: [t0 - 1 - DIL, ...): 66 / 66 / 70
    static constexpr int CP = ((XIN + DIL - 1) / DIL + 3) / 4 * 4;   // entries per class, 16-byte aligned quads: 68 / 24 / 16
    static constexpr int XW1 = DIL * CP;                             // x row pitch: 68 / 72 / 80
    static constexpr int XW2 = 68;                                   // xt row pitch (columns 0 .. NX1 - 1 written, 0 .. 4 NQ2 + 1 read)
    static constexpr int LDS_FLOATS = C * (XW1 > XW2 ? XW1 : XW2);
};

template <int C, int DIL>
__global__ __launch_bounds__(64 * (C / 64), 2) void conv_xlq_pair3_kernel(const PairArgs a) {
    using G = PairGeom<C, DIL>;
    constexpr int NWV = G::NWV, QPC = G::QPC, NQ1 = G::NQ1, NQ2 = G::NQ2, BN = G::BN, XIN = G::XIN, CP = G::CP, XW1 = G::XW1, XW2 = G::XW2;
    constexpr int NKS = C / 4;
    extern __shared__ __attribute__((aligned(16))) float Xs[];      // [C][DIL][CP] (x, activated), then [C][XW2] (xt, activated)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * BN;
    const int T = a.T;
    const float* xb = a.x + (long)b * a.bstride;
    {
        // stage the activated x tile (conv_xlq_kernel's 16-byte form): wave w its 64 rows, every load in flight before the first LDS write; zeros outside [0, T)
        const int tbase = t0 - 1 - DIL;
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
            if constexpr (DIL == 1) *reinterpret_cast<f32x4*>(Xs + (w * 64 + r) * XW1 + 4 * q) = o;
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 4 * q + c;
                    if (j < XIN) Xs[(w * 64 + r) * XW1 + (j % DIL) * CP + j / DIL] = o[c];
                }
            }
        }
    }
    __syncthreads();

    const int q4 = lane & 15, rq = lane >> 4;
    f32x4 M[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 6; ++p) M[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    {   // ---- conv1: lane (class q1 / QPC, quad q1 % QPC); the sixteenth quad lane of the dilated forms repeats the fifteenth and writes nothing
        const int q1 = min(q4, NQ1 - 1);
        const float* xl = Xs + rq * XW1 + (q1 / QPC) * CP + 4 * (q1 % QPC);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(a.w1f)), 0, NKS * NWV * 6 * 1024, 0x00020000);
        kloop3<C, XW1>(xl, rs, w, lane, M);
    }
    __syncthreads();            // every wave has read its last x fragment: the image is free
    {   // ---- xt = leaky(conv1 + b1) into the image, natural column order: column j = frame t0 - 1 + j; zero outside [0, T)
        const int j0 = DIL == 1 ? 4 * q4 : (q4 / QPC) + 4 * DIL * (q4 % QPC);      // this lane's quad: columns j0, j0 + DIL, j0 + 2 DIL, j0 + 3 DIL
        const int tq = t0 - 1 + j0;
        const bool qv = q4 < NQ1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = w * 64 + 16 * i + 4 * rq + r;
                f32x4 y = out_quad(M[i], r, a.b1[row]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int t = tq + c * DIL;
                    y[c] = (t >= 0 && t < T) ? leaky(y[c], a.slope) : 0.f;
                }
                if constexpr (DIL == 1) *reinterpret_cast<f32x4*>(Xs + row * XW2 + j0) = y;
                else if (qv) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) Xs[row * XW2 + j0 + c * DIL] = y[c];
                }
            }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 6; ++p) M[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    {   // ---- conv2 (dilation 1): quad q of the outputs [t0, t0 + BN) reads xt frames t0 + 4 q - 1 .. = image columns 4 q .. 4 q + 5
        const int q2 = min(q4, NQ2 - 1);
        const float* xl = Xs + rq * XW2 + 4 * q2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(a.w2f)), 0, NKS * NWV * 6 * 1024, 0x00020000);
        kloop3<C, XW2>(xl, rs, w, lane, M);
    }

    // ---- y = ((conv2 + b2) + x) [+ y_old]: conv_xlq_kernel's epilogue at dilation 1 (two m-tiles per round trip)
    float* yb = a.y + (long)b * a.bstride;
    const int tq = t0 + 4 * q4;
    const bool qv = q4 < NQ2;
    const bool vec = ((a.ld & 3) == 0) && ((reinterpret_cast<size_t>(yb) & 15) == 0) && ((reinterpret_cast<size_t>(xb) & 15) == 0) && tq + 3 < T;
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += 2) {
        float bi[2][4];
        f32x4 xr[2][4], yo[2][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = w * 64 + 16 * (i0 + ii) + 4 * rq + r;
                bi[ii][r] = a.b2[row];
                const long o = (long)row * a.ld + tq;
                if (vec) {
                    xr[ii][r] = *reinterpret_cast<const f32x4*>(xb + o);
                    yo[ii][r] = a.accum ? *reinterpret_cast<const f32x4*>(yb + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const long oc = (long)row * a.ld + min(tq + c, T - 1);
                        xr[ii][r][c] = xb[oc];
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
                f32x4 y = out_quad(M[i], r, bi[ii][r]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = y[c];
                    v += xr[ii][r][c];
                    if (a.accum) v += yo[ii][r][c];
                    y[c] = v;
                }
                const long o = (long)row * a.ld + tq;
                if (qv) {
                    if (vec) *reinterpret_cast<f32x4*>(yb + o) = y;
                    else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (tq + c < T) yb[o + c] = y[c];
                    }
                }
            }
    }
}

template <int C, int DIL>
int launch_pair3(const PairArgs& a, hipStream_t stream) {
    using G = PairGeom<C, DIL>;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xlq_pair3_kernel<C, DIL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
        attr_set = true;
    }
    dim3 grid((a.T + G::BN - 1) / G::BN, a.B);
    hipLaunchKernelGGL((conv_xlq_pair3_kernel<C, DIL>), grid, dim3(64 * (C / 64)), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int C>
int launch_pair3_d(const PairArgs& a, hipStream_t s) {
    if (a.dil == 1) return launch_pair3<C, 1>(a, s);
    if (a.dil == 3) return launch_pair3<C, 3>(a, s);
    if (a.dil == 5) return launch_pair3<C, 5>(a, s);
    return -2;
}

}  // namespace

// The k = 3 pair in its fused F(4,3) form (a->w1f / a->w2f = to_wino43_iter_fragments of the two convs' weights).  Returns 0, -2 (shape not covered: C = 64 / 128,
// k = 3, dilation 1 / 3 / 5) or -3 (HIP error).  y must not alias x.
extern "C" int cmtts_launch_conv_xlq_pair(const PairArgs* a, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (a->k != 3 || a->T < 1 || a->B < 1 || !a->w1f || !a->w2f) return -2;
    if (a->C == 64) return launch_pair3_d<64>(*a, s);
    if (a->C == 128) return launch_pair3_d<128>(*a, s);
    return -2;
}
