// Shared epilogue of the implicit-GEMM Conv1D kernels (fp32 MFMA: conv_mfma.hip, 16-bit operands:
// conv_mfma16.hip): bias, alpha, activation, residual (+ per-batch channel vector), division, accumulate,
// length mask, strided / split outputs — see ConvOut in conv_args.h.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// erf in fp32 without branches: both ranges evaluated, one selected (|a| <= 0.927734375: a + a P(a^2); above:
// sign(a) (1 - exp(Q(|a|)))), < 1 ulp with an exact exp (checked against float64 over [-6, 6]); the library erff costs
// ~4x as many wave instructions because its two ranges sit under exec-mask branches that a mixed wave runs both of.
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);     // r in [-inf, -0.9]: v_exp_f32's relative error (1 ulp + the scaled argument's) stays below 1e-7 absolute
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_GELU_ERF: return v * 0.5f * (1.0f + erf_fast(v * 0.70710678118654752440f));
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// Epilogue of one 32x32 accumulator tile (16 registers per lane: rows m_base + acc_row(r), one
// column n).  All optional operands (bias, per-batch vector, residual, old Y for accumulate) are
// first gathered with UNCONDITIONAL loads from clamped in-bounds addresses — 16 loads in flight —
// and only the final store is predicated; a load under a per-lane branch would be serialised with
// a vmcnt(0) each.  Branches on the ConvOut pointers are wave-uniform (scalar).
__device__ __forceinline__ void epi_tile(const ConvOut& o, const f32x16& acc, int m_base, int rbase, int n, int M, int N,
                                         int zq, int zr) {
    const int t = n * o.ostride + o.ooff_base + zr * o.ooff_mul;
    const bool ok_n = n < N && t >= 0 && t < o.Tout;
    const int t_c = min(max(t, 0), o.Tout - 1);
    // wave-uniform bases (SGPR pairs) + 32-bit unsigned per-lane offsets: one VGPR per address
    float* __restrict__ yb = o.Y + (zq * o.y_zs0 + zr * o.y_zs1);
    const float* __restrict__ rb = o.res ? o.res + (zq * o.r_zs0 + zr * o.r_zs1) : nullptr;
    const float* __restrict__ vb = o.bvec ? o.bvec + zq * o.bvec_zs : nullptr;
    const bool keep = !(o.lens && (int64_t)t >= o.lens[zq]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // two batches of 8 rows
        float bi[8], rv[8], yv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = h * 8 + q;
            const int m_c = min(m_base + (r & 3) + 8 * (r >> 2) + rbase, M - 1);
            const unsigned row = (unsigned)(m_c - o.row_off);
            bi[q] = o.bias ? o.bias[(unsigned)m_c] : 0.f;
            rv[q] = 0.f;
            if (rb) rv[q] = rb[row * (unsigned)o.ldr + (unsigned)t_c];
            if (vb) rv[q] += vb[(unsigned)m_c];
            yv[q] = o.accum ? yb[row * (unsigned)o.ldy + (unsigned)t_c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = h * 8 + q;
            const int m = m_base + (r & 3) + 8 * (r >> 2) + rbase;
            float v = acc[r];
            if (o.bias) v += bi[q];
            v *= o.alpha;
            v = act_apply(v, o.act);
            if (rb || vb) v += rv[q];
            if (o.div != 1.0f) v = v / o.div;
            if (o.rmul != 0.0f) v = v * o.rmul;
            if (o.accum) v += yv[q];
            if (!keep) v = 0.f;
            if (ok_n && m < M) yb[(unsigned)(m - o.row_off) * (unsigned)o.ldy + (unsigned)t] = v;
        }
        asm volatile("" ::: "memory");
    }
}

// The common case with everything known at compile time (X-resident kernels of the text side): bias, alpha, ACT, residual,
// length mask; unit output stride, no row offset / division / accumulate / per-batch vector.  Same operations in the same
// order as epi_tile => identical bits.
__device__ __forceinline__ bool epi_simple(const ConvOut& o) {
    return o.ostride == 1 && o.ooff_base == 0 && o.ooff_mul == 0 && o.row_off == 0 && o.div == 1.0f && o.rmul == 0.0f && !o.accum && !o.bvec;
}
template <int ACT>
__device__ __forceinline__ void epi_tile_simple(const ConvOut& o, const f32x16& acc, int m_base, int rbase, int n, int M, int N, int zq) {
    const bool ok_n = n < N && n < o.Tout;
    const int t_c = min(n, o.Tout - 1);
    float* __restrict__ yb = o.Y + zq * o.y_zs0;
    const float* __restrict__ rb = o.res ? o.res + zq * o.r_zs0 : nullptr;
    const bool keep = !(o.lens && (int64_t)n >= o.lens[zq]);
    float bi[16], rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m_c = min(m_base + (r & 3) + 8 * (r >> 2) + rbase, M - 1);
        bi[r] = o.bias ? o.bias[(unsigned)m_c] : 0.f;
        rv[r] = rb ? rb[(unsigned)m_c * (unsigned)o.ldr + (unsigned)t_c] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m_base + (r & 3) + 8 * (r >> 2) + rbase;
        float v = acc[r];
        if (o.bias) v += bi[r];
        v *= o.alpha;
        v = act_apply(v, ACT);
        if (rb) v += rv[r];
        if (!keep) v = 0.f;
        if (ok_n && m < M) yb[(unsigned)m * (unsigned)o.ldy + (unsigned)n] = v;
    }
}
