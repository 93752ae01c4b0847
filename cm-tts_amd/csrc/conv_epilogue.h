// Shared epilogue of the implicit-GEMM Conv1D kernels (fp32 MFMA: conv_mfma.hip, 16-bit operands:
// conv_mfma16.hip): bias, alpha, activation, residual (+ per-batch channel vector), division, accumulate,
// length mask, strided / split outputs — see ConvOut in conv_args.h.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_GELU_ERF: return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
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
            if (o.accum) v += yv[q];
            if (!keep) v = 0.f;
            if (ok_n && m < M) yb[(unsigned)(m - o.row_off) * (unsigned)o.ldy + (unsigned)t] = v;
        }
        asm volatile("" ::: "memory");
    }
}

