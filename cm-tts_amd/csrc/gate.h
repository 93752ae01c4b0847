// sigmoid(g) * tanh(f) of the gated residual block (model/blocks.py:678-679), shared by the fused
// kernel and the generic conv kernel's EPI_GATED epilogue so both paths stay bitwise identical.
// Hardware exp (v_exp_f32) and reciprocal (v_rcp_f32, 1 ulp) based: absolute error <= ~3e-7 on outputs in
// (-1, 1), far inside the 1e-3 parity bound; the exact expf/tanhf/div forms cost ~4x the VALU work in
// an epilogue that nothing hides: next to v_mfma_f32_32x32x2_f32 every VALU instruction costs its own issue time
// (tools/filler_probe.hip: ~4 cycles per v_add_f32, ~8 per transcendental, with one OR two waves per SIMD — the
// fp32 matrix instruction runs on the vector ALUs).  Round 5: __frcp_rn compiles to the full IEEE division
// sequence on this toolchain (v_div_scale / v_rcp / 4 fma / v_div_fmas / v_div_fixup: ten instructions per
// reciprocal, 2/3 of the gate phase); __builtin_amdgcn_rcpf is the one instruction the comment above meant.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float cmtts_gate(float g, float f) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g));                  // sigmoid(g)
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * f));   // tanh(f); saturates cleanly at +-1
    return s * th;
}

// (o + residual) / sqrt(2) of the residual block (model/blocks.py:683) as a multiplication by the fp32 reciprocal — every form of the block
// (per-layer, split, persistent, 16-bit, the generic conv epilogue's ConvOut.rmul) uses this constant, so they stay bitwise equal to each
// other; against the reference's true division a result differs by at most one ulp.
#define CMTTS_RSQRT2 0.70710678118654752440f
