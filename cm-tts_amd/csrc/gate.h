// sigmoid(g) * tanh(f) of the gated residual block (model/blocks.py:678-679), shared by the fused
// kernel and the generic conv kernel's EPI_GATED epilogue so both paths stay bitwise identical.
// Hardware exp (v_exp_f32) and reciprocal (v_rcp_f32) based: absolute error <= ~3e-7 on outputs in
// (-1, 1), far inside the 1e-3 parity bound; the exact expf/tanhf/div forms cost ~4x the VALU work in
// an epilogue that one wave per SIMD cannot hide.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float cmtts_gate(float g, float f) {
    const float s = __frcp_rn(1.0f + __expf(-g));                  // sigmoid(g)
    const float th = 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * f));   // tanh(f); saturates cleanly at +-1
    return s * th;
}
