// fp32 -> 16-bit operand conversion for the reduced-precision kernels: one v_cvt_pk_{bf16,f16}_f32 per
// pair (round to nearest even), MODE 1 = bf16, 2 = fp16.  Returns lo | hi << 16.
#pragma once
#include <hip/hip_runtime.h>

typedef float cvt_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cvt_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 cvt_f16x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __forceinline__ unsigned pack16(float lo, float hi) {
    const cvt_f32x2 v = {lo, hi};
    if (MODE == 1) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cvt_bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cvt_f16x2));
}
