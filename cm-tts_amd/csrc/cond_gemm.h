// Arguments of the X-resident stacked GEMM (cond_gemm.hip).
#pragma once

struct CondGemmArgs {
    const float* X;       // [B][K = 256][T]
    const float* Wf;      // fragment-order weights [K/8][M/32][64][4] (k-major [K][M] re-packed by cmtts_finalize)
    const float* bias;    // [M]
    float* Y;             // [B][M][T]
    int B, T, M, K;
    int force;            // take the kernel even where the generic one would finish sooner (tests)
};

#ifdef __cplusplus
extern "C" {
#endif
int cmtts_launch_cond_gemm(const CondGemmArgs* a, void* stream);   // 0, -2 (unsupported shape), -3 (HIP error)
// the same GEMM with 16-bit operands (cond_gemm16.hip): wf16 = to_fragment16 of the stacked weights, mode 1 = bf16, 2 = fp16
int cmtts_launch_cond_gemm16(const CondGemmArgs* a, const void* wf16, int mode, void* stream);
#ifdef __cplusplus
}
#endif
