// Arguments of the X-resident stacked GEMM (cond_gemm.hip).
#pragma once

struct CondGemmArgs {
    const float* X;       // [B][K = 256][T]
    const float* Wf;      // fragment-order weights [K/8][M/32][64][4] (k-major [K][M] re-packed by cmtts_finalize)
    const float* bias;    // [M]
    float* Y;             // [B][M][T]
    int B, T, M, K;
    int force;            // take the kernel even where the generic one would finish sooner (tests)
    int flat;             // 1: tiles run over the columns of all utterances as one axis (c = b * T + t): no per-utterance padding to 64 columns
    int row_split;        // > 1: that many workgroups per frame tile, each a contiguous share of the 512-row passes (few tiles, many rows)
};

#ifdef __cplusplus
extern "C" {
#endif
int cmtts_launch_cond_gemm(const CondGemmArgs* a, void* stream);   // 0, -2 (unsupported shape), -3 (HIP error)
// cp[b][r][t] = (mel2ph[b][t] > 0 ? p1[b][r][mel2ph - 1] : 0) + p2[r][pidx[b][t]]   (p1 [B][M][ldp], p2 [M][ld2], cp [B][M][T])
int cmtts_launch_cond_expand(const float* p1, int ldp, int L, const float* p2, int ld2, const int64_t* mel2ph, const int64_t* pidx,
                             float* cp, int B, int M, int T, void* stream);
// the same GEMM with 16-bit operands (cond_gemm16.hip): wf16 = to_fragment16 of the stacked weights, mode 1 = bf16, 2 = fp16
int cmtts_launch_cond_gemm16(const CondGemmArgs* a, const void* wf16, int mode, void* stream);
#ifdef __cplusplus
}
#endif
