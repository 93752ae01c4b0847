// Tail of the persistent denoiser kernels (denoiser_persist.hip, denoiser_persist_lp.hip): the skip head of
// Denoiser.forward (model/modules.py:634-637: sum(skips)/sqrt(NL) -> skip_projection -> ReLU -> output_projection) and
// the sampler's post-scaling (karras_diffusion.py:406,852), in fp32, with the arithmetic of the generic conv epilogues
// and of mel_post_kernel in the same order (bitwise equal to the separate launches).  Every wave passes its TPW x 32 rows x
// 64 frames of the skip sum (MFMA accumulator layout, rows (w * TPW + i) * 32 ..; TPW = 1 for the 8-wave kernels, 2 for the 4-wave
// kernel of denoiser_persist4.hip); u_lds / z_lds are two [256][PT_LD] fp32 LDS buffers that no wave reads any more.
#pragma once
#include <hip/hip_runtime.h>
#include "persist_args.h"

namespace persist_tail {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PT_C = 256, PT_NT = 2, PT_LD = 68, PT_RING = 6;

__device__ __forceinline__ int pt_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float pt_ldg(const float* base, unsigned idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 4u));
}
__device__ __forceinline__ int pt_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// xold / noise / out: the utterance batch's [B][T][n_mels] tensors (a.xold / a.noise / a.out, or a ragged group's); Tc: frames at and
// beyond Tc are not stored (Tc = T unless the utterance is trimmed)
// skip = TPW x PT_NT accumulator tiles, [i * PT_NT + j]
template <int TPW = 1>
__device__ __forceinline__ void run(const PersistArgs& a, float* u_lds, float* z_lds, const f32x16* skip, int w, int lane,
                                    int b, int t0, int T, const float* xold, const float* noise, float* out, int Tc) {
    constexpr int C = PT_C, NT = PT_NT, U_LD = PT_LD, RING = PT_RING;
    const int l31 = lane & 31, khalf = lane >> 5;
    auto opaque = [](int v) { return pt_opaque(v); };
    auto acc_row = [](int r, int ln) { return pt_row(r, ln); };
    auto ldg = [](const float* p, unsigned i) { return pt_ldg(p, i); };
    // ---- skip head in-kernel: sum(skips)/sqrt(NL) -> skip_projection -> ReLU -> output_projection -> c_out*F + c_skip*x
    // (+ re-noising), the arithmetic of the generic conv epilogues and of mel_post_kernel, in the same order.
    {
        const int ln = opaque(lane), c31 = ln & 31;
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    u_lds[((w * TPW + i) * 32 + acc_row(r, ln)) * U_LD + j * 32 + c31] = skip[i * NT + j][r] / a.skip_div;
    }
    __syncthreads();   // (A) skip tile staged; every wave has left the last output projection (z is free)
    constexpr int NGC = C / 8;
    auto load_bt = [&](float (&dst)[4][NT], const float* src, int g) {
        const float* bs = src + (g * 8 + khalf) * U_LD + l31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j) dst[kk][j] = bs[2 * kk * U_LD + j * 32];
    };
    f32x16 h[NT];
    auto gemm_tile = [&](const float* wfrag, int mtiles, int mt, const float* src) {   // h = W[mt] * src, K = 256
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[j][r] = 0.f;
        f32x4 Af[RING];
        float Bt[2][4][NT];
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
            Af[s] = *reinterpret_cast<const f32x4*>(wfrag + (((long)min(s, NGC - 1) * mtiles + mt) * 64 + lane) * 4);
        load_bt(Bt[0], src, 0);
#pragma unroll 1
        for (int it = 0; it < NGC; it += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                Af[(s + RING - 1) % RING] =
                    *reinterpret_cast<const f32x4*>(wfrag + (((long)min(it + s + RING - 1, NGC - 1) * mtiles + mt) * 64 + lane) * 4);
                load_bt(Bt[(s + 1) & 1], src, min(it + s + 1, NGC - 1));
                __builtin_amdgcn_sched_barrier(0);
                if (it + s < NGC) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            h[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[s][kk], Bt[s & 1][kk][j], h[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int mrow0 = (w * TPW + i) * 32;
        gemm_tile(a.Wsf, C / 32, w * TPW + i, u_lds);          // skip_projection rows [mrow0, +32)
        const int ln = opaque(lane), c31 = ln & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + acc_row(r, ln);
            const float bi = ldg(a.bs, (unsigned)m);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v = h[j][r] + bi;
                v *= 1.0f;
                v = v > 0.f ? v : 0.f;
                z_lds[m * U_LD + j * 32 + c31] = v;
            }
        }
    }
    __syncthreads();   // (B) relu(skip_projection) complete
    const int otiles = (a.n_mels + 31) / 32;     // <= 4 (n_mels <= 128): one tile per wave of either layout
    if (w < otiles) {
        const int mrow0 = w * 32;
        gemm_tile(a.Wpf, otiles, w, z_lds);      // output_projection rows [32w, +32) of n_mels
        const int ln = opaque(lane), c31 = ln & 31;
        const int M = a.n_mels;
        // Round 5: every load of this epilogue is an UNCONDITIONAL load from a clamped address, all of an n-tile in flight before the first
        // use, and only the stores are predicated.  Written with the loads under the per-lane `m < M && t < Tc` branch (and the non-finite
        // check's flag read behind another) the compiler emitted, per element, bias load / wait / xold load / wait / noise load / wait /
        // store / wait: ~128 dependent round trips on the three waves that own the mel rows — 2-3 % of a launch.  Same expressions per
        // element, same bits.
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = ldg(a.bp, (unsigned)min(mrow0 + acc_row(r, ln), M - 1));
        bool bad = false;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = t0 + j * 32 + c31;
            const long ob = ((long)b * T + min(t, T - 1)) * M;
            float xo[16], nz[16];
            if (xold) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xo[r] = xold[ob + min(mrow0 + acc_row(r, ln), M - 1)];
            }
            if (noise) {
#pragma unroll
                for (int r = 0; r < 16; ++r) nz[r] = noise[ob + min(mrow0 + acc_row(r, ln), M - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + acc_row(r, ln);
                const bool ok = m < M && t < Tc;
                const float F = h[j][r] + (m < M ? bi[r] : 0.f);
                float v = a.c_out * F;
                if (xold) v = __builtin_fmaf(a.c_skip, xo[r], v);
                if (noise) v = __builtin_fmaf(nz[r] * a.nstd, 0.85f, v);
                if (ok) out[ob + m] = v;
                // a non-finite mel value (16-bit operands beyond the fp16 range, non-finite weights / inputs) is reported, not just returned
                bad |= ok && !(fabsf(v) <= 3.402823466e38f);
            }
        }
        // code 2 in the pinned word cmtts_poll_error() reads, unless an earlier error is still pending there
        if (bad && a.tmo && *(volatile unsigned*)a.tmo == 0u) *(volatile unsigned*)a.tmo = 2u;
    }
}

}  // namespace persist_tail
