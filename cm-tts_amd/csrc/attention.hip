// Fused multi-head self-attention of the FFT blocks (model/blocks.py:266-312: F.multi_head_attention_forward's
// softmax(q k^T / sqrt(dh) + key_padding_mask) v, no biases) for gfx950 — one launch per layer instead of three
// (S^T = K^T Q GEMM -> softmax_cols -> O = V P^T GEMM with the [2B, L, L] scores crossing HBM twice).
//
// Workgroup = one (utterance, head); wave w = queries 32w .. 32w+31 against ALL keys (L <= 192: up to 6 waves).
//   * K and V head tiles (dh = 128 channels x 64 keys per chunk) are staged through LDS, coalesced along the key axis, and
//     shared by the workgroup's waves; a wave's 32 query columns of Q live in registers as MFMA B operands;
//   * S^T = K^T Q on v_mfma_f32_32x32x2_f32: one 32x32 accumulator per 32 keys (<= 6), K^T fragments = conflict-free
//     ds_read_b32 (32 consecutive keys of one channel per half-wave);
//   * softmax over keys entirely in registers: a query's scores sit in one lane column of the accumulators (16 registers x
//     two lane halves x the key tiles), so the row maximum and sum are in-lane reductions plus ONE cross-half exchange
//     (__shfl_xor 32) — no LDS, no [L, L] buffer anywhere;
//   * O = V P^T feeds the probabilities to the matrix pipe straight from the accumulator registers: MFMA k-step r of key tile
//     m multiplies keys {32m + (r&3) + 8(r>>2) + 4h} — exactly what lane half h holds in register r — so P never moves; the
//     V fragment addresses follow that key order (V^T tile in LDS with a 65-float row stride: conflict-free).
// Numerics: same operations as the three-launch path (scale after the dot product, max-subtracted expf, division by the
// sum) in a different summation order (fp32: ~1e-7 relative); every (utterance, head, query block) is computed the same
// way whatever the batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "attention.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// No floating-point contraction in this file: attention_kernel and attention_qb_kernel must round s * scale - max the same way.  Left to the
// compiler (-ffp-contract=fast-honor-pragmas, HIP's default) the first fused it into one fma where product and difference share a basic block,
// the second — product before a barrier, difference behind it — could not: 1 ulp apart on every probability (__fmul_rn / __fsub_rn do not
// stop the fusion: they are plain operators to the optimiser).
#pragma clang fp contract(off)

namespace {

constexpr int DH = 128;          // head dimension (hidden 256 / 2 heads)
constexpr int KB = 64;           // keys per staged chunk
constexpr int VLD = KB + 1;      // row stride of the V tile [DH][KB]: lanes walk down the channels

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int NMT>               // 32-key tiles = waves (query blocks): L in (32 (NMT-1), 32 NMT]
__global__ __launch_bounds__(64 * NMT) void attention_kernel(const AttnArgs a) {
    constexpr int NCH = (NMT + 1) / 2;
    constexpr int NTHREADS = 64 * NMT;
    __shared__ __attribute__((aligned(16))) float tile[DH * VLD];     // K chunks [d][KB], then V chunks [d][VLD] (padded rows)
    float* Ks = tile;
    float* Vs = tile;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int z = blockIdx.x, b = z / a.H, h = z - b * a.H;
    const int L = a.L, ld = a.ld;
    const int len = min((int)a.lens[b], L);
    const float* qb = a.qkv + (long)b * a.bstride + (long)(h * DH) * ld;                   // Q_h [DH][ld]
    const float* kb = a.qkv + (long)b * a.bstride + (long)(a.H * DH + h * DH) * ld;        // K_h
    const float* vb = a.qkv + (long)b * a.bstride + (long)(2 * a.H * DH + h * DH) * ld;    // V_h

    // ---- this wave's queries as B operands: lane (i = l31, khalf) holds Q[d = 2 kk + khalf][32 w + i] for kk < 64
    float Qr[DH / 2];
    {
        const int i_c = min(32 * w + l31, L - 1);
#pragma unroll
        for (int kk = 0; kk < DH / 2; ++kk) Qr[kk] = qb[(long)(2 * kk + khalf) * ld + i_c];
    }

    auto stage = [&](const float* src, float* dst, int dst_ld, int key0) {
        // [DH][KB] tile, keys beyond L zero-filled; float4 loads along the key axis (rows are 16-B aligned: ld % 4 == 0).
        // ALL of a thread's loads are issued (unconditional, clamped addresses) before the first LDS write: one memory round
        // trip per chunk instead of one per float4
        constexpr int NV = DH * (KB / 4);
        constexpr int PER = (NV + NTHREADS - 1) / NTHREADS;
        f32x4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = min(tid + u * NTHREADS, NV - 1);
            const int d = idx / (KB / 4), c4 = idx - d * (KB / 4);
            const int j = min(key0 + 4 * c4, ld - 4);
            v[u] = *reinterpret_cast<const f32x4*>(src + (long)d * ld + j);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < NV) {
                const int d = idx / (KB / 4), c4 = idx - d * (KB / 4);
                const int j = key0 + 4 * c4;
                const bool in = j <= ld - 4;          // a float4 starting beyond the row lies wholly beyond L
                float* p = dst + d * dst_ld + 4 * c4;
                p[0] = in && j + 0 < L ? v[u][0] : 0.f;
                p[1] = in && j + 1 < L ? v[u][1] : 0.f;
                p[2] = in && j + 2 < L ? v[u][2] : 0.f;
                p[3] = in && j + 3 < L ? v[u][3] : 0.f;
            }
        }
    };

    // ---- S^T[key][query] = sum_d K[d][key] Q[d][query]
    f32x16 S[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[m][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c > 0) __syncthreads();
        stage(kb, Ks, KB, c * KB);
        __syncthreads();
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl) {
            const int m = 2 * c + mtl;
            if (m < NMT) {
                const float* ks = Ks + khalf * KB + mtl * 32 + l31;
                float av = ks[0];
#pragma unroll
                for (int kk = 0; kk < DH / 2; ++kk) {
                    const float nav = kk + 1 < DH / 2 ? ks[(kk + 1) * 2 * KB] : 0.f;
                    S[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Qr[kk], S[m], 0, 0, 0);
                    av = nav;
                }
            }
        }
    }

    // ---- softmax over keys, per query column, in registers
    {
        float mx = -INFINITY;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * m + acc_row(r, lane);
                const float v = S[m][r] * a.scale;      // (a rounded product: no contraction in this file, above)
                S[m][r] = v;
                if (key < len) mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * m + acc_row(r, lane);
                const float e = key < len ? expf(S[m][r] - mx) : 0.f;
                S[m][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[m][r] = S[m][r] / (sum > 0.f ? sum : 1.f);     // len == 0: all probabilities 0
    }

    // ---- O[d][query] = sum_key V[d][key] P[key][query]; P stays where the accumulators left it
    __syncthreads();                 // every wave is done with the last K chunk: the tile is re-used for V
    f32x16 O[DH / 32];
#pragma unroll
    for (int mt = 0; mt < DH / 32; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[mt][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c > 0) __syncthreads();
        stage(vb, Vs, VLD, c * KB);
        __syncthreads();
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl) {
            const int m = 2 * c + mtl;
            if (m < NMT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;     // key of this k-step for this lane half
#pragma unroll
                    for (int mt = 0; mt < DH / 32; ++mt)
                        O[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(mt * 32 + l31) * VLD + kl], S[m][r], O[mt], 0, 0, 0);
                }
            }
        }
    }

    // ---- store channel-major [H*DH][ld], queries contiguous
    float* ob = a.out + (long)b * a.obstride + (long)(h * DH) * ld;
    const int i = 32 * w + l31;
    if (i < L) {
#pragma unroll
        for (int mt = 0; mt < DH / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ob[(long)(mt * 32 + acc_row(r, lane)) * ld + i] = O[mt][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same attention with the QUERIES split over workgroups too (L <= 128).  attention_kernel above runs one (utterance, head)
// per workgroup: 64 workgroups at B = 32 — a quarter of the chip — each wave walking all key tiles twice behind four staging round
// trips (27 us per FFT block for 10 us of MFMAs per wave).  Here a workgroup is one (utterance, head, block of 32 queries), four waves:
//   * K_h and V_h are staged WHOLE (<= 128 keys: one round trip for K, V requested behind it and written while the scores form);
//   * wave j forms the score tile of keys 32 j .. 32 j + 31 (one 64-step chain instead of NMT), the column maxima meet in LDS;
//   * the un-normalised probabilities of all tiles go through LDS in the accumulator layout, every wave adds them up in
//     attention_kernel's order (tile by tile, register by register, then the other lane half) and divides — so all four hold the
//     same P, bit for bit what attention_kernel holds;
//   * wave w forms output channels 32 w .. 32 w + 31 over all keys (one 16 NMT-step chain instead of four).
// Every chain has attention_kernel's operands in attention_kernel's order: BITWISE equal to it (tests/test_gpu_parity.py::
// test_attention_qb_bitwise); 3 x the workgroups, a third of the serial MFMA work per wave, two staging round trips instead of four.
template <int NMT>               // 32-key tiles (= query blocks of the launch): L in (32 (NMT-1), 32 NMT], NMT <= 4
__global__ __launch_bounds__(256) void attention_qb_kernel(const AttnArgs a) {
    constexpr int NK = 32 * NMT;         // staged keys
    constexpr int VL = NK + 1;           // row stride of the V tile [DH][VL]: lanes walk down the channels
    extern __shared__ __attribute__((aligned(16))) float qsm[];
    float* Ks = qsm;                     // [DH][NK]
    float* Vs = Ks + DH * NK;            // [DH][VL]
    float* Es = Vs + ((DH * VL + 3) & ~3);      // [NMT][16][64]: exp(s - max) in the accumulator layout of the wave that formed it
    float* Mx = Es + NMT * 16 * 64;      // [4][32] column maxima per wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int z = blockIdx.x, b = z / a.H, h = z - b * a.H, qb = blockIdx.y;
    const int L = a.L, ld = a.ld;
    const int len = min((int)a.lens[b], L);
    const float* qp = a.qkv + (long)b * a.bstride + (long)(h * DH) * ld;
    const float* kp = a.qkv + (long)b * a.bstride + (long)(a.H * DH + h * DH) * ld;
    const float* vp = a.qkv + (long)b * a.bstride + (long)(2 * a.H * DH + h * DH) * ld;

    constexpr int NV = DH * (NK / 4);
    constexpr int PER = NV / 256;        // 8 NMT float4 per thread
    f32x4 kv[PER], vv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = tid + u * 256;
        const int d = idx / (NK / 4), c4 = idx - d * (NK / 4);
        kv[u] = *reinterpret_cast<const f32x4*>(kp + (long)d * ld + min(4 * c4, ld - 4));
    }
    // this block's queries as B operands (wave j < NMT multiplies them with key tile j): lane (i = l31, khalf) holds Q[2 kk + khalf][32 qb + i]
    float Qr[DH / 2];
    {
        const int i_c = min(32 * qb + l31, L - 1);
#pragma unroll
        for (int kk = 0; kk < DH / 2; ++kk) Qr[kk] = qp[(long)(2 * kk + khalf) * ld + i_c];
    }
    auto put = [&](const f32x4 (&v)[PER], float* dst, int dst_ld) {      // keys beyond L zero-filled (a float4 that starts beyond ld - 4 lies wholly beyond L)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + u * 256;
            const int d = idx / (NK / 4), c4 = idx - d * (NK / 4);
            const int j = 4 * c4;
            const bool in = j <= ld - 4;
            float* p = dst + d * dst_ld + j;
            p[0] = in && j + 0 < L ? v[u][0] : 0.f;
            p[1] = in && j + 1 < L ? v[u][1] : 0.f;
            p[2] = in && j + 2 < L ? v[u][2] : 0.f;
            p[3] = in && j + 3 < L ? v[u][3] : 0.f;
        }
    };
    put(kv, Ks, NK);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = tid + u * 256;
        const int d = idx / (NK / 4), c4 = idx - d * (NK / 4);
        vv[u] = *reinterpret_cast<const f32x4*>(vp + (long)d * ld + min(4 * c4, ld - 4));
    }
    __syncthreads();

    // ---- S^T tile w: keys 32 w .. 32 w + 31 against the block's 32 queries
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    if (w < NMT) {
        const float* ks = Ks + khalf * NK + w * 32 + l31;
        float av = ks[0];
#pragma unroll
        for (int kk = 0; kk < DH / 2; ++kk) {
            const float nav = kk + 1 < DH / 2 ? ks[(kk + 1) * 2 * NK] : 0.f;
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Qr[kk], S, 0, 0, 0);
            av = nav;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * w + acc_row(r, lane);
            const float v = S[r] * a.scale;
            S[r] = v;
            if (key < len) mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (khalf == 0) Mx[w * 32 + l31] = mx;
    }
    put(vv, Vs, VL);
    __syncthreads();
    if (w < NMT) {
        float mx = Mx[l31];
#pragma unroll
        for (int j = 1; j < NMT; ++j) mx = fmaxf(mx, Mx[j * 32 + l31]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * w + acc_row(r, lane);
            Es[(w * 16 + r) * 64 + lane] = key < len ? expf(S[r] - mx) : 0.f;
        }
    }
    __syncthreads();
    // ---- every wave: the probabilities of all key tiles, summed in attention_kernel's order
    float P[NMT][16];
    {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                P[m][r] = Es[(m * 16 + r) * 64 + lane];
                sum += P[m][r];
            }
        sum += __shfl_xor(sum, 32);
        const float den = sum > 0.f ? sum : 1.f;     // len == 0: all probabilities 0
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[m][r] = P[m][r] / den;
    }
    // ---- O rows 32 w .. 32 w + 31 = sum_key V[d][key] P[key][query]
    f32x16 O;
#pragma unroll
    for (int r = 0; r < 16; ++r) O[r] = 0.f;
    const float* vs = Vs + (w * 32 + l31) * VL + 4 * khalf;
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            O = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[m * 32 + (r & 3) + 8 * (r >> 2)], P[m][r], O, 0, 0, 0);
    float* ob = a.out + (long)b * a.obstride + (long)(h * DH) * ld;
    const int i = 32 * qb + l31;
    if (i < L) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(long)(w * 32 + acc_row(r, lane)) * ld + i] = O[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Long sequences (L > 192: the FastspeechDecoder over the frame axis, texts up to max_seq_len = 1000 — config/LJSpeech/model.yaml:55;
// round 3, VERDICT r02 missing #5): the same attention with the keys walked in 64-key chunks and an ONLINE softmax, so that neither
// the scores nor the probabilities of a query ever exist in full.  Workgroup = (utterance, head, block of NWQ x 32 queries): the grid
// also splits the QUERIES, not only the heads (B H ceil(L / 128) workgroups).  Per chunk:
//   S = K_c^T Q (two 32-key accumulators)  ->  m' = max(m, max_keys S), alpha = exp(m - m'), P = exp(S - m') (0 on padded keys),
//   l = l alpha + sum_keys P,  O = O alpha + V_c P^T  (P fed from the accumulator registers exactly as above);  finally O / l.
// K and V chunks have their own LDS tiles (65 KB together: two barriers per chunk); chunks that lie wholly beyond the utterance's
// length are skipped.  Same operations as the two-pass softmax up to the order of roundings (fp32, ~1e-7 relative).
template <int NWQ>
__global__ __launch_bounds__(64 * NWQ) void attention_long_kernel(const AttnArgs a) {
    constexpr int NTHREADS = 64 * NWQ;
    extern __shared__ __attribute__((aligned(16))) float ltile[];
    float* Ks = ltile;                       // [DH][KB]
    float* Vs = ltile + DH * KB;             // [DH][VLD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int z = blockIdx.x, b = z / a.H, h = z - b * a.H;
    const int L = a.L, ld = a.ld;
    const int len = min((int)a.lens[b], L);
    const int q0 = blockIdx.y * (32 * NWQ) + 32 * w;
    const float* qb = a.qkv + (long)b * a.bstride + (long)(h * DH) * ld;
    const float* kb = a.qkv + (long)b * a.bstride + (long)(a.H * DH + h * DH) * ld;
    const float* vb = a.qkv + (long)b * a.bstride + (long)(2 * a.H * DH + h * DH) * ld;

    float Qr[DH / 2];
    {
        const int i_c = min(q0 + l31, L - 1);
#pragma unroll
        for (int kk = 0; kk < DH / 2; ++kk) Qr[kk] = qb[(long)(2 * kk + khalf) * ld + i_c];
    }
    f32x16 O[DH / 32];
#pragma unroll
    for (int mt = 0; mt < DH / 32; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[mt][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    constexpr int NV = DH * (KB / 4);
    constexpr int PER = (NV + NTHREADS - 1) / NTHREADS;
    const int nch = (len + KB - 1) / KB;
    for (int c = 0; c < nch; ++c) {
        const int key0 = c * KB;
        {   // stage K_c and V_c: all loads (unconditional, clamped addresses) before the first LDS write
            f32x4 kv[PER], vv[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int idx = min(tid + u * NTHREADS, NV - 1);
                const int d = idx / (KB / 4), c4 = idx - d * (KB / 4);
                const int j = min(key0 + 4 * c4, ld - 4);
                kv[u] = *reinterpret_cast<const f32x4*>(kb + (long)d * ld + j);
                vv[u] = *reinterpret_cast<const f32x4*>(vb + (long)d * ld + j);
            }
            if (c > 0) __syncthreads();          // every wave has finished the previous chunk's MFMAs
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int idx = tid + u * NTHREADS;
                if (idx < NV) {
                    const int d = idx / (KB / 4), c4 = idx - d * (KB / 4);
                    const int j = key0 + 4 * c4;
                    const bool in = j <= ld - 4;
                    float* pk = Ks + d * KB + 4 * c4;
                    float* pv = Vs + d * VLD + 4 * c4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = in && j + e < L;
                        pk[e] = ok ? kv[u][e] : 0.f;
                        pv[e] = ok ? vv[u][e] : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        f32x16 S[2];
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[mtl][r] = 0.f;
            const float* ks = Ks + khalf * KB + mtl * 32 + l31;
            float av = ks[0];
#pragma unroll
            for (int kk = 0; kk < DH / 2; ++kk) {
                const float nav = kk + 1 < DH / 2 ? ks[(kk + 1) * 2 * KB] : 0.f;
                S[mtl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Qr[kk], S[mtl], 0, 0, 0);
                av = nav;
            }
        }
        // online softmax update of this lane's query column (both lane halves hold the same queries, different keys)
        float cmx = -INFINITY;
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + 32 * mtl + acc_row(r, lane);
                const float v = S[mtl][r] * a.scale;
                S[mtl][r] = v;
                if (key < len) cmx = fmaxf(cmx, v);
            }
        cmx = fmaxf(cmx, __shfl_xor(cmx, 32));
        const float mnew = fmaxf(mrun, cmx);                  // finite: chunk c < nch holds at least one valid key
        const float alpha = expf(mrun - mnew);                // first chunk: exp(-inf) = 0 on O = 0, l = 0
        float csum = 0.f;
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + 32 * mtl + acc_row(r, lane);
                const float e = key < len ? expf(S[mtl][r] - mnew) : 0.f;
                S[mtl][r] = e;
                csum += e;
            }
        csum += __shfl_xor(csum, 32);
        lrun = lrun * alpha + csum;
        mrun = mnew;
#pragma unroll
        for (int mt = 0; mt < DH / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[mt][r] *= alpha;
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = mtl * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
#pragma unroll
                for (int mt = 0; mt < DH / 32; ++mt)
                    O[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(mt * 32 + l31) * VLD + kl], S[mtl][r], O[mt], 0, 0, 0);
            }
    }
    float* ob = a.out + (long)b * a.obstride + (long)(h * DH) * ld;
    const int i = q0 + l31;
    if (i < L) {
        const float inv = lrun > 0.f ? lrun : 1.f;            // len == 0: all probabilities 0
#pragma unroll
        for (int mt = 0; mt < DH / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ob[(long)(mt * 32 + acc_row(r, lane)) * ld + i] = O[mt][r] / inv;
    }
}

int g_attn_qb = 1;          // internal switch "attn_qb": L <= 128 on attention_qb_kernel (1) or attention_kernel (0); same bits

}  // namespace

extern "C" int cmtts_attention_set_qb(int on) { const int p = g_attn_qb; if (on == 0 || on == 1) g_attn_qb = on; return p; }

// 0 = launched, -2 = shape not covered (head_dim != 128, unaligned rows: the caller runs the three-launch path), -3 = HIP error
extern "C" int cmtts_launch_attention(const AttnArgs* ap, void* stream_) {
    const AttnArgs& a = *ap;
    hipStream_t s = (hipStream_t)stream_;
    if (a.B <= 0 || a.L <= 0) return 0;
    if (a.dh != DH || (a.ld & 3)) return -2;
    if (a.L > 192) {          // key-chunked online softmax, queries split over workgroups
        constexpr int NWQ = 4;
        const size_t lds = (size_t)(DH * KB + DH * VLD) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_long_kernel<NWQ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return -3;
            attr_set = true;
        }
        dim3 grid(a.B * a.H, (a.L + 32 * NWQ - 1) / (32 * NWQ));
        hipLaunchKernelGGL(attention_long_kernel<NWQ>, grid, dim3(64 * NWQ), lds, s, a);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    const int nmt = (a.L + 31) / 32;
    if (g_attn_qb && nmt <= 4) {      // queries split over workgroups too (round 6; bitwise attention_kernel)
        const size_t lds = (size_t)(DH * 32 * nmt + ((DH * (32 * nmt + 1) + 3) & ~3) + nmt * 16 * 64 + 4 * 32) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            const int mx = (int)((size_t)(DH * 128 + ((DH * 129 + 3) & ~3) + 4 * 16 * 64 + 4 * 32) * sizeof(float));
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_qb_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(attention_qb_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(attention_qb_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(attention_qb_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
                return -3;
            attr_set = true;
        }
        dim3 gq(a.B * a.H, nmt);       // the query blocks of one (utterance, head) are B H workgroup ids apart: the same XCD (its L2 holds their K / V) whenever B H % 8 == 0
        switch (nmt) {
            case 1: hipLaunchKernelGGL(attention_qb_kernel<1>, gq, dim3(256), lds, s, a); break;
            case 2: hipLaunchKernelGGL(attention_qb_kernel<2>, gq, dim3(256), lds, s, a); break;
            case 3: hipLaunchKernelGGL(attention_qb_kernel<3>, gq, dim3(256), lds, s, a); break;
            default: hipLaunchKernelGGL(attention_qb_kernel<4>, gq, dim3(256), lds, s, a); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    dim3 grid(a.B * a.H);
    switch (nmt) {
        case 1: hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(64), 0, s, a); break;
        case 2: hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(128), 0, s, a); break;
        case 3: hipLaunchKernelGGL(attention_kernel<3>, grid, dim3(192), 0, s, a); break;
        case 4: hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL(attention_kernel<5>, grid, dim3(320), 0, s, a); break;
        default: hipLaunchKernelGGL(attention_kernel<6>, grid, dim3(384), 0, s, a); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
