// Small HBM-bound kernels of the CM-TTS inference path (gfx950, wave64).
// Every kernel keeps the frame/phoneme axis on consecutive lanes (coalesced 256-B wave accesses);
// reductions over channels go through LDS, gathers rely on L2.  Each kernel cites the reference
// lines it implements (paths relative to the reference root).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "kernels.h"

namespace {

__device__ __forceinline__ float sin_exact(float a) { return (float)sin((double)a); }
__device__ __forceinline__ float cos_exact(float a) { return (float)cos((double)a); }

// Sinusoidal position embedding value (model/blocks.py:45-62): row p = [sin(p*w_j) | cos(p*w_j)],
// row 0 (padding) is all zeros.  omega[j] = exp(-j*ln(1e4)/(C/2-1)) is precomputed on the host.
__device__ __forceinline__ float pos_embed(int p, int c, int C, const float* omega, const float* tab, int tab_rows) {
    if (p == 0) return 0.f;
    if (p < tab_rows) return tab[(long)p * C + c];      // host-built table (same fp32-arg / f64-sin recipe)
    const int half = C >> 1;
    const int j = c < half ? c : c - half;
    const float arg = (float)p * omega[j];
    return c < half ? sin_exact(arg) : cos_exact(arg);
}

// positions = cumsum(flag) * flag over [0,T) for one batch row (utils/tools.py:810-822), computed by
// a 256-thread block into LDS `pos`.  `counts` is 256 ints of LDS.
template <typename FlagFn>
__device__ void block_positions(FlagFn flag, int T, int* pos, int* counts) {
    const int tid = threadIdx.x;
    const int seg = (T + 255) / 256;
    const int start = tid * seg;
    const int end = min(T, start + seg);
    int cnt = 0;
    for (int t = start; t < end; ++t) cnt += flag(t) ? 1 : 0;
    counts[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int c = counts[i]; counts[i] = run; run += c; }
    }
    __syncthreads();
    int run = counts[tid];
    for (int t = start; t < end; ++t) {
        const bool f = flag(t);
        run += f ? 1 : 0;
        pos[t] = f ? run : 0;
    }
    __syncthreads();
}

// ---- FastspeechEncoder.forward_embedding (model/modules.py:145-151) + first mask (:94)
// grid (B, C / EMB_CG): every workgroup recomputes the utterance's positions (a T-element scan) and fills
// EMB_CG channels, so a batch of 32 utterances is 256+ workgroups instead of 32.
constexpr int EMB_CG = 32;
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* texts, const int64_t* lens,
                                                           const float* E, const float* omega, const float* tab,
                                                           int tab_rows, float* x, int L, int ld, int C, float scale) {
    extern __shared__ int sh[];
    int* counts = sh;
    int* pos = sh + 256;
    const int b = blockIdx.x;
    const int64_t* tok = texts + (long)b * L;
    block_positions([&](int t) { return tok[t] != 0; }, L, pos, counts);
    const int len = (int)lens[b];
    const int c0 = blockIdx.y * EMB_CG;             // this workgroup's channel slice
    for (int idx = threadIdx.x; idx < EMB_CG * L; idx += 256) {
        const int cl = idx / L, l = idx - cl * L, c = c0 + cl;
        if (c >= C) break;
        float v = 0.f;
        if (l < len) v = scale * E[tok[l] * C + c] + pos_embed(pos[l], c, C, omega, tab, tab_rows);
        x[((long)b * C + c) * ld + l] = v;
    }
}

// ---- LayerNorm over the 256 channels of a channel-major tensor (model/blocks.py:88-107 eps 1e-12,
// model/modules.py:74 eps 1e-5); optional zeroing of columns >= lens[b].
// Reductions by WAVE SHUFFLES (round 3; BASELINE.json north_star): a wave = 8 columns x 8 channel phases (lane = 8 y + x), a lane sums the 32
// channels y + 8 i of its column, and the eight partial sums of a column — lanes x, 8 + x, ..., 56 + x — are added in the order y = 0..7 with
// eight __shfl: the SAME association as the LDS reduction it replaces and as conv_xres.hip's fused LayerNorm prologue (bit-identical), without
// LDS and without the three workgroup barriers.
__global__ __launch_bounds__(256) void layernorm_ct_kernel(const float* in, float* out, const float* gamma,
                                                           const float* beta, float eps, const int64_t* lens,
                                                           int T, int ld) {
    constexpr int C = 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tx = lane & 7, ty = lane >> 3;
    const int t = blockIdx.x * 32 + w * 8 + tx;
    const int b = blockIdx.y;
    const bool ok = t < T;
    float v[32];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = ty + 8 * i;
        v[i] = ok ? in[((long)b * C + c) * ld + t] : 0.f;
        sum += v[i];
    }
    float tot = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) tot += __shfl(sum, y * 8 + tx);
    const float mean = tot / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float d = v[i] - mean; sq = __fmaf_rn(d, d, sq); }   // explicit: conv_xres.hip's fused LayerNorm repeats exactly this sequence
    float var = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) var += __shfl(sq, y * 8 + tx);
    var = var / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    const bool keep = !(lens && (int64_t)t >= lens[b]);
    if (ok) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int c = ty + 8 * i;
            const float y = __fmaf_rn((v[i] - mean) * rstd, gamma[c], beta[c]);
            out[((long)b * C + c) * ld + t] = keep ? y : 0.f;
        }
    }
}

// ---- softmax over keys for transposed scores ST[z][j][i] (F.multi_head_attention_forward via
// model/blocks.py:303-312): column i = one query; keys j >= lens[b] are masked (-inf -> prob 0).
// Workgroup = 64 queries x 4 key slices (keys j = slice, slice + 4, ...); the slice maxima / sums meet in LDS.
__global__ __launch_bounds__(256) void softmax_cols_kernel(float* st, const int64_t* lens, int H, int L, int ld, long zs) {
    __shared__ float red[4][64];
    const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    const int z = blockIdx.y;
    const int len = min((int)lens[z / H], L);
    float* col = st + (long)z * zs + min(i, L - 1);
    float mx = -INFINITY;
    for (int j = sl; j < len; j += 4) mx = fmaxf(mx, col[(long)j * ld]);
    red[sl][il] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][il], red[1][il]), fmaxf(red[2][il], red[3][il]));
    __syncthreads();
    float sum = 0.f;
    for (int j = sl; j < len; j += 4) sum += expf(col[(long)j * ld] - mx);
    red[sl][il] = sum;
    __syncthreads();
    sum = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
    if (i >= L) return;
    for (int j = sl; j < len; j += 4) col[(long)j * ld] = expf(col[(long)j * ld] - mx) / sum;
    for (int j = len + sl; j < L; j += 4) col[(long)j * ld] = 0.f;
}

// ---- x[b][c][l] += vec[b][c] (speaker embedding broadcast, model/modules.py:349-352)
// lens (optional): columns l >= lens[b] do not exist for utterance b (a ragged text batch: its group's padded length) and stay 0
__global__ void add_rowvec_kernel(float* x, const float* vec, int C, int L, int ld, const int64_t* lens) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (l < L && !(lens && (int64_t)l >= lens[b])) x[((long)b * C + c) * ld + l] += vec[(long)b * C + c];
}

// ---- PitchPredictor/EnergyPredictor input: xs + alpha * PE[positions(xs[...,0] != 0)]
// (model/modules.py:548-549; the float-zero test on channel 0 is part of the semantics)
// grid (B, C / POS_CG): positions recomputed per workgroup, POS_CG channels each.
constexpr int POS_CG = 8;
__global__ __launch_bounds__(256) void pos_embed_add_kernel(const float* x, float* out, const float* alpha,
                                                            const float* omega, const float* tab, int tab_rows,
                                                            int C, int T, int ld, const int64_t* lens) {
    extern __shared__ int sh[];
    int* counts = sh;
    int* pos = sh + 256;
    const int b = blockIdx.x;
    const float* x0 = x + (long)b * C * ld;
    block_positions([&](int t) { return x0[t] != 0.f; }, T, pos, counts);
    const float al = alpha[0];
    const int c0 = blockIdx.y * POS_CG;
    const int valid = lens ? (int)min((int64_t)T, lens[b]) : T;       // x * nonpadding (FFTBlocks.forward, modules.py:95)
    for (int cl = 0; cl < POS_CG && c0 + cl < C; ++cl) {
        const int c = c0 + cl;
        for (int t = threadIdx.x; t < T; t += 256) {
            const long off = ((long)b * C + c) * ld + t;
            const float v = x[off] + al * pos_embed(pos[t], c, C, omega, tab, tab_rows);
            out[off] = t < valid ? v : 0.f;
        }
    }
}

// The same with the length regulator's gather in front (round 4): x[c][t] = mel2ph[t] > 0 ? src[c][mel2ph[t] - 1] : padv[c] is never written to a
// buffer — src = the pitch predictor's input projection over the phonemes (row stride ldl), padv = its bias (a padding frame projects to the bias)
__global__ __launch_bounds__(256) void pos_embed_add_lr_kernel(const float* src, int ldl, const int64_t* mel2ph, const float* padv, float* out,
                                                               const float* alpha, const float* omega, const float* tab, int tab_rows,
                                                               int C, int T) {
    extern __shared__ int sh[];
    int* counts = sh;
    int* pos = sh + 256;
    const int b = blockIdx.x;
    const int64_t* mp = mel2ph + (long)b * T;
    const float* s0 = src + (long)b * C * ldl;
    const float pad0 = padv[0];
    block_positions([&](int t) { const int64_t ph = mp[t]; return (ph > 0 ? s0[ph - 1] : pad0) != 0.f; }, T, pos, counts);
    const float al = alpha[0];
    const int c0 = blockIdx.y * POS_CG;
    for (int cl = 0; cl < POS_CG && c0 + cl < C; ++cl) {
        const int c = c0 + cl;
        const float* sc = s0 + (long)c * ldl;
        const float pc = padv[c];
        for (int t = threadIdx.x; t < T; t += 256) {
            const int64_t ph = mp[t];
            const float xv = ph > 0 ? sc[ph - 1] : pc;
            out[((long)b * C + c) * T + t] = xv + al * pos_embed(pos[t], c, C, omega, tab, tab_rows);
        }
    }
}

// ---- Linear(C -> O<=16) over channel-major input, output time-major [B][T][O]
// (duration/energy/cwt predictor heads, model/modules.py:505-506,554).  Workgroup = 64 positions x 4
// channel slices (one wave each, 8 independent loads in flight), partial sums meet in LDS.
__global__ __launch_bounds__(256) void chan_linear_kernel(const float* x, const float* W, const float* bias,
                                                          float* out, const int64_t* lens, int C, int T, int ld, int O) {
    extern __shared__ float wsh[];     // [O][C] weights, then [3][16][64] partial sums
    float* part = wsh + O * C;
    for (int i = threadIdx.x; i < O * C; i += 256) wsh[i] = W[i];
    __syncthreads();
    const int tl = threadIdx.x & 63, cs = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    const int b = blockIdx.y;
    const int tc = min(t, T - 1);
    const int cq = C / 4, c0 = cs * cq;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
    const float* xb = x + (long)b * C * ld + tc;
#pragma unroll 8
    for (int c = c0; c < c0 + cq; ++c) {
        const float xv = xb[(long)c * ld];
#pragma unroll
        for (int o = 0; o < 16; ++o)
            if (o < O) acc[o] = fmaf(xv, wsh[o * C + c], acc[o]);
    }
    if (cs > 0) {
#pragma unroll
        for (int o = 0; o < 16; ++o)
            if (o < O) part[((cs - 1) * 16 + o) * 64 + tl] = acc[o];
    }
    __syncthreads();
    if (cs == 0 && t < T) {
        const bool keep = !(lens && (int64_t)t >= lens[b]);
#pragma unroll
        for (int o = 0; o < 16; ++o)
            if (o < O) {
                const float v = ((acc[o] + part[(0 * 16 + o) * 64 + tl]) + part[(1 * 16 + o) * 64 + tl]) + part[(2 * 16 + o) * 64 + tl];
                out[((long)b * T + t) * O + o] = keep ? v + bias[o] : 0.f;
            }
    }
}

// ---- LayerNorm over 256 channels + Linear(256 -> O <= 16) in ONE launch: the head of the duration / energy / cwt predictors
// (the last conv block's LayerNorm, model/modules.py:497-499 / 546-548, feeding self.linear, :505-506 / :554).  A wave =
// 16 positions x 4 channel quarters (lane = 16 q + column): every lane holds its 64 channels in registers (64 loads in
// flight), and the mean, the variance and the O dot products are reduced across the four quarters with two __shfl_xor each
// — no LDS scratch, no barrier after the weights are staged.  ln_lens: positions >= ln_lens[b] enter the linear layer as
// zeros (the masked LayerNorm output); out_lens: those positions are written as 0.
constexpr int LNL_PAD = 65;      // a quarter's 64 weights + 1: the four quarters of one lane group hit different LDS banks
// EE (round 6, O == 1: the energy predictor's head): the energy embedding in the same launch — get_energy_embedding + the add of
// model/modules.py:318-328,358-363: the position's prediction (x control, or the teacher-forced target) is bucketized and every lane
// adds its 64 channels of energy_embedding[bucket] to the encoder output column, out1 = xin + E[idx].  energy_embed_kernel's expressions
// (the same bucketize, the same single add per element): the same bits; one launch and its dependent boundary less on the chain
// energy predictor -> out1 -> pitch predictor input.
struct EnergyEpi {
    const float* xin;       // [B][256][ld]: the variance adaptor's input x (speaker vector added)
    const float* e_target;  // teacher-forced energy [B][T] or null
    float e_control;
    const float* bins;      // [nbins] ascending
    int nbins;
    const float* E;         // energy_embedding.weight [nbins + 1][256]
    float* out1;            // [B][256][ld]
    int64_t* e_idx;         // [B][T]
    float* e_scaled;        // [B][T]: prediction x control (written when control != 1 and no target)
};
template <int O, bool EE = false>
__global__ __launch_bounds__(256) void ln_linear_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        const int64_t* ln_lens, const int64_t* out_lens, int T, int ld, const EnergyEpi ee = EnergyEpi{}) {
    constexpr int C = 256, CQ = 64;
    constexpr int NV = (O + 2) * C / 4;            // float4 of W rows, gamma, beta
    constexpr int PER = (NV + 255) / 256;
    __shared__ float lsm[(O + 2) * 4 * LNL_PAD];   // [O + 2][4][LNL_PAD]
    {   // all of a thread's loads before its first LDS write: one memory round trip
        float4 wv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i4 = min((int)threadIdx.x + 256 * u, NV - 1);
            const int r = i4 >> 6, c4 = i4 & 63;
            const float* src = r < O ? W + (long)r * C : (r == O ? gamma : beta);
            wv[u] = *reinterpret_cast<const float4*>(src + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i4 = threadIdx.x + 256 * u;
            if (i4 < NV) {
                const int r = i4 >> 6, c = 4 * (i4 & 63);
                float* d = lsm + (r * 4 + (c >> 6)) * LNL_PAD + (c & 63);
                d[0] = wv[u].x; d[1] = wv[u].y; d[2] = wv[u].z; d[3] = wv[u].w;
            }
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int t = blockIdx.x * 64 + w * 16 + col;
    const int b = blockIdx.y;
    const int tc = min(t, T - 1);
    const float* xb = x + ((long)b * C + q * CQ) * ld + tc;
    float v[CQ];
#pragma unroll
    for (int i = 0; i < CQ; ++i) v[i] = xb[(long)i * ld];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CQ; ++i) s += v[i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CQ; ++i) { const float d = v[i] - mean; sq = __fmaf_rn(d, d, sq); }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    const bool ln_keep = !(ln_lens && (int64_t)t >= ln_lens[b]);
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    const float* wq = lsm + q * LNL_PAD;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        float y = __fmaf_rn((v[i] - mean) * rstd, wq[(O * 4) * LNL_PAD + i], wq[((O + 1) * 4) * LNL_PAD + i]);
        if (!ln_keep) y = 0.f;
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = __fmaf_rn(y, wq[(o * 4) * LNL_PAD + i], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        acc[o] += __shfl_xor(acc[o], 16);
        acc[o] += __shfl_xor(acc[o], 32);
    }
    const bool keep = !(out_lens && (int64_t)t >= out_lens[b]);
    if (q == 0 && t < T) {
#pragma unroll
        for (int o = 0; o < O; ++o) out[((long)b * T + t) * O + o] = keep ? acc[o] + bias[o] : 0.f;
    }
    if constexpr (EE) {
        static_assert(!EE || O == 1, "energy head");
        if (t < T) {      // all four quarter lanes of the column hold the prediction
            const float pred = keep ? acc[0] + bias[0] : 0.f;
            float v;
            if (ee.e_target) v = ee.e_target[(long)b * T + t];
            else { v = pred * ee.e_control; if (q == 0 && ee.e_control != 1.0f) ee.e_scaled[(long)b * T + t] = v; }
            int lo = 0, hi = ee.nbins;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ee.bins[mid] >= v) hi = mid; else lo = mid + 1; }
            if (v != v) lo = ee.nbins;   // NaN sorts last
            if (q == 0) ee.e_idx[(long)b * T + t] = lo;
            const float* e = ee.E + (long)lo * C + q * CQ;
            const float* xi = ee.xin + ((long)b * C + q * CQ) * ld + t;
            float* o1 = ee.out1 + ((long)b * C + q * CQ) * ld + t;
            float xv[CQ], ev[CQ];
#pragma unroll
            for (int i = 0; i < CQ; ++i) { xv[i] = xi[(long)i * ld]; ev[i] = e[i]; }
#pragma unroll
            for (int i = 0; i < CQ; ++i) o1[(long)i * ld] = xv[i] + ev[i];
        }
    }
}

// ---- out[b][c][t] = mask(((p_0 + p_1) + ... + p_{n-1}) + bias[c] + res[b][c][t]): the reduction of the FFN linear's K-segment
// partial sums (part: [B][nseg][C][ld]) with the generic conv epilogue's order (accumulator, + bias, + residual, length mask)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nseg, const float* __restrict__ bias,
                                                              const float* res, const int64_t* lens, float* out, int C, int L, int ld) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (t >= L || c >= C) return;
    const float* p = part + ((long)b * nseg * C + c) * ld + t;
    float v = p[0];
    for (int s = 1; s < nseg; ++s) v += p[(long)s * C * ld];
    v += bias[c];
    v += res[((long)b * C + c) * ld + t];
    if (lens && (int64_t)t >= lens[b]) v = 0.f;
    out[((long)b * C + c) * ld + t] = v;
}

// ---- out[b][:] = row[:] for b < B (the cached step-embedding row to every utterance) and out = a + b element-wise
__global__ void broadcast_row_kernel(const float* __restrict__ row, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(long)blockIdx.y * n + i] = row[i];
}
__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// ---- tiny dense layer out[b][n] = act(sum_k in[b][k] * Wt[k][n] + bias[n]) + add[b][n].
// Workgroup = 64 output columns x KS K-slices (one wave per slice), DB batch rows per thread: Wt (up to
// 5 MB for the stacked per-layer projections) is streamed once per DB rows, 8 independent loads in
// flight per thread; in[b][k] is wave-uniform (scalar loads); the 4 partial sums meet in LDS.
constexpr int DB = 8;
template <int KS>
__global__ __launch_bounds__(64 * KS) void dense_small_kernel(const float* in, long in_bs, long in_ks, const float* Wt,
                                                              const float* bias, const float* add, float* out,
                                                              int B, int K, int N, int act) {
    __shared__ float part[KS - 1][DB][64];
    const int nl = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl;
    const int nc = min(n, N - 1);
    const int b0 = blockIdx.y * DB;
    const int kq = (K + KS - 1) / KS;
    const int k0 = ks * kq, k1 = min(K, k0 + kq);
    float acc[DB];
#pragma unroll
    for (int j = 0; j < DB; ++j) acc[j] = 0.f;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
        const float wv = Wt[(long)k * N + nc];
#pragma unroll
        for (int j = 0; j < DB; ++j) {
            const int b = min(b0 + j, B - 1);
            acc[j] = fmaf(in[(long)b * in_bs + (long)k * in_ks], wv, acc[j]);
        }
    }
    if (ks > 0) {
#pragma unroll
        for (int j = 0; j < DB; ++j) part[ks - 1][j][nl] = acc[j];
    }
    __syncthreads();
    if (ks == 0 && n < N) {
#pragma unroll
        for (int j = 0; j < DB; ++j) {
            const int b = b0 + j;
            if (b >= B) break;
            float v = acc[j];
#pragma unroll
            for (int q = 0; q < KS - 1; ++q) v += part[q][j][nl];
            if (bias) v += bias[n];
            if (act == DENSE_RELU) v = v > 0.f ? v : 0.f;
            else if (act == DENSE_MISH) {
                const float sp = v > 20.f ? v : log1pf(expf(v));
                v = v * tanhf(sp);
            }
            if (add) v += add[(long)b * N + n];
            out[(long)b * N + n] = v;
        }
    }
}

// ---- cwt_stats_layers (model/modules.py:212-215,279): Linear(K0, N0) -> ReLU -> Linear(N0, N1) -> ReLU -> Linear(N1, N2) on one input row per
// utterance, as ONE launch (round 6).  As three dense_small_kernel<4> launches (8 / 4 / 1 workgroups, every thread a serial walk over its
// K slice behind scalar loads of the input) the MLP took 50-80 us per layer beside the frame-level convs and ended exactly where the pitch
// chain joins it.  Workgroup = utterance, 512 threads = dense_small_kernel<4>'s four K slices x 128 output columns; the input row and the
// hidden rows live in LDS; per output the same fmaf chain per slice and the same order of the slice sums, bias, ReLU => the same bits.
constexpr int SM_COLS = 128;
__global__ __launch_bounds__(4 * SM_COLS) void stats_mlp_kernel(const float* __restrict__ in, long in_bs, long in_ks, const float* __restrict__ W0, const float* __restrict__ b0,
                                                                const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                                                                float* __restrict__ out, int K0, int N0, int N1, int N2) {
    extern __shared__ float sm_lds[];          // x0[K0] | h1[N0] | h2[N1] | part[3][SM_COLS]
    float* x0 = sm_lds;
    float* h1 = x0 + K0;
    float* h2 = h1 + N0;
    float* part = h2 + N1;
    const int nl = threadIdx.x & (SM_COLS - 1), ks = threadIdx.x / SM_COLS;
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < K0; k += blockDim.x) x0[k] = in[(long)b * in_bs + (long)k * in_ks];
    __syncthreads();
    auto layer = [&](const float* x, int K, const float* __restrict__ Wt, const float* __restrict__ bias, int N, bool relu, float* y) {
        const int kq = (K + 3) / 4;
        const int k0 = ks * kq, k1 = min(K, k0 + kq);
        for (int n0 = 0; n0 < N; n0 += SM_COLS) {
            const int n = n0 + nl, nc = min(n, N - 1);
            float acc = 0.f;
#pragma unroll 8
            for (int k = k0; k < k1; ++k) acc = fmaf(x[k], Wt[(long)k * N + nc], acc);
            if (ks > 0) part[(ks - 1) * SM_COLS + nl] = acc;
            __syncthreads();
            if (ks == 0 && n < N) {
                float v = acc;
#pragma unroll
                for (int q = 0; q < 3; ++q) v += part[q * SM_COLS + nl];
                if (bias) v += bias[n];
                if (relu) v = v > 0.f ? v : 0.f;
                y[n] = v;
            }
            __syncthreads();
        }
    };
    layer(x0, K0, W0, b0, N0, true, h1);
    layer(h1, N0, W1, b1, N1, true, h2);
    layer(h2, N1, W2, b2, N2, false, out + (long)b * N2);
}

// ---- energy bucketize + embedding add (model/modules.py:319-329,358-363); torch.bucketize
// right=False = first i with bins[i] >= v
constexpr int ENE_CG = 32;
__global__ void energy_embed_kernel(const float* x, const float* e_pred, float* e_scaled, const float* e_target,
                                    float e_control, const float* bins, int nbins, const float* E, float* out1,
                                    int64_t* e_idx, int C, int L, int ld) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    // get_energy_embedding (model/modules.py:318-328): the target is bucketized when given, else
    // prediction * control (which is also what is returned as the prediction)
    float v;
    if (e_target) v = e_target[(long)b * L + l];
    else { v = e_pred[(long)b * L + l] * e_control; if (blockIdx.z == 0 && e_control != 1.0f) e_scaled[(long)b * L + l] = v; }
    int lo = 0, hi = nbins;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (bins[mid] >= v) hi = mid; else lo = mid + 1; }
    if (v != v) lo = nbins;   // NaN sorts last
    const int c0 = blockIdx.z * ENE_CG, c1 = min(C, c0 + ENE_CG);     // channel slice of this workgroup
    if (c0 == 0) e_idx[(long)b * L + l] = lo;
    const float* e = E + (long)lo * C;
    for (int c = c0; c < c1; ++c) {
        const long off = ((long)b * C + c) * ld + l;
        out1[off] = x[off] + e[c];
    }
}

// ---- d = clamp(round(exp(log_d) - 1) * d_control, 0) (half-to-even), cumulative sums, mel_len
// (model/modules.py:369-372, utils/tools.py:788-791)
__global__ void durations_kernel(const float* logd, float d_control, float* d_rounded, int* cum, int64_t* mel_len,
                                 int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int run = 0;
    for (int l = 0; l < L; ++l) {
        float d = rintf(expf(logd[(long)b * L + l]) - 1.0f) * d_control;
        d = d > 0.f ? d : 0.f;
        d_rounded[(long)b * L + l] = d;
        run += (int)d;            // LengthRegulator.expand: int(expand_size)
        cum[(long)b * L + l] = run;
    }
    mel_len[b] = run;
}

// The same, one WAVE per utterance: the durations of 64 phonemes at a time in parallel and an in-wave inclusive scan of their
// integer parts (__shfl_up; integer adds, so the order does not matter) instead of one thread walking the utterance.
__global__ __launch_bounds__(64) void durations_wave_kernel(const float* logd, float d_control, float* d_rounded, int* cum,
                                                            int64_t* mel_len, int L) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int carry = 0;
    for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        float d = 0.f;
        if (l < L) {
            d = rintf(expf(logd[(long)b * L + l]) - 1.0f) * d_control;
            d = d > 0.f ? d : 0.f;
            d_rounded[(long)b * L + l] = d;
        }
        int v = (int)d;            // LengthRegulator.expand: int(expand_size)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(v, off);
            if (lane >= off) v += n;
        }
        v += carry;
        if (l < L) cum[(long)b * L + l] = v;
        carry = __shfl(v, 63);
    }
    if (lane == 0) mel_len[b] = carry;
}

// ---- cumulative sums of already-rounded durations (LengthRegulator.expand: max(int(d), 0))
__global__ void cumsum_durations_kernel(const float* dur, int* cum, int64_t* mel_len, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int run = 0;
    for (int l = 0; l < L; ++l) {
        const int d = (int)dur[(long)b * L + l];
        run += d > 0 ? d : 0;
        cum[(long)b * L + l] = run;
    }
    mel_len[b] = run;
}

// ---- mel2ph[b][t] = 1 + #{l : cum[l] <= t} for t < cum[L-1], else 0 (utils/tools.py:793-797)
__global__ void mel2ph_kernel(const int* cum, int64_t* mel2ph, int L, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T) return;
    const int* c = cum + (long)b * L;
    int64_t r = 0;
    if (t < c[L - 1]) {
        int lo = 0, hi = L;           // first l with cum[l] > t
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[mid] > t) hi = mid; else lo = mid + 1; }
        r = lo + 1;
    }
    mel2ph[(long)b * T + t] = r;
}

// ---- length regulator gather (model/modules.py:421-448): frame t copies phoneme mel2ph-1, pad = 0
// padv (optional, [C]): the value of a padding frame — the bias of a k = 1 projection that was applied BEFORE the gather (W 0 + b = b)
__global__ void length_regulate_kernel(const float* out1, const int64_t* mel2ph, float* xlr, int C, int ldl, int T, const float* padv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int64_t ph = mel2ph[(long)b * T + t];
    xlr[((long)b * C + c) * T + t] = ph > 0 ? out1[((long)b * C + c) * ldl + (ph - 1)] : (padv ? padv[c] : 0.f);
}

// ---- inverse CWT -> f0 -> coarse pitch bucket (model/modules.py:274-300; utils/pitch_tools.py
// 244-250 inverse_cwt_torch, 261-279 cwt2f0/_norm, 38-47 norm_f0, 64-78 denorm_f0, 26-35 f0_to_coarse)
// spec: [B][T][O] with the 10 wavelet scales first (predicted cwt, or the teacher-forced target with O = 10);
// mean / stdv: [B] with strides (the stats head's [B,2] or separate target vectors); uv: predicted logit
// column uv_logit[(b*T+t)*uv_ld] > 0 or target bytes uv_mask[b*T+t] != 0.
__global__ __launch_bounds__(256) void pitch_index_kernel(const float* cwt, int O, const float* mean_p, const float* std_p,
                                                          int stat_ld, float std_scale, const float* uv_logit, int uv_ld,
                                                          const uint8_t* uv_mask, float eps, float* r_ws,
                                                          int64_t* p_idx, float* f0_denorm, int T) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* cw = cwt + (long)b * T * O;
    float* r = r_ws + (long)b * T;
    double s = 0.0;
    for (int t = tid; t < T; t += 256) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) acc += cw[(long)t * O + j] * powf((float)j + 1.0f + 2.5f, -2.5f);
        r[t] = acc;
        s += (double)acc;
    }
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    const float mean_r = (float)(red[0] / (double)T);
    __syncthreads();
    double q = 0.0;
    for (int t = tid; t < T; t += 256) { const double d = (double)r[t] - (double)mean_r; q += d * d; }
    red[tid] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    const float std_r = (float)sqrt(red[0] / (double)(T - 1));       // torch.std: unbiased
    const float mean = mean_p[b * stat_ld];
    const float stdv = std_p[b * stat_ld] * std_scale;
    const float mel_min = (float)(1127.0 * log(1.0 + 50.0 / 700.0));
    const float mel_rng = (float)(1127.0 * log(1.0 + 1100.0 / 700.0) - 1127.0 * log(1.0 + 50.0 / 700.0));
    for (int t = tid; t < T; t += 256) {
        const float rn = (r[t] - mean_r) / std_r;
        const float f0 = expf(rn * stdv + mean);
        const float f0n = log2f(f0 + eps);
        float f0d = exp2f(f0n);
        if (uv_logit && uv_logit[((long)b * T + t) * uv_ld] > 0.f) f0d = 0.f;
        if (uv_mask && uv_mask[(long)b * T + t]) f0d = 0.f;
        f0_denorm[(long)b * T + t] = f0d;
        float mel = 1127.0f * logf(1.0f + f0d / 700.0f);
        if (mel > 0.f) mel = (mel - mel_min) * 254.0f / mel_rng + 1.0f;
        if (mel <= 1.0f) mel = 1.0f;
        if (mel > 255.0f) mel = 255.0f;
        p_idx[(long)b * T + t] = (int64_t)(mel + 0.5f);
    }
}

// ---- out[b][c][t] = x[b][c][t] + E[idx[b][t]][c] (pitch embedding add, model/modules.py:300,395)
// cond[b][c][t] = LR(out1)[b][c][t] + E[idx[b][t]][c] with the length regulator's gather done here (round 4): frame t copies phoneme
// mel2ph[t] - 1 of out1 (row stride ldl), a padding frame 0 — the value length_regulate_kernel would have written to a buffer
__global__ void lr_gather_add_kernel(const float* out1, const int64_t* mel2ph, int ldl, const int64_t* idx, const float* E, float* out, int C, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int64_t ph = mel2ph[(long)b * T + t];
    const float xv = ph > 0 ? out1[((long)b * C + c) * ldl + (ph - 1)] : 0.f;
    out[((long)b * C + c) * T + t] = xv + E[idx[(long)b * T + t] * C + c];
}
__global__ void gather_add_kernel(const float* x, const int64_t* idx, const float* E, float* out, int C, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long off = ((long)b * C + c) * T + t;
    out[off] = x[off] + E[idx[(long)b * T + t] * C + c];
}

// ---- hin[b][m][t] = scale * x[b][t][m]  (c_in pre-scaling + [B,1,T,80] -> [B,80,T],
// karras_diffusion.py:405, tts_net.py:31)
__global__ void mel_prep_kernel(const float* x, const float* scale_b, float scale, float* hin, int T, int M) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const float s = scale_b ? scale_b[b] : scale;
    hin[((long)b * M + m) * T + t] = s * x[((long)b * T + t) * M + m];
}

// ---- out[b][t][m] = c_out*F[b][m][t] + c_skip*xold[b][t][m] (+ noise*nstd)
// (karras_diffusion.py:406 and the re-noising of stochastic_iterative_sampler :852)
// flag (pinned host word, may be null): set to 2 when a non-finite value is written — 16-bit operands that left the fp16 range, non-finite
// weights / inputs — unless an earlier error is still pending there (cmtts_poll_error)
__global__ void mel_post_kernel(const float* F, const float* xold, const float* noise, float c_out, float c_skip,
                                float nstd, float* out, int T, int M, unsigned* flag) {
    const int m = threadIdx.x;
    const int t = blockIdx.x * blockDim.y + threadIdx.y;
    const int b = blockIdx.y;
    if (t >= T || m >= M) return;
    const long o = ((long)b * T + t) * M + m;
    float v = c_out * F[((long)b * M + m) * T + t];
    if (xold) v += c_skip * xold[o];
    if (noise) v += noise[o] * nstd * 0.85f;     // randn_like(x) * sqrt(next_t^2 - t_min^2) * 0.85
    out[o] = v;
    if (flag && !(fabsf(v) <= 3.402823466e38f) && *(volatile unsigned*)flag == 0u) *(volatile unsigned*)flag = 2u;
}

// ---- DiffusionEmbedding (model/blocks.py:633-640): [sin(t*w) | cos(t*w)]
__global__ void diff_embed_kernel(const float* t, const float* omega, float* emb, int C) {
    const int c = threadIdx.x, b = blockIdx.x;
    if (c >= C) return;
    const int half = C >> 1;
    const int j = c < half ? c : c - half;
    const float arg = t[b] * omega[j];
    emb[(long)b * C + c] = c < half ? sin_exact(arg) : cos_exact(arg);
}

// ---- HiFi-GAN tail: leaky_relu(x/pre_div, slope) -> Conv1d(C->1, KW) -> tanh (hifigan/models.py:161-163)
// A thread produces POST_V consecutive samples: the division by 3 and the LeakyReLU of an input value are evaluated once
// for the POST_V + KW - 1 values a thread touches instead of once per (sample, tap) — 10 instead of 28 at KW = 7 — and every
// sample still accumulates over (channel, tap) in ascending order: bitwise the values of the one-sample-per-thread form.
constexpr int POST_V = 4;
constexpr int POST_KW_MAX = 7;
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float pre_div, float slope, float* __restrict__ wav, int C, int T, int ld, int KW) {
    extern __shared__ float wsh[];
    for (int i = threadIdx.x; i < C * KW; i += 256) wsh[i] = w[i];
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * POST_V;
    const int b = blockIdx.y;
    if (t0 >= T) return;
    const int pad = KW / 2;
    float acc[POST_V];
#pragma unroll
    for (int v = 0; v < POST_V; ++v) acc[v] = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* xr = x + ((long)b * C + c) * ld;
        float xv[POST_V + POST_KW_MAX - 1];
#pragma unroll
        for (int q = 0; q < POST_V + POST_KW_MAX - 1; ++q) {
            const int tt = t0 + q - pad;
            float v = (q < POST_V + KW - 1 && tt >= 0 && tt < T) ? xr[tt] : 0.f;
            if (pre_div != 1.0f) v = v / pre_div;
            xv[q] = v > 0.f ? v : v * slope;
        }
#pragma unroll
        for (int k = 0; k < POST_KW_MAX; ++k) {
            if (k < KW) {
                const float wk = wsh[c * KW + k];
#pragma unroll
                for (int v = 0; v < POST_V; ++v) acc[v] = fmaf(wk, xv[v + k], acc[v]);
            }
        }
    }
    const float bs = bias[0];
#pragma unroll
    for (int v = 0; v < POST_V; ++v)
        if (t0 + v < T) wav[(long)b * T + t0 + v] = tanhf(acc[v] + bs);
}

// The same arithmetic with 16-byte loads (rows 16-byte aligned, KW <= 7): per channel a thread fetches the float4 that holds its four samples
// and the two neighbouring ones — 3 loads instead of 10 scalars whose lanes sit 16 bytes apart (every scalar instruction touched the same 1 KB) —
// and four channels' loads are in flight together.  The generator's last stage is 537 MB at 32 x 512 frames: 0.33 ms -> 0.2 ms in every precision mode.
__global__ __launch_bounds__(256) void conv_post_v4_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float pre_div, float slope, float* __restrict__ wav, int C, int T, int ld, int KW) {
    extern __shared__ float wsh[];
    for (int i = threadIdx.x; i < C * KW; i += 256) wsh[i] = w[i];
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * POST_V;
    const int b = blockIdx.y;
    if (t0 >= T) return;
    const int pad = KW / 2;
    float acc[POST_V];
#pragma unroll
    for (int v = 0; v < POST_V; ++v) acc[v] = 0.f;
    const int tl = max(t0 - 4, 0), tr = min(t0 + 4, ((T + 3) & ~3) - 4);       // clamped (aligned) addresses; the values are masked below
    constexpr int CU = 4;
    for (int c0 = 0; c0 < C; c0 += CU) {
        float4 Lq[CU], Mq[CU], Rq[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const float* xr = x + ((long)b * C + min(c0 + u, C - 1)) * ld;
            Lq[u] = *reinterpret_cast<const float4*>(xr + tl);
            Mq[u] = *reinterpret_cast<const float4*>(xr + t0);
            Rq[u] = *reinterpret_cast<const float4*>(xr + tr);
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            if (c0 + u >= C) break;
            const float raw[12] = {Lq[u].x, Lq[u].y, Lq[u].z, Lq[u].w, Mq[u].x, Mq[u].y, Mq[u].z, Mq[u].w, Rq[u].x, Rq[u].y, Rq[u].z, Rq[u].w};
            float xv[POST_V + POST_KW_MAX - 1];
#pragma unroll
            for (int q = 0; q < POST_V + POST_KW_MAX - 1; ++q) {
                const int tt = t0 + q - pad;
                const int ri = q - pad + 4;                 // index into raw: tt = t0 - 4 + ri
                float v = (q < POST_V + KW - 1 && tt >= 0 && tt < T && ri >= 0 && ri < 12) ? raw[ri < 0 ? 0 : (ri > 11 ? 11 : ri)] : 0.f;
                if (pre_div != 1.0f) v = v / pre_div;
                xv[q] = v > 0.f ? v : v * slope;
            }
#pragma unroll
            for (int k = 0; k < POST_KW_MAX; ++k) {
                if (k < KW) {
                    const float wk = wsh[(c0 + u) * KW + k];
#pragma unroll
                    for (int v = 0; v < POST_V; ++v) acc[v] = fmaf(wk, xv[v + k], acc[v]);
                }
            }
        }
    }
    const float bs = bias[0];
#pragma unroll
    for (int v = 0; v < POST_V; ++v)
        if (t0 + v < T) wav[(long)b * T + t0 + v] = tanhf(acc[v] + bs);
}

// ---- (wav * 32768).astype(int16): truncation toward zero through int32, low 16 bits kept
// (utils/model.py:195-198; +1.0 wraps to -32768 exactly as numpy's cast does)
__global__ void wav_to_int16_kernel(const float* wav, int16_t* pcm, long n, float max_wav) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pcm[i] = (int16_t)(int)(wav[i] * max_wav);
}

// ---- [B][R][Cn] -> [B][Cn][R] through a 32x33 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, int R, int Cn) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cn) ? in[((long)b * R + r) * Cn + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cn && r < R) out[((long)b * Cn + c) * R + r] = tile[tx][i];
    }
}

__global__ void scale_kernel(const float* in, float* out, long n, float sc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * sc;
}

__global__ void fill_float_kernel(float* p, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void fill_lens_kernel(int64_t* lens, int64_t v, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) lens[i] = v;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace

void k_embed_tokens(const int64_t* texts, const int64_t* lens, const float* E, const float* omega, const float* tab,
                    int tab_rows, float* x, int B, int L, int ld, int C, float scale, hipStream_t s) {
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(B, cdiv(C, EMB_CG)), dim3(256), (256 + L) * sizeof(int), s, texts, lens, E, omega,
                       tab, tab_rows, x, L, ld, C, scale);
}
void k_layernorm_ct(const float* in, float* out, const float* gamma, const float* beta, float eps,
                    const int64_t* lens, int B, int T, int ld, hipStream_t s) {
    hipLaunchKernelGGL(layernorm_ct_kernel, dim3(cdiv(T, 32), B), dim3(256), 0, s, in, out, gamma, beta, eps, lens, T, ld);
}
void k_softmax_cols(float* st, const int64_t* lens, int nz, int H, int L, int ld, long zs, hipStream_t s) {
    hipLaunchKernelGGL(softmax_cols_kernel, dim3(cdiv(L, 64), nz), dim3(256), 0, s, st, lens, H, L, ld, zs);
}
void k_add_rowvec(float* x, const float* vec, int B, int C, int L, int ld, hipStream_t s, const int64_t* lens) {
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(cdiv(L, 64), C, B), dim3(64), 0, s, x, vec, C, L, ld, lens);
}
void k_pos_embed_add(const float* x, float* out, const float* alpha, const float* omega, const float* tab,
                     int tab_rows, int B, int C, int T, int ld, hipStream_t s, const int64_t* lens) {
    hipLaunchKernelGGL(pos_embed_add_kernel, dim3(B, cdiv(C, POS_CG)), dim3(256), (256 + T) * sizeof(int), s, x, out, alpha, omega, tab,
                       tab_rows, C, T, ld, lens);
}
void k_pos_embed_add_lr(const float* src, int ldl, const int64_t* mel2ph, const float* padv, float* out, const float* alpha, const float* omega,
                        const float* tab, int tab_rows, int B, int C, int T, hipStream_t s) {
    hipLaunchKernelGGL(pos_embed_add_lr_kernel, dim3(B, cdiv(C, POS_CG)), dim3(256), (256 + T) * sizeof(int), s, src, ldl, mel2ph, padv, out, alpha,
                       omega, tab, tab_rows, C, T);
}
void k_chan_linear(const float* x, const float* W, const float* bias, float* out, const int64_t* lens, int B,
                   int C, int T, int ld, int O, hipStream_t s) {
    hipLaunchKernelGGL(chan_linear_kernel, dim3(cdiv(T, 64), B), dim3(256), (size_t)(O * C + 3 * 16 * 64) * sizeof(float), s,
                       x, W, bias, out, lens, C, T, ld, O);
}
void k_dense_small(const float* in, long in_bs, long in_ks, const float* Wt, const float* bias, const float* add,
                   float* out, int B, int K, int N, int act, hipStream_t s) {
    // few output columns and a long K (the step-embedding MLP: 1024 -> 256) would leave 16 workgroups looping
    // 256 times: cut K into 16 slices there, 4 otherwise
    if (K >= 512 && cdiv(N, 64) * cdiv(B, DB) < 128)
        hipLaunchKernelGGL(dense_small_kernel<16>, dim3(cdiv(N, 64), cdiv(B, DB)), dim3(1024), 0, s, in, in_bs, in_ks, Wt,
                           bias, add, out, B, K, N, act);
    else
        hipLaunchKernelGGL(dense_small_kernel<4>, dim3(cdiv(N, 64), cdiv(B, DB)), dim3(256), 0, s, in, in_bs, in_ks, Wt,
                           bias, add, out, B, K, N, act);
}
bool k_stats_mlp(const float* in, long in_bs, long in_ks, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                 const float* b2, float* out, int B, int K0, int N0, int N1, int N2, hipStream_t s) {
    const size_t lds = (size_t)(K0 + N0 + N1 + 3 * SM_COLS) * sizeof(float);
    if (lds > 48 * 1024) return false;
    hipLaunchKernelGGL(stats_mlp_kernel, dim3(B), dim3(4 * SM_COLS), lds, s, in, in_bs, in_ks, W0, b0, W1, b1, W2, b2, out, K0, N0, N1, N2);
    return true;
}
void k_energy_embed(const float* x, const float* e_pred, float* e_scaled, const float* e_target, float e_control,
                    const float* bins, int nbins, const float* E, float* out1, int64_t* e_idx, int B, int C, int L, int ld,
                    hipStream_t s) {
    // every channel slice reads the unscaled prediction; the scaled copy (returned as the prediction when a
    // control is set) is written to a second buffer and copied back afterwards
    hipLaunchKernelGGL(energy_embed_kernel, dim3(cdiv(L, 64), B, cdiv(C, ENE_CG)), dim3(64), 0, s, x, e_pred, e_scaled, e_target,
                       e_control, bins, nbins, E, out1, e_idx, C, L, ld);
}
void k_durations(const float* logd, float d_control, float* d_rounded, int* cum, int64_t* mel_len, int B, int L,
                 hipStream_t s) {
    hipLaunchKernelGGL(durations_wave_kernel, dim3(B), dim3(64), 0, s, logd, d_control, d_rounded, cum, mel_len, L);
}
void k_durations_serial(const float* logd, float d_control, float* d_rounded, int* cum, int64_t* mel_len, int B, int L,
                        hipStream_t s) {
    hipLaunchKernelGGL(durations_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, logd, d_control, d_rounded, cum, mel_len, B, L);
}
void k_broadcast_row(const float* row, float* out, int B, int n, hipStream_t s) {
    hipLaunchKernelGGL(broadcast_row_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, s, row, out, n);
}
void k_add_rows(const float* a, const float* b, float* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, out, n);
}
void k_reduce_partials(const float* part, int nseg, const float* bias, const float* res, const int64_t* lens, float* out, int B, int C,
                       int L, int ld, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(L, 64), cdiv(C, 4), B), dim3(256), 0, s, part, nseg, bias, res, lens, out, C, L, ld);
}
// the energy predictor's head with the energy embedding in the same launch (EE instance)
void k_ln_linear_energy(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias, float* out,
                        const int64_t* ln_lens, const int64_t* out_lens, int B, int T, int ld, const float* xin, const float* e_target, float e_control,
                        const float* bins, int nbins, const float* E, float* out1, int64_t* e_idx, float* e_scaled, hipStream_t s) {
    const EnergyEpi ee{xin, e_target, e_control, bins, nbins, E, out1, e_idx, e_scaled};
    hipLaunchKernelGGL((ln_linear_kernel<1, true>), dim3(cdiv(T, 64), B), dim3(256), 0, s, x, gamma, beta, eps, W, bias, out, ln_lens, out_lens, T, ld, ee);
}
bool k_ln_linear(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias, float* out,
                 const int64_t* ln_lens, const int64_t* out_lens, int B, int T, int ld, int O, hipStream_t s) {
    const dim3 grid(cdiv(T, 64), B), block(256);
    switch (O) {     // the heads of this model family: 1 (duration, energy), 10 / 11 (cwt spectrogram without / with the uv logit)
        case 1: hipLaunchKernelGGL(ln_linear_kernel<1>, grid, block, 0, s, x, gamma, beta, eps, W, bias, out, ln_lens, out_lens, T, ld); return true;
        case 10: hipLaunchKernelGGL(ln_linear_kernel<10>, grid, block, 0, s, x, gamma, beta, eps, W, bias, out, ln_lens, out_lens, T, ld); return true;
        case 11: hipLaunchKernelGGL(ln_linear_kernel<11>, grid, block, 0, s, x, gamma, beta, eps, W, bias, out, ln_lens, out_lens, T, ld); return true;
        default: return false;
    }
}
void k_cumsum_durations(const float* dur, int* cum, int64_t* mel_len, int B, int L, hipStream_t s) {
    hipLaunchKernelGGL(cumsum_durations_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, dur, cum, mel_len, B, L);
}
void k_mel2ph(const int* cum, int64_t* mel2ph, int B, int L, int T, hipStream_t s) {
    hipLaunchKernelGGL(mel2ph_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, s, cum, mel2ph, L, T);
}
void k_length_regulate(const float* out1, const int64_t* mel2ph, float* xlr, int B, int C, int ldl, int T,
                       hipStream_t s, const float* padv) {
    hipLaunchKernelGGL(length_regulate_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, s, out1, mel2ph, xlr, C, ldl, T, padv);
}
void k_pitch_index(const float* cwt, int O, const float* mean_p, const float* std_p, int stat_ld, float std_scale,
                   const float* uv_logit, int uv_ld, const uint8_t* uv_mask, float eps, float* r_ws, int64_t* p_idx,
                   float* f0_denorm, int B, int T, hipStream_t s) {
    hipLaunchKernelGGL(pitch_index_kernel, dim3(B), dim3(256), 0, s, cwt, O, mean_p, std_p, stat_ld, std_scale, uv_logit,
                       uv_ld, uv_mask, eps, r_ws, p_idx, f0_denorm, T);
}
void k_gather_add(const float* x, const int64_t* idx, const float* E, float* out, int B, int C, int T, hipStream_t s) {
    hipLaunchKernelGGL(gather_add_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, s, x, idx, E, out, C, T);
}
void k_lr_gather_add(const float* out1, const int64_t* mel2ph, int ldl, const int64_t* idx, const float* E, float* out, int B, int C, int T,
                     hipStream_t s) {
    hipLaunchKernelGGL(lr_gather_add_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, s, out1, mel2ph, ldl, idx, E, out, C, T);
}
void k_mel_prep(const float* x, const float* scale_b, float scale, float* hin, int B, int T, int M, hipStream_t s) {
    hipLaunchKernelGGL(mel_prep_kernel, dim3(cdiv(T, 256), M, B), dim3(256), 0, s, x, scale_b, scale, hin, T, M);
}
void k_mel_post(const float* F, const float* xold, const float* noise, float c_out, float c_skip, float nstd,
                float* out, int B, int T, int M, unsigned* flag, hipStream_t s) {
    // blockDim = (M rounded to 16 | 3 rows): 80 mels -> (80, 3) = 240 threads
    const int ty = 256 / M > 0 ? 256 / M : 1;
    hipLaunchKernelGGL(mel_post_kernel, dim3(cdiv(T, ty), B), dim3(M, ty), 0, s, F, xold, noise, c_out, c_skip, nstd,
                       out, T, M, flag);
}
void k_diff_embed(const float* t, const float* omega, float* emb, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(diff_embed_kernel, dim3(B), dim3(C), 0, s, t, omega, emb, C);
}
int g_post_v4 = 1;               // conv_post with 16-byte loads (same bits); 0 = the scalar-load kernel (internal switch "post_v4")
void k_conv_post(const float* x, const float* w, const float* bias, float pre_div, float slope, float* wav, int B,
                 int C, int T, int ld, int KW, hipStream_t s) {
    // 16-byte loads when every row is 16-byte aligned and holds whole float4s up to the last sample's (the workspace rows do: ld >= T rounded up)
    if (g_post_v4 && KW <= POST_KW_MAX && KW / 2 <= 4 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0 && ld >= ((T + 3) & ~3) && T >= 4) {
        hipLaunchKernelGGL(conv_post_v4_kernel, dim3(cdiv(T, 256 * POST_V), B), dim3(256), (size_t)C * KW * sizeof(float), s, x, w,
                           bias, pre_div, slope, wav, C, T, ld, KW);
        return;
    }
    hipLaunchKernelGGL(conv_post_kernel, dim3(cdiv(T, 256 * POST_V), B), dim3(256), (size_t)C * KW * sizeof(float), s, x, w,
                       bias, pre_div, slope, wav, C, T, ld, KW);
}
void k_wav_to_int16(const float* wav, int16_t* pcm, long n, float max_wav, hipStream_t s) {
    hipLaunchKernelGGL(wav_to_int16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, wav, pcm, n, max_wav);
}
__global__ void length_mask_kernel(const int64_t* __restrict__ lens, uint8_t* __restrict__ mask, int B, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * W) return;
    const int b = (int)(i / W), t = (int)(i - (long)b * W);
    mask[i] = t >= lens[b] ? 1 : 0;
}
void k_length_mask(const int64_t* lens, uint8_t* mask, int B, int W, hipStream_t s) {
    hipLaunchKernelGGL(length_mask_kernel, dim3(cdiv((long)B * W, 256)), dim3(256), 0, s, lens, mask, B, W);
}
void k_transpose(const float* in, float* out, int B, int R, int Cn, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cn, 32), cdiv(R, 32), B), dim3(256), 0, s, in, out, R, Cn);
}
// ---- out[b][c] = table[clamp(idx[b])][c]: speaker_emb = nn.Embedding(n_speaker, hidden)(speakers), model/cmtts.py:77-78
__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out, int C, int n_rows) {
    const int b = blockIdx.x;
    long r = idx[b];
    r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(long)b * C + c] = table[r * C + c];
}
void k_gather_rows(const float* table, const int64_t* idx, float* out, int B, int C, int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, s, table, idx, out, C, n_rows);
}
// ---- dst[r][0..width) = src[r][0..width) for `rows` rows with different row strides (re-pitching a channel-major
// tensor between its padded workspace form and the caller's dense form).  hipMemcpy2DAsync splits such a copy into dozens
// of blit kernels (33 per call at B = 32: 0.15 ms of a 13.5 ms step); this is one launch.
__global__ void copy_rows_kernel(float* __restrict__ dst, int dst_ld, const float* __restrict__ src, int src_ld, int width, long rows) {
    const int t = blockIdx.y * blockDim.x + threadIdx.x;
    const long r = (long)blockIdx.x * 4 + threadIdx.y;
    if (t < width && r < rows) dst[r * dst_ld + t] = src[r * src_ld + t];
}
void k_copy_rows(float* dst, int dst_ld, const float* src, int src_ld, int width, long rows, hipStream_t s) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(cdiv(rows, 4), cdiv(width, 64)), dim3(64, 4), 0, s, dst, dst_ld, src, src_ld, width, rows);
}
void k_scale(const float* in, float* out, long n, float sc, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, in, out, n, sc);
}
void k_fill_float(float* p, float v, int n, hipStream_t s) {
    hipLaunchKernelGGL(fill_float_kernel, dim3(cdiv(n, 64)), dim3(64), 0, s, p, v, n);
}
void k_fill_lens(int64_t* lens, int64_t v, int B, hipStream_t s) {
    hipLaunchKernelGGL(fill_lens_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, lens, v, B);
}
