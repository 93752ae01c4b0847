// HiFi-GAN ResBlock pair for the narrow stages (C = 64 and C = 32), fp32 — one launch per
//   xt = conv1(leaky_relu(x, 0.1))  [k taps, dilation d];   y = conv2(leaky_relu(xt, 0.1)) + x  [k taps, dilation 1]
// (hifigan/models.py:96-103, one iteration of ResBlock.forward's loop), instead of two launches of the generic kernel.
//
// Why: at C = 32 / 64 a layer-granular conv does 2*C*k = 192 ... 1408 FLOP per output element against 8-12 B of HBM
// traffic — on or below the fp32 ridge (157 TFLOP/s : 6.3 TB/s = 25 FLOP/B) — and re-stages a weight tile every
// (16-channel chunk, tap) iteration for a K loop that is only C*k/2 = 48 ... 352 MFMAs long: SQ counters put the matrix
// pipe at 57 % (C = 64) and 44 % (C = 32) busy (profiles/r01_pmc_mfma_busy.md).  Here a workgroup keeps a column tile on
// chip for BOTH convs:
//
//   * LDS holds the x tile [C][N1 + 2*r1] (r1 = d*(k-1)/2: conv1's halo), stored ACTIVATED (LeakyReLU applied once by the
//     staging pass, not per MFMA operand), and the activated xt tile [C][N1 + k - 1]: x crosses HBM once per pair (plus
//     10-24 % halo re-reads served by L2), xt never leaves the CU, the raw residual operand is prefetched from global (L2)
//     into registers before conv2's K loop, y is written once: 2 tensor passes instead of 5.  conv1 is evaluated on N1
//     columns of which N1 - (k-1) produce outputs (conv2's halo is recomputed, not exchanged: 1-8 % extra MFMAs).
//     N1 = 256 (C = 32) / 128 (C = 64): <= 80 KB => TWO workgroups per CU, one stages / stores while the other multiplies;
//   * every wave owns ONE 32-row m-tile x TWO 32-column n-tiles; its weights stream L2 -> VGPR in MFMA A-fragment order,
//     packed host-side in the K loop's ITERATION order ([chunk][tap][half][m-tile][lane][4]: one linear walk, one
//     global_load_dwordx4 = the A operand of 4 k-steps = 8 MFMAs) through a 4-deep register ring: no weight tile in LDS,
//     no barrier inside a conv's K loop, three barriers per tile.  128 B of weights per MFMA — the per-CU L2 -> CU ingest
//     measured in tools/mfma_probe.hip carries 192;
//   * B operands are conflict-free ds_read_b32 (32 consecutive floats per half-wave); the LDS reads and the A load of a
//     later k-group are THREADED between the MFMAs of the current one (sched_group_barrier), because a wave issues in
//     order and a bunch of loads between two runs of MFMAs idles the pipe while they issue.
//
// conv_xl_kernel (C = 128 / 256) is the same K loop for ONE conv per launch with the whole activated x tile staged once.
//
// Accumulation order = the generic kernel's (16-channel chunk, tap, k) and the same epilogue expressions ((acc + b) + x
// [+ y_old]) => BITWISE equal to the two-launch path (tests/test_gpu_parity.py::test_vocoder_pair_kernel_bitwise).
#include <hip/hip_runtime.h>
#include "resblock_pair.h"
#include "xcd_map.h"
#ifndef XCD_MAP
#define XCD_MAP 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// -DPAIR_DBG=5 (tools/pair_phases.py): every wave stamps the cycle counter at its phase boundaries; results are unchanged
#ifndef PAIR_DBG
#define PAIR_DBG 0
#endif

namespace {

// Columns of xt per workgroup: 256 at C = 32 and 128 at C = 64 — both give 8 accumulator tiles (4 waves x 2) and
// <= 80 KB of LDS, i.e. TWO workgroups per CU: one stages / stores while the other multiplies (one 144-KB workgroup per
// CU at C = 64 left the matrix pipe idle through every staging and epilogue phase).
#define N1_OF(C) ((C) == 32 ? 256 : 128)
constexpr int NT = 2;            // n-tiles per wave
constexpr int RING = 4;          // register ring depth of the weight stream (k-groups of 8 input channels in flight)
constexpr int R1MAX = 25;        // largest conv1 halo: k = 11, d = 5

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// One conv of the pair on this wave's (m-tile mt) x (n-tiles nq*2, nq*2+1): acc = W * act(src), K loop over
// (16-channel chunk, tap, 8-channel half) in the generic kernel's order.  src: LDS tile [C][ld]; output column c reads
// src column c + tap * dil.  wfrag holds the A fragments in ITERATION order ([chunk][tap][half][m-tile][lane][4], packed by
// cmtts_finalize), so the weight stream is one linear walk: the only per-group index arithmetic left is a handful of
// scalar updates of the B-operand offset.  Loads are threaded between the MFMAs (one ds_read2_b32 + the A load of a
// later group per k-step): a wave's instructions issue in order, so loads bunched between two groups of MFMAs would
// leave the matrix pipe idle while they issue (first version of this kernel: 60-68 % pipe busy).
template <int C, int KT, int CIN = C>
__device__ __forceinline__ void conv_loop(f32x16 (&acc)[NT], const float* __restrict__ wfrag, const float* __restrict__ src,
                                          int ld, int dil, int mt, int col0, int lane) {
    constexpr int MTILES = C / 32;              // output m-tiles (C output channels); CIN input channels
    constexpr int NG = (CIN / 8) * KT;          // k-groups: (chunk, tap, half); a multiple of RING for CIN in {32, 64, ...}
    static_assert(NG % RING == 0, "no tail in the ring loop");
    const int l31 = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* wl = wfrag + mt * 256 + lane * 4;
    auto load_a = [&](f32x4& dst, int it) {
        dst = *reinterpret_cast<const f32x4*>(wl + (long)min(it, NG - 1) * (MTILES * 256));
    };
    const float* bl = src + khalf * ld + col0 + l31;     // this lane's B element of (row 0, tap 0)
    // scalar walk over (chunk, tap, half): boff = float offset of the group's first row / tap column
    int boff = 0, tap = 0, half = 0;
    auto advance = [&]() {
        if (half == 0) { half = 1; boff += 8 * ld; }
        else {
            half = 0; boff -= 8 * ld; ++tap; boff += dil;
            if (tap == KT) { tap = 0; boff += 16 * ld - KT * dil; }
        }
    };
    auto load_b = [&](float (&dst)[4][NT], int off) {
        const float* bs = bl + off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j) dst[kk][j] = bs[2 * kk * ld + j * 32];
    };
    f32x4 A[RING];
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load_a(A[s], s);
    float Bv[2][4][NT];
    load_b(Bv[0], boff);
#pragma unroll 1
    for (int it = 0; it < NG; it += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            advance();                                   // -> offset of group it + s + 1 (past the end: harmless, in-bounds reads)
            const int off_n = min(boff, (CIN - 8) * ld + (KT - 1) * max(dil, 0));   // past-the-end prefetch: harmless in-bounds read
            const float* bs = bl + off_n;
            load_a(A[(s + RING - 1) % RING], it + s + RING - 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int j = 0; j < NT; ++j) Bv[(s + 1) & 1][kk][j] = bs[2 * kk * ld + j * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][kk], Bv[s & 1][kk][j], acc[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one ds_read2_b32 (next group's k-step kk) ...
                if (kk == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // ... the A load of a later group ...
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);     // ... under this k-step's MFMAs
            }
        }
    }
}

// conv_loop for the transposed convs: two taps at dilation -1 (output column c reads tile columns c + 1, c), CIN input channels,
// m-tile `mt` of `mtiles` (run-time: s * CO / 32 stacked rows) in the iteration-order fragments [chunk][tap][half][mtiles][64][4]
template <int CIN>
__device__ __forceinline__ void conv_loop_rt(f32x16 (&acc)[NT], const float* __restrict__ wfrag, const float* __restrict__ src,
                                             int ld, int mt, int mtiles, int lane) {
    constexpr int KT = 2;
    constexpr int NG = (CIN / 8) * KT;
    static_assert(NG % RING == 0, "no tail in the ring loop");
    const int l31 = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* wl = wfrag + mt * 256 + lane * 4;
    const long gstride = (long)mtiles * 256;
    auto load_a = [&](f32x4& dst, int it) { dst = *reinterpret_cast<const f32x4*>(wl + (long)min(it, NG - 1) * gstride); };
    const float* bl = src + khalf * ld + 1 + l31;            // tap 0 reads column c + 1 (x[m]), tap 1 column c (x[m - 1])
    int boff = 0, tap = 0, half = 0;
    auto advance = [&]() {
        if (half == 0) { half = 1; boff += 8 * ld; }
        else {
            half = 0; boff -= 8 * ld; ++tap; boff -= 1;
            if (tap == KT) { tap = 0; boff += 16 * ld + KT; }
        }
    };
    auto load_b = [&](float (&dst)[4][NT], int off) {
        const float* bs = bl + off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j) dst[kk][j] = bs[2 * kk * ld + j * 32];
    };
    f32x4 A[RING];
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load_a(A[s], s);
    float Bv[2][4][NT];
    load_b(Bv[0], boff);
#pragma unroll 1
    for (int it = 0; it < NG; it += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            advance();
            const int off_n = min(boff, (CIN - 8) * ld);
            const float* bs = bl + off_n;
            load_a(A[(s + RING - 1) % RING], it + s + RING - 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int j = 0; j < NT; ++j) Bv[(s + 1) & 1][kk][j] = bs[2 * kk * ld + j * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][kk], Bv[s & 1][kk][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (kk == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            }
        }
    }
}

template <int C, int KT>
__global__ __launch_bounds__(256, 2) void resblock_pair_kernel(const PairArgs a) {
    constexpr int N1 = N1_OF(C);                // columns of xt a workgroup evaluates
    constexpr int XW = N1 + 2 * R1MAX, XTW = N1 + 10;
    constexpr int NWAVES = 4;
    constexpr int NTW = (C / 32) * (N1 / 32) / NWAVES;      // 32x32 tiles per wave = NT
    static_assert(NTW == NT, "every wave owns one m-tile x NT n-tiles");
    constexpr int R2 = (KT - 1) / 2;
    constexpr int TT = N1 - 2 * R2;             // output columns per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                           // [C][XW]   leaky(x), column j <-> t = t0 - R2 - r1 + j
    float* XTs = smem + C * XW;                 // [C][XTW]  leaky(xt), column c <-> t = t0 - R2 + c
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int WPM = NWAVES / (C / 32);      // waves per m-tile
    const int mt = w / WPM, nq = w % WPM;
    const int l31 = lane & 31;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * TT;
    const int T = a.T, dil = a.dil;
    const int r1 = dil * R2;
    const int xw = N1 + 2 * r1;                 // x columns this tile needs
    const float* xb = a.x + (long)b * a.bstride;
    // PAIR_DBG == 5: cycle stamps of every wave at the phase boundaries (tools/pair_phases.py)
    auto stamp = [&](int slot) {
        if (PAIR_DBG == 5 && a.dbg && lane == 0)
            a.dbg[(((long)b * gridDim.x + blockIdx.x) * NWAVES + w) * 8 + slot] = (long long)__builtin_readcyclecounter();
    };
    stamp(0);

    // ---- stage the x tile, ACTIVATED (zero outside [0, T)): rows over waves, columns over lanes; all of a lane's loads
    // (column blocks x rows) are in flight before the first LDS store: one HBM round trip per tile
    {
        const int tbase = t0 - R2 - r1;
        constexpr int ROWS_PER_WAVE = C / NWAVES;          // 8 / 16
        constexpr int XBLK = (XW + 63) / 64;               // 5 / 3
        float v[XBLK][ROWS_PER_WAVE];
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const int t_c = min(max(tbase + jb * 64 + lane, 0), T - 1);
#pragma unroll
            for (int q = 0; q < ROWS_PER_WAVE; ++q)
                v[jb][q] = xb[(long)(w * ROWS_PER_WAVE + q) * a.ld + t_c];
        }
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const int j = jb * 64 + lane;
            const int t = tbase + j;
            if (j < xw) {
#pragma unroll
                for (int q = 0; q < ROWS_PER_WAVE; ++q)
                    Xs[(w * ROWS_PER_WAVE + q) * XW + j] = (t >= 0 && t < T) ? leaky(v[jb][q], a.slope) : 0.f;
            }
        }
        // slack columns of the xt tile (read only by the k - 1 columns whose outputs are never stored): defined values
        if (tid < C) {
#pragma unroll
            for (int q = 0; q < XTW - N1; ++q) XTs[tid * XTW + N1 + q] = 0.f;
        }
    }
    stamp(1);
    __syncthreads();
    stamp(2);

    f32x16 acc[NT];
    const int col0 = nq * (NT * 32);
    // ---- conv1: xt column c = sum_tap W1[tap] . leaky(x)[c + tap * dil]
    conv_loop<C, KT>(acc, (const float*)a.w1f, Xs, XW, dil, mt, col0, lane);
    stamp(3);
    {   // epilogue: (acc + b1), zero outside the sequence (conv2's zero padding), stored ACTIVATED for conv2
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.b1[mt * 32 + acc_row(r, lane)];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = col0 + j * 32 + l31;
            const int t = t0 - R2 + c;
            const bool in = t >= 0 && t < T;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[j][r] + bi[r];
                XTs[(mt * 32 + acc_row(r, lane)) * XTW + c] = in ? leaky(v, a.slope) : 0.f;
            }
        }
    }
    // the residual operand x (raw; the LDS tile holds it activated) and, for the MRF sum, the old y: requested now, they
    // arrive under conv2's K loop (x was read by this workgroup a moment ago: L2)
    float* yb = a.y + (long)b * a.bstride;
    float xres[NT][16], yo[NT][16];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t_c = min(t0 + col0 + j * 32 + l31, T - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long off = (long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c;
            xres[j][r] = xb[off];
            yo[j][r] = a.accum ? yb[off] : 0.f;
        }
    }
    stamp(4);
    __syncthreads();
    stamp(5);

    // ---- conv2: y column o = sum_tap W2[tap] . xt_act[o + tap]   (o < TT valid)
    conv_loop<C, KT>(acc, (const float*)a.w2f, XTs, XTW, 1, mt, col0, lane);
    stamp(6);
    {   // epilogue: ((acc + b2) + x) [+ y_old]  — the generic kernel's expression order
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.b2[mt * 32 + acc_row(r, lane)];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int o = col0 + j * 32 + l31;
            const int t = t0 + o;
            const bool ok = o < TT && t < T;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + acc_row(r, lane);
                float v = acc[j][r] + bi[r];
                v += xres[j][r];
                if (a.accum) v += yo[j][r];
                if (ok) yb[(long)m * a.ld + t] = v;
            }
        }
    }
    stamp(7);
}

template <int C, int KT>
int launch_pair(const PairArgs& a, hipStream_t stream) {
    constexpr int N1 = N1_OF(C);
    constexpr int TT = N1 - (KT - 1);
    const size_t lds = (size_t)C * (N1 + 2 * R1MAX + N1 + 10) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_pair_kernel<C, KT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.T + TT - 1) / TT, a.B);
    hipLaunchKernelGGL((resblock_pair_kernel<C, KT>), grid, dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- One ResBlock conv of the WIDE stages (C = 128 / 256: the pair does not fit in LDS) on the same K loop: the
// workgroup's whole x tile [C][64 + (k-1)*d] is staged once, activated, and every wave walks all of K for its m-tile x two
// n-tiles with weights streamed L2 -> VGPR: no per-(chunk, tap) weight staging, no barrier in the K loop (the generic
// kernel: one barrier + one weight tile per 16 MFMAs per wave, 69 % pipe busy at these shapes).  C = 128: 58 KB of LDS,
// two workgroups per CU; C = 256: 117 KB, one 8-wave workgroup.  Epilogue = the generic kernel's: (acc + bias) [+ res]
// [+ y_old]; same accumulation order => bitwise equal.
constexpr int XL_BN = 64;

// NWV = waves per workgroup: C / 32 (all m-tiles of a column tile in one workgroup) or — round 4, launches of a few column tiles: one
// request through the C = 256 stage is 19 tiles — fewer, the m-tiles split over gridDim.z workgroups that each stage the tile: one
// wave per SIMD instead of two, four times the CUs; a wave's accumulation chains are the same, so are the bits
template <int C, int KT, int CIN = C, int NWV = C / 32>
__global__ __launch_bounds__(64 * NWV, NWV == C / 32 ? 2 : 1) void conv_xl_kernel(const ConvXlArgs a) {
    constexpr int XW = XL_BN + (KT - 1) * 5 + 2;            // widest halo of this kernel size (dilation <= 5): k = 7 at C = 128 leaves
                                                            // 49 KB per workgroup = three per CU, k = 3 38 KB
    constexpr int NWAVES = NWV;                             // one m-tile per wave, both n-tiles
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                        // [C][XW] leaky(x), column j <-> t = t0 - pad + j
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = (int)blockIdx.z * NWV + w;                // this wave's m-tile
    const int l31 = lane & 31;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * XL_BN;
    const int T = a.T, dil = a.dil;
    const int pad = dil * ((KT - 1) / 2);
    const int xw = XL_BN + 2 * pad;
    const float* xb = a.x + (long)b * (a.xbstride ? a.xbstride : a.bstride);
    {
        const int tbase = t0 - pad;
        constexpr int ROWS_PER_WAVE = CIN / NWAVES;         // 32 (16 for the 128 -> 256 predictor conv)
        static_assert(ROWS_PER_WAVE % 16 == 0, "staging walks 16 rows at a time");
        constexpr int XBLK = (XW + 63) / 64;                // 2
#pragma unroll
        for (int h = 0; h < ROWS_PER_WAVE; h += 16) {       // 32 loads in flight per lane
            float v[XBLK][16];
#pragma unroll
            for (int jb = 0; jb < XBLK; ++jb) {
                const int t_c = min(max(tbase + jb * 64 + lane, 0), T - 1);
#pragma unroll
                for (int q = 0; q < 16; ++q) v[jb][q] = xb[(long)(w * ROWS_PER_WAVE + h + q) * a.ld + t_c];
            }
#pragma unroll
            for (int jb = 0; jb < XBLK; ++jb) {
                const int j = jb * 64 + lane;
                const int t = tbase + j;
                if (j < xw) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        Xs[(w * ROWS_PER_WAVE + h + q) * XW + j] = (t >= 0 && t < T) ? leaky(v[jb][q], a.slope) : 0.f;
                }
            }
        }
    }
    // residual / old y of this wave's tiles: requested before the K loop, consumed after it
    float* yb = a.y + (long)b * a.bstride;
    const float* rb = a.res ? a.res + (long)b * a.bstride : nullptr;
    float xres[NT][16], yo[NT][16];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t_c = min(t0 + j * 32 + l31, T - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long off = (long)(mt * 32 + acc_row(r, lane)) * a.ld + t_c;
            xres[j][r] = rb ? rb[off] : 0.f;
            yo[j][r] = a.accum ? yb[off] : 0.f;
        }
    }
    __syncthreads();
    f32x16 acc[NT];
    conv_loop<C, KT, CIN>(acc, a.wf, Xs, XW, dil, mt, 0, lane);
    float bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bi[r] = a.bias[mt * 32 + acc_row(r, lane)];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = t0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[j][r] + bi[r];
            if (a.relu) v = v > 0.f ? v : 0.f;
            if (rb) v += xres[j][r];
            if (a.accum) v += yo[j][r];
            if (t < T) yb[(long)(mt * 32 + acc_row(r, lane)) * a.ld + t] = v;
        }
    }
}

#ifndef XLW_MT
#define XLW_MT(DIL, KT) ((DIL) == 1 || (KT) == 3 ? 2 : 1)      // m-tiles per wave of conv_xlw_kernel: two unless the tile (dilation 3 / 5 at k >= 7) would leave one wave per SIMD
#endif
// ---- conv_xl in its Winograd form (resblock_pair.h: WinoTab).  A workgroup computes NP output PAIRS (t, t + DIL) per m-tile: 32 pairs = 64
// columns at dilation 1, 30 pairs = 60 columns at dilation 3 / 5 (pair p = q DIL + r covers columns q 2 DIL + r and + DIL).  The activated x
// tile lives in LDS split by pair parity: column j = Q 2 DIL + R of the tile goes to E[Q DIL + R] (R < DIL) or O[Q DIL + R - DIL], so that
// X(m) of pair p — the input `m` dilated taps to the right of the pair's first output — is E[p + (m / 2) DIL] (m even) or
// O[p + (m / 2) DIL] (m odd): unit stride across the lanes, compile-time offsets.  Every wave owns one m-tile and the four accumulators
// M0..M3 of the tile's pairs; per table entry and k-step one ds_read2_b32 + one VALU form the transformed input, one MFMA consumes it.
// NOT bitwise the direct form (fp32 Winograd: the same products regrouped; ~1e-6 relative per conv).
// MT = m-tiles per wave: with two, a transformed input feeds two MFMAs (1.25 instead of 2.25 LDS / VALU / load instructions per MFMA — what
// bounds this kernel: at MT = 1 the k = 3 instances were SLOWER than the direct form), the workgroup has C / 64 waves and smaller tiles keep
// four (C = 128) / two (C = 256) of them on a CU; the residual operands are then read after the K loop (no registers to park them in).
template <int C, int KT, int DIL, int MT>
__global__ __launch_bounds__(64 * (C / (32 * MT)), 2) void conv_xlw_kernel(const ConvXlArgs a) {
    using TAB = WinoTab<KT>;
    constexpr int NWAVES = C / (32 * MT);
    constexpr int NP = DIL == 1 ? 32 : 30;                  // pairs per tile
    constexpr int BN = 2 * NP;                              // output columns per tile
    constexpr int NEO = 32 + ((KT - 1) / 2) * DIL;          // entries of E and of O per row
    constexpr int XWW = 2 * NEO;
    constexpr int PAD = DIL * ((KT - 1) / 2);
    constexpr int XIN = BN + (KT - 1) * DIL;                // staged input columns
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                        // [C][E | O]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt0 = w * MT;
    const int l31 = lane & 31, khalf = lane >> 5;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * BN;
    const int T = a.T;
    const float* xb = a.x + (long)b * a.bstride;
    {
        const int tbase = t0 - PAD;
        constexpr int ROWS_PER_WAVE = C / NWAVES;           // 32 MT
        constexpr int XBLK = (XIN + 63) / 64;
        int eo[XBLK];
#pragma unroll
        for (int jb = 0; jb < XBLK; ++jb) {
            const int j = jb * 64 + lane, Q = j / (2 * DIL), R = j - Q * (2 * DIL);
            eo[jb] = R < DIL ? Q * DIL + R : NEO + Q * DIL + R - DIL;
        }
#pragma unroll
        for (int h = 0; h < ROWS_PER_WAVE; h += 16) {
            float v[XBLK][16];
#pragma unroll
            for (int jb = 0; jb < XBLK; ++jb) {
                const int t_c = min(max(tbase + jb * 64 + lane, 0), T - 1);
#pragma unroll
                for (int q = 0; q < 16; ++q) v[jb][q] = xb[(long)(w * ROWS_PER_WAVE + h + q) * a.ld + t_c];
            }
#pragma unroll
            for (int jb = 0; jb < XBLK; ++jb) {
                const int j = jb * 64 + lane;
                const int t = tbase + j;
                if (j < XIN) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        Xs[(w * ROWS_PER_WAVE + h + q) * XWW + eo[jb]] = (t >= 0 && t < T) ? leaky(v[jb][q], a.slope) : 0.f;
                }
            }
        }
    }
    // this lane's pair: columns ca and ca + DIL of the tile
    const int pq = l31 / DIL, pr = l31 - pq * DIL;
    const int ta = t0 + pq * 2 * DIL + pr, tb = ta + DIL;
    const bool pv = l31 < NP;
    float* yb = a.y + (long)b * a.bstride;
    const float* rb = a.res ? a.res + (long)b * a.bstride : nullptr;
    constexpr bool PRE = MT == 1;                           // residual / old y requested before the K loop
    float xres[PRE ? 2 : 1][16], yo[PRE ? 2 : 1][16];
    if (PRE) {
        const int ta_c = min(ta, T - 1), tb_c = min(tb, T - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = (long)(mt0 * 32 + acc_row(r, lane)) * a.ld;
            xres[0][r] = rb ? rb[row + ta_c] : 0.f;
            xres[PRE ? 1 : 0][r] = rb ? rb[row + tb_c] : 0.f;
            yo[0][r] = a.accum ? yb[row + ta_c] : 0.f;
            yo[PRE ? 1 : 0][r] = a.accum ? yb[row + tb_c] : 0.f;
        }
    }
    __syncthreads();
    f32x16 M[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) M[i][q][r] = 0.f;
    {
        // weights: iteration order [chunk][entry][half][m-tile][lane][4]; one fragment = the A operands of 4 k-steps of one entry
        constexpr int NF = (C / 16) * TAB::N * 2;           // fragment groups of a wave
        constexpr int RINGW = (TAB::N * 2) % 6 == 0 ? 6 : 4;      // groups in flight + the one in use; divides a chunk's count (8 / 20 / 30), so ring slots are compile-time
        static_assert((TAB::N * 2) % RINGW == 0, "ring slots repeat per chunk");
        const float* wl = a.wf + mt0 * 256 + lane * 4;
        auto load_a = [&](f32x4 (&dst)[MT], int it) {
#pragma unroll
            for (int i = 0; i < MT; ++i) dst[i] = *reinterpret_cast<const f32x4*>(wl + (long)min(it, NF - 1) * ((C / 32) * 256) + i * 256);
        };
        f32x4 A[RINGW][MT];
#pragma unroll
        for (int s = 0; s < RINGW - 1; ++s) load_a(A[s], s);
        const float* bl = Xs + khalf * XWW + l31;
        // LDS operands of step n + 1 (one table entry x one half-chunk: 4 k-steps) are requested in front of step n's MFMAs
        float XA[2][4], XB[2][4];
        auto loadx = [&](float (&xa)[4], float (&xb)[4], const float* bc, int e, int h) {
            const int ea = TAB::e[e].a, ebb = TAB::e[e].b, sg = TAB::e[e].sgn;
            const int offa = ((ea & 1) ? NEO : 0) + (ea >> 1) * DIL, offb = ((ebb & 1) ? NEO : 0) + (ebb >> 1) * DIL;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float* row = bc + (8 * h + 2 * kk) * XWW;
                xa[kk] = row[offa];
                xb[kk] = sg != 0 ? row[offb] : 0.f;
            }
        };
        loadx(XA[0], XB[0], bl, 0, 0);
#pragma unroll 1
        for (int c = 0; c < C / 16; ++c) {
            const float* bc = bl + c * 16 * XWW;
            int it = c * (TAB::N * 2);
#pragma unroll
            for (int e = 0; e < TAB::N; ++e) {
                const int sg = TAB::e[e].sgn, ac = TAB::e[e].acc;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = e * 2 + h;
                    const int slot = n % RINGW;
                    load_a(A[(slot + RINGW - 1) % RINGW], it + RINGW - 1);
                    if (n + 1 < TAB::N * 2) loadx(XA[(n + 1) & 1], XB[(n + 1) & 1], bc, (n + 1) >> 1, (n + 1) & 1);
                    else loadx(XA[(n + 1) & 1], XB[(n + 1) & 1], bl + min(c + 1, C / 16 - 1) * 16 * XWW, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        float v = XA[n & 1][kk];
                        if (sg > 0) v = v + XB[n & 1][kk];
                        else if (sg < 0) v = v - XB[n & 1][kk];
#pragma unroll
                        for (int i = 0; i < MT; ++i) M[i][ac] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[slot][i][kk], v, M[i][ac], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ++it;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = a.bias[(mt0 + i) * 32 + acc_row(r, lane)];
        float xr2[2][16], yo2[2][16];
        if (!PRE) {
            const int ta_c = min(ta, T - 1), tb_c = min(tb, T - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = (long)((mt0 + i) * 32 + acc_row(r, lane)) * a.ld;
                xr2[0][r] = rb ? rb[row + ta_c] : 0.f;
                xr2[1][r] = rb ? rb[row + tb_c] : 0.f;
                yo2[0][r] = a.accum ? yb[row + ta_c] : 0.f;
                yo2[1][r] = a.accum ? yb[row + tb_c] : 0.f;
            }
        }
        if (pv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = (long)((mt0 + i) * 32 + acc_row(r, lane)) * a.ld;
                float va = ((M[i][0][r] + M[i][1][r]) + M[i][2][r]) + bi[r];
                float vb = ((M[i][1][r] - M[i][2][r]) - M[i][3][r]) + bi[r];
                if (a.relu) { va = va > 0.f ? va : 0.f; vb = vb > 0.f ? vb : 0.f; }
                if (rb) { va += PRE ? xres[0][r] : xr2[0][r]; vb += PRE ? xres[PRE ? 1 : 0][r] : xr2[1][r]; }
                if (a.accum) { va += PRE ? yo[0][r] : yo2[0][r]; vb += PRE ? yo[PRE ? 1 : 0][r] : yo2[1][r]; }
                if (ta < T) yb[row + ta] = va;
                if (tb < T) yb[row + tb] = vb;
            }
        }
    }
}

template <int C, int KT, int DIL>
int launch_xlw(const ConvXlArgs& a, hipStream_t stream) {
    constexpr int MT = XLW_MT(DIL, KT);
    constexpr int NEO = 32 + ((KT - 1) / 2) * DIL;
    constexpr int BN = DIL == 1 ? 64 : 60;
    const size_t lds = (size_t)C * 2 * NEO * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xlw_kernel<C, KT, DIL, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.T + BN - 1) / BN, a.B);
    hipLaunchKernelGGL((conv_xlw_kernel<C, KT, DIL, MT>), grid, dim3(64 * (C / (32 * MT))), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int C, int KT>
int launch_xlw_dil(const ConvXlArgs& a, hipStream_t s) {
    if (a.dil == 1) return launch_xlw<C, KT, 1>(a, s);
    if (a.dil == 3) return launch_xlw<C, KT, 3>(a, s);
    if (a.dil == 5) return launch_xlw<C, KT, 5>(a, s);
    return -2;
}

int g_xl_split = 1;      // internal switch "voc_xl_split": m-tiles of conv_xl over several workgroups when the launch is a few column tiles

template <int C, int KT, int CIN = C>
int launch_xl(const ConvXlArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)CIN * (XL_BN + (KT - 1) * 5 + 2) * sizeof(float);
    constexpr int NWS = 2;                      // waves per workgroup of the split form: C / 64 workgroups per column tile
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xl_kernel<C, KT, CIN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xl_kernel<C, KT, CIN, NWS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const long tiles = (long)((a.T + XL_BN - 1) / XL_BN) * a.B;
    // two waves per SIMD in C / 32-wave workgroups once every CU has one; below that the m-tiles spread over C / 64 workgroups
    // (C = 128 has one wave per SIMD already; above ~40 tiles the four-fold staging of the split form costs what its K loops gain: one
    //  510-frame request, 64 tiles, 3.81 ms unsplit / 3.95 split; one 150-frame request, 19 tiles, 2.43 / 1.99)
    if (g_xl_split && C / 32 > 4 && tiles * (C / 64) <= 160) {
        dim3 grid((a.T + XL_BN - 1) / XL_BN, a.B, C / (32 * NWS));
        hipLaunchKernelGGL((conv_xl_kernel<C, KT, CIN, NWS>), grid, dim3(64 * NWS), lds, stream, a);
    } else {
        dim3 grid((a.T + XL_BN - 1) / XL_BN, a.B);
        hipLaunchKernelGGL((conv_xl_kernel<C, KT, CIN>), grid, dim3(2 * C), lds, stream, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------------------------------
// ConvTranspose1d(CIN -> CO, kernel 2 s, stride s, padding s / 2) of the HiFi-GAN upsamplers (hifigan/models.py:152-153), all s
// output phases in one X-resident launch:  y[co][s m + r - s/2] = b[co] + sum_ci sum_{q in {0, 1}} W[ci][co][r + s q] act(x[ci][m - q]),
// i.e. a two-tap convolution with s * CO output rows (row = r * CO + co) whose store interleaves the phases.  The generic kernel
// ran it as s separate launches-in-z that each re-staged the input through 16-channel chunks (47-57 % of the fp32 pipe, 5.2 ms
// of a 32 x 512-frame batch); here a workgroup stages act(x) = leaky_relu(x / pre_div, 0.1) of 64 + 1 columns once for NW
// m-tiles (NW <= 8 waves: 256 rows) and runs conv_loop on it (taps in the generic kernel's order q = 0, 1: dilation -1).  Same
// staging arithmetic (true division, then the slope), accumulation order and epilogue ((acc + b)) => the same bits.
struct ConvTArgs {
    const float* x;       // [B][CIN][ldx]
    float* y;             // [B][CO][ldy]
    const float* wf;      // iteration-order fragments of the [2][CIN][s * CO] two-tap weights
    const float* bias;    // [CO]
    long xbstride, ybstride;
    int B, CO, Ti, To, ldx, ldy, s;
    float pre_div, slope;
};

template <int CIN, int NW>
__global__ __launch_bounds__(64 * NW, CIN >= 512 ? 2 : (NW >= 8 ? 4 : 2)) void convT_xl_kernel(const ConvTArgs a, int MTILES_rt) {
    constexpr int XW = XL_BN + 1 + 2;                        // columns t0 - 1 .. t0 + 63 (+ pad against bank conflicts)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                        // [CIN][XW] act(x), column j <-> m = t0 - 1 + j
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (XCD_MAP) xcd_tile(bx_, by_);          // consecutive tiles of an utterance on ONE XCD (xcd_map.h)
    const int b = by_;
    const int t0 = bx_ * XL_BN;
    const int Ti = a.Ti;
    const float* xb = a.x + (long)b * a.xbstride;
    {
        constexpr int ROWS = CIN / NW;                       // rows per wave
        const int jcol[2] = {lane, lane + 64};
#pragma unroll
        for (int h = 0; h < ROWS; h += 16) {
            float v[2][16];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int m_c = min(max(t0 - 1 + min(jcol[jb], XL_BN), 0), Ti - 1);      // lanes past the last column re-read it (no traffic) instead of the next tile's columns
#pragma unroll
                for (int q = 0; q < 16; ++q) v[jb][q] = xb[(long)(w * ROWS + h + q) * a.ldx + m_c];
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int j = jcol[jb];
                const int m = t0 - 1 + j;
                const bool ok = m >= 0 && m < Ti;
                if (j < XL_BN + 1) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        float u = ok ? v[jb][q] : 0.f;
                        if (a.pre_div != 1.0f) u = u / a.pre_div;
                        u = u > 0.f ? u : u * a.slope;
                        Xs[(w * ROWS + h + q) * XW + j] = u;
                    }
                }
            }
        }
    }
    __syncthreads();
    // wave -> (phase, 32-channel block); the workgroup walks over its share of the channel blocks with the x tile resident: one
    // staging pass feeds all s * CO stacked rows (a grid split over blockIdx.z only where the tiles alone cannot fill the chip).
    // (Interleaving the phases through an LDS output tile to store whole lines instead of 4 bytes at a stride of 4 s was tried:
    // no gain, the stores are not what bounds this kernel.)
    const int S = a.s;
    const int phase = w % S, cbl = w / S;                    // NW % s == 0 (launcher)
    const int per = NW / S;                                  // channel blocks per pass
    const int passes = (a.CO / 32) / per / gridDim.z;
    const int pd = S / 2;
    float* yb = a.y + (long)b * a.ybstride;
    for (int ps = 0; ps < passes; ++ps) {
        const int cb = (blockIdx.z * passes + ps) * per + cbl;
        const int mt = phase * (a.CO / 32) + cb;
        f32x16 acc[NT];
        conv_loop_rt<CIN>(acc, a.wf, Xs, XW, mt, MTILES_rt, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb * 32 + acc_row(r, lane);
            const float bi = a.bias[co];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = t0 + j * 32 + l31;
                const int t = n * S + phase - pd;
                if (n <= Ti && t >= 0 && t < a.To) yb[(long)co * a.ldy + t] = acc[j][r] + bi;
            }
        }
    }
}

template <int CIN, int NW>
int launch_convT(const ConvTArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)CIN * (XL_BN + 3) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(convT_xl_kernel<CIN, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const int mtiles = a.s * a.CO / 32;
    if (NW % a.s || mtiles % NW) return -2;                  // a pass = all s phases of NW / s channel blocks
    const int npass = mtiles / NW;                           // passes over the x tile if one workgroup did them all
    const long tiles = (long)((a.Ti + 1 + XL_BN - 1) / XL_BN) * a.B;
    int zs = 1;                                              // split the passes over blockIdx.z until ~16 workgroups per CU exist (a workgroup of ups1 runs 0.3 ms: with 2080 of them on 512 slots the fifth, nearly empty round costs 20 %)
    while (tiles * zs < 4096 && zs * 2 <= npass && npass % (zs * 2) == 0) zs *= 2;
    dim3 grid((a.Ti + 1 + XL_BN - 1) / XL_BN, a.B, zs);
    hipLaunchKernelGGL((convT_xl_kernel<CIN, NW>), grid, dim3(64 * NW), lds, stream, a, mtiles);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

long long* g_pair_dbg = nullptr;

}  // namespace

extern "C" void cmtts_pair_set_debug(long long* dbg) { g_pair_dbg = dbg; }
extern "C" int cmtts_xl_set_split(int on) { const int p = g_xl_split; if (on == 0 || on == 1) g_xl_split = on; return p; }

// 0 = launched, -2 = shape not covered (the caller runs the two generic launches), -3 = HIP error
extern "C" int cmtts_launch_resblock_pair(const PairArgs* ap, void* stream_) {
    PairArgs a = *ap;
    a.dbg = g_pair_dbg;
    hipStream_t s = (hipStream_t)stream_;
    if (a.B <= 0 || a.T <= 0) return 0;
    if (a.dil * (a.k - 1) / 2 > R1MAX || a.x == a.y) return -2;
    if (a.C == 64) {
        if (a.k == 3) return launch_pair<64, 3>(a, s);
        if (a.k == 7) return launch_pair<64, 7>(a, s);
        if (a.k == 11) return launch_pair<64, 11>(a, s);
    } else if (a.C == 32) {
        if (a.k == 3) return launch_pair<32, 3>(a, s);
        if (a.k == 7) return launch_pair<32, 7>(a, s);
        if (a.k == 11) return launch_pair<32, 11>(a, s);
    }
    return -2;
}

// One conv (k taps, dilation dil, 'same' padding) of a C = 128 / 256 ResBlock: 0 = launched, -2 = not covered, -3 = HIP error
extern "C" int cmtts_launch_conv_xl(const ConvXlArgs* ap, void* stream_) {
    const ConvXlArgs& a = *ap;
    hipStream_t s = (hipStream_t)stream_;
    if (a.B <= 0 || a.T <= 0) return 0;
    if (a.dil < 1 || a.dil > 5 || a.x == a.y) return -2;          // the X tile is sized for dilation <= 5 (HiFi-GAN: 1, 3, 5)
    if (a.cin && a.cin != a.C) {      // variance-predictor input conv over frames: cwt_hidden 128 -> filter_size 256, kernel 5
        if (a.C == 256 && a.cin == 128 && a.k == 5 && !a.res && !a.accum) return launch_xl<256, 5, 128>(a, s);
        return -2;
    }
    if (a.C == 128) {
        if (a.k == 3) return launch_xl<128, 3>(a, s);
        if (a.k == 7) return launch_xl<128, 7>(a, s);
        if (a.k == 11) return launch_xl<128, 11>(a, s);
    } else if (a.C == 256) {
        if (a.k == 3) return launch_xl<256, 3>(a, s);
        if (a.k == 5) return launch_xl<256, 5>(a, s);        // variance-predictor convs over frames (filter_size 256, kernel 5)
        if (a.k == 7) return launch_xl<256, 7>(a, s);
        if (a.k == 11) return launch_xl<256, 11>(a, s);
    }
    return -2;
}

extern "C" int cmtts_launch_conv_xlw(const ConvXlArgs* ap, void* stream_) {
    const ConvXlArgs& a = *ap;
    hipStream_t s = (hipStream_t)stream_;
    if (a.B <= 0 || a.T <= 0) return 0;
    if (a.x == a.y || (a.cin && a.cin != a.C)) return -2;
    // launches of a few column tiles: the direct form (its split instance spreads the m-tiles over more workgroups; one 100-frame request
    // through the Winograd form: 2.7 instead of 2.15 ms for the generator)
    if (!a.wino_force && (long)((a.T + 63) / 64) * a.B < 1024) return -2;
#ifdef XLW_ONLY_K            // quick experimental builds: one kernel size
    if (a.k != XLW_ONLY_K) return -2;
#endif
    if (a.C == 128) {
        if (a.k == 3) return launch_xlw_dil<128, 3>(a, s);
        if (a.k == 7) return launch_xlw_dil<128, 7>(a, s);
        if (a.k == 11) return launch_xlw_dil<128, 11>(a, s);
    } else if (a.C == 256) {
        if (a.k == 3) return launch_xlw_dil<256, 3>(a, s);
        if (a.k == 7) return launch_xlw_dil<256, 7>(a, s);
        if (a.k == 11) return launch_xlw_dil<256, 11>(a, s);
    } else if (a.C == 64) {      // the C = 64 stage as two launches per pair (xt through HBM) where 2/3 of the MFMAs outweigh the two tensor passes the pair kernel saves
        if (a.k == 7) return launch_xlw_dil<64, 7>(a, s);
        if (a.k == 11) return launch_xlw_dil<64, 11>(a, s);
    }
    return -2;
}

// HiFi-GAN upsampler ConvTranspose1d(cin -> co, kernel 2 s, stride s, padding s / 2) on [B][cin][ldx] -> [B][co][ldy], input
// activation leaky_relu(x / pre_div, slope).  wf: to_fragment_iter_order of the [2][cin][s * co] two-tap weights (row = phase * co +
// channel).  0 = launched, -2 = shape not covered, -3 = HIP error.
extern "C" int cmtts_launch_convT(const float* x, float* y, const float* wf, const float* bias, long xbstride, long ybstride, int B,
                                  int cin, int co, int Ti, int To, int ldx, int ldy, int s, float pre_div, float slope, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (B <= 0 || Ti <= 0) return 0;
    if (!wf || (s * co) % 32 || To != Ti * s || s < 2 || (s & 1)) return -2;
    ConvTArgs a{x, y, wf, bias, xbstride, ybstride, B, co, Ti, To, ldx, ldy, s, pre_div, slope};
    const int mtiles = s * co / 32;
    if (cin == 512 && mtiles >= 8) return launch_convT<512, 8>(a, st);
    if (cin == 256 && mtiles >= 8) return launch_convT<256, 8>(a, st);
    if (cin == 128 && mtiles >= 4) return launch_convT<128, 4>(a, st);
    if (cin == 64 && mtiles >= 2) return launch_convT<64, 2>(a, st);
    return -2;
}
