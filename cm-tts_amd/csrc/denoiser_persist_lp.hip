// Persistent denoiser stack with 16-bit MFMA operands (bf16 / fp16, fp32 accumulate): denoiser_persist.hip's
// structure (all residual layers in one launch, x and the skip sum resident in fp32 registers, edge columns exchanged
// between tiles as tagged granules, two barriers per layer) around resblock_fused_lp.hip's contractions
// (v_mfma_f32_32x32x16_{bf16,f16}; u and z transposed and converted in LDS, weights in 16-bit fragment order).
// BASELINE.json configs[2] (bf16) and configs[4] (fp16 denoiser).  Per layer HBM sees cp only (1 KB/frame instead of
// 5 KB); at this MFMA rate the layer time is set by the weight fill (1.05 MB per workgroup per layer).
// Same arithmetic and (tap, k-group) accumulation order as resblock_fused_lp.hip: BITWISE equal to the per-layer
// 16-bit kernels (tests/test_gpu_parity.py::test_persistent_denoiser_lp_bitwise).
//
// MODE 3 ("fp16x3"): every operand is carried as TWO fp16 numbers, hi = fp16(v) and lo = fp16(v - hi) (22 significant
// bits together), and every product as three MFMAs, a_lo b_hi + a_hi b_lo + a_hi b_hi, accumulated in fp32 (the dropped
// a_lo b_lo term is 2^-22 relative): fp32-class results (|d mel| vs float64 within 2x of the exact-fp32 kernels', tested)
// at 3/16 of the fp32 MFMA cost.  The weight stream doubles (hi and lo fragments = the fp32 kernel's bytes), so a layer
// takes about half of the fp32 kernel's time instead of a quarter.  Exploratory: the bench headline stays exact fp32.
#include <hip/hip_runtime.h>
#include "cvt16.h"
#include "gate.h"
#include "persist_args.h"
#include "persist_tail.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) unsigned long long gu64;

namespace {

constexpr int C = 256;
constexpr int NW = 8;           // waves per workgroup, each owning 2 m-tiles x 2 n-tiles
constexpr int MT = 2;
constexpr int RING = 6;         // 16-channel k-groups of weights in flight (4 in the split mode: two fragment sets)
constexpr int FN = 64;
constexpr int NT = FN / 32;
constexpr int RS = 260;         // 16-bit elements per LDS row (520 B)
constexpr unsigned SPIN_LIMIT = 1u << 20;

__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float ldg(const float* base, unsigned idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 4u));
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned tag, float v) {
    __hip_atomic_store((gu64*)g, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if (MODE == 1)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// fp32 pair -> one dword of two 16-bit values in the hi image (and, MODE 3, the fp16 remainders in the lo image IMG
// elements further on)
// fp16 modes: an operand beyond the fp16 range (|v| > 65504, or NaN) cannot be represented — the conversion saturates / overflows and the
// result is finite but WRONG (measured: a mel of the right shape and the wrong values), so the u conversions record it and the kernel
// reports it through the pinned error word (code 3, cmtts_poll_error); bf16 has fp32's exponent range and no such check.
template <int MODE>
__device__ __forceinline__ void note_range(bool& ovf, float v0, float v1, bool valid) {
    if (MODE >= 2) ovf |= valid && !(fabsf(v0) <= 65504.0f && fabsf(v1) <= 65504.0f);
}
template <int MODE>
__device__ __forceinline__ void put2(unsigned short* hi_img, int off, float v0, float v1, bool valid, int IMG) {
    if (MODE != 3) {
        *reinterpret_cast<unsigned*>(hi_img + off) = valid ? pack16<MODE>(v0, v1) : 0u;
    } else {
        const unsigned ph = pack16<2>(v0, v1);
        const cvt_f16x2 h = __builtin_bit_cast(cvt_f16x2, ph);
        const unsigned pl = pack16<2>(v0 - (float)h[0], v1 - (float)h[1]);
        *reinterpret_cast<unsigned*>(hi_img + off) = valid ? ph : 0u;
        *reinterpret_cast<unsigned*>(hi_img + IMG + off) = valid ? pl : 0u;
    }
}
template <int MODE>
__device__ __forceinline__ void put1(unsigned short* hi_img, int off, float v, bool valid, int IMG) {
    if (MODE != 3) {
        hi_img[off] = valid ? (unsigned short)pack16<MODE>(v, 0.f) : (unsigned short)0;
    } else {
        const unsigned ph = pack16<2>(v, 0.f);
        const cvt_f16x2 h = __builtin_bit_cast(cvt_f16x2, ph);
        hi_img[off] = valid ? (unsigned short)ph : (unsigned short)0;
        hi_img[IMG + off] = valid ? (unsigned short)pack16<2>(v - (float)h[0], 0.f) : (unsigned short)0;
    }
}

template <int MODE>
__global__ __launch_bounds__(64 * NW, 2) void denoiser_persist_lp_kernel(const PersistArgs a) {
    constexpr int MM = MODE == 3 ? 2 : MODE;          // MFMA element type: fp16 in the split mode
    constexpr int RINGM = MODE == 3 ? 4 : RING;       // even: the B double buffer alternates with s
    constexpr int IMG = (2 * FN + 2) * RS;            // elements of one set of images (u^T + z^T); the lo set follows the hi set
    constexpr long W3LO = (long)3 * (C / 16) * (2 * C / 32) * 64, WOLO = (long)(C / 16) * (2 * C / 32) * 64;   // u32x4 per fragment set
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
    unsigned short* ut = lds16;                       // u^T [FN + 2][RS], row j = frame t0 - 1 + j
    unsigned short* zt = lds16 + (FN + 2) * RS;       // z^T [FN][RS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int t0 = tile * FN;
    const int T = a.T;
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* cp_b = a.cp + (long)b * a.cp_bstride;
    const float* dp_b = a.dp + (long)b * a.vec_stride;
    const float* dv_b = a.d + (long)b * a.vec_stride;
    const int mrow0 = w * 32;                         // this wave's 32 rows of x (state tile 0) and of the skip sum (tile 1)
    bool ovf = false;                                 // fp16 modes: a conv input left the fp16 range (note_range)
    // one word per wave behind the edge scratch: the last layer whose gate output (z^T channels 32 w ..) this wave has written — what the output
    // projection waits for, k-block by k-block, instead of barrier (3) (denoiser_persist.hip, round 5: same protocol, same argument)
    int* zflag = reinterpret_cast<int*>(lds16 + (MODE == 3 ? 2 : 1) * IMG) + NW * 64;
    if (tid < NW) zflag[tid] = 0;

    // ---- layer-0 staging (as resblock_fused_lp.hip): u^T[j][m] = cvt(cp + (x + dp)); lane = frame, waves over channel pairs
    {
        const float* xin = a.x0 + (long)b * C * T;
        const int t = t0 + lane;
        const int t_c = min(t, T - 1);
#pragma unroll 1
        for (int i = 0; i < C / (2 * NW); i += 4) {
            float x0[4], x1[4], c0[4], c1[4], d0[4], d1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = 2 * (w + NW * (i + q));
                x0[q] = xin[(unsigned)(m * T + t_c)];
                x1[q] = xin[(unsigned)((m + 1) * T + t_c)];
                c0[q] = cp_b[(unsigned)(m * T + t_c)];
                c1[q] = cp_b[(unsigned)((m + 1) * T + t_c)];
                d0[q] = dp_b[m];
                d1[q] = dp_b[m + 1];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = 2 * (w + NW * (i + q));
                const float u0 = c0[q] + (x0[q] + d0[q]);
                const float u1 = c1[q] + (x1[q] + d1[q]);
                put2<MODE>(ut, (1 + lane) * RS + m, u0, u1, t < T, IMG);
                note_range<MODE>(ovf, u0, u1, t < T);
            }
        }
        {
            const int m = tid & (C - 1);
            const bool right = tid >= C;
            const int th = right ? t0 + FN : t0 - 1;
            const int thc = min(max(th, 0), T - 1);
            const float uh = cp_b[(unsigned)(m * T + thc)] + (xin[(unsigned)(m * T + thc)] + dp_b[m]);
            put1<MODE>(ut, (right ? FN + 1 : 0) * RS + m, uh, th >= 0 && th < T, IMG);
            note_range<MODE>(ovf, uh, 0.f, th >= 0 && th < T);
        }
    }
    f32x16 st[MT][NT];
    {
        const float* xin = a.x0 + (long)b * C * T;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t_c = min(t0 + j * 32 + l31, T - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[i][j][r] = i == 0 ? ldg(xin, (unsigned)((mrow0 + acc_row(r, lane)) * T + t_c)) : 0.f;
            }
    }

    f32x16 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // weights [group][m-tile (16)][lane][8 x 16-bit]: k=3 conv tiles 2w, 2w+1; projection tiles w (residual half), NW+w (skip half).
    // Operand sets: index 0 = the (hi) fragments; MODE 3 adds index 1 = the lo fragments (W + W3LO / WOLO, LDS image + IMG)
    constexpr int NS = MODE == 3 ? 2 : 1;
    auto load_a = [&](u32x4 (&dst)[NS][MT], const void* wfrag, int group) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                dst[q][i] = *(reinterpret_cast<const u32x4*>(wfrag) + q * W3LO + ((long)group * (2 * C / 32) + w * MT + i) * 64 + lane);
    };
    auto load_ao = [&](u32x4 (&dst)[NS][MT], const void* wfrag, int group) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                dst[q][i] = *(reinterpret_cast<const u32x4*>(wfrag) + q * WOLO + ((long)group * (2 * C / 32) + i * NW + w) * 64 + lane);
    };
    auto load_b = [&](u32x4 (&dst)[NS][NT], const unsigned short* src, int kg, int row_off) {      // 16 bytes = 8 k-values of one frame
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned short* p = src + q * IMG + (j * 32 + l31 + row_off) * RS + kg * 16 + khalf * 8;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(p);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(p + 4);
                dst[q][j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            }
    };
    auto mma_group = [&](const u32x4 (&af)[NS][MT], const u32x4 (&bv)[NS][NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (MODE == 3) {      // small terms first: a_lo b_hi + a_hi b_lo, then a_hi b_hi
                    acc[i][j] = mma16<MM>(af[1][i], bv[0][j], acc[i][j]);
                    acc[i][j] = mma16<MM>(af[0][i], bv[1][j], acc[i][j]);
                }
                acc[i][j] = mma16<MM>(af[0][i], bv[0][j], acc[i][j]);
            }
    };

    bool gave_up = false;
    // -DLP_STAMP (tools/lp_phases.sh; a timing-only build loaded through CMTTS_LIB, never the product library): cycle stamps of
    // layer NL / 2 per wave, the slots of tools/persist_timing.py
#ifdef LP_STAMP
    const int bid_dbg = blockIdx.x + gridDim.x * blockIdx.y;
#define LPSTAMP(slot) do { if (a.dbg && l == a.NL / 2 && lane == 0) a.dbg[((long)bid_dbg * NW + w) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LPSTAMP(slot) do { } while (0)
#endif
    for (int l = 0; l < a.NL; ++l) {
        const bool more = l + 1 < a.NL;
        constexpr int NGB = 3 * (C / 16);        // k=3 conv: group = tap * 16 + k-group
        constexpr int NGC = C / 16;
        u32x4 A[RINGM][NS][MT];
#pragma unroll
        for (int s = 0; s < RINGM - 1; ++s) load_a(A[s], a.W3f[l], s);         // the weight stream does not depend on u
        LPSTAMP(0);
        __syncthreads();   // (1) u^T of layer l complete
        LPSTAMP(1);
        if (more) {        // pull the next layer's cp tile towards L2: one dword per 128-B line
            const float* cpn = cp_b + (long)(l + 1) * C * T;
            const int tl = opaque(tid);
            const float warm = cpn[(unsigned)((tl >> 1) * T + min(t0 + (tl & 1) * 32, T - 1))];
            asm volatile("" ::"v"(warm));
        }

        // =========================================================== phase B: gated k=3 conv, 48 k-groups
        {
            zero_acc();
            u32x4 Bv[2][NS][NT];
            load_b(Bv[0], ut, 0, 0);
#pragma unroll 1
            for (int it = 0; it < NGB; it += RINGM) {
#pragma unroll
                for (int s = 0; s < RINGM; ++s) {
                    // -DLP_ABL=n (timing-only builds, wrong results): 1 = no weight loads, 2 = no LDS operand loads, 4 = no MFMAs in this loop
#if !defined(LP_ABL) || !(LP_ABL & 1)
                    load_a(A[(s + RINGM - 1) % RINGM], a.W3f[l], min(it + s + RINGM - 1, NGB - 1));
#endif
                    const int nx = min(it + s + 1, NGB - 1);
#if !defined(LP_ABL) || !(LP_ABL & 2)
                    load_b(Bv[(s + 1) & 1], ut, nx & 15, nx >> 4);
#endif
                    __builtin_amdgcn_sched_barrier(0);
#if !defined(LP_ABL) || !(LP_ABL & 4)
                    if (it + s < NGB) mma_group(A[s], Bv[s & 1]);
#else
                    if (it + s < NGB) { acc[0][0][0] += __builtin_bit_cast(float, A[s][0][0][0]) + __builtin_bit_cast(float, Bv[s & 1][0][0][0]); }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        LPSTAMP(2);
#pragma unroll
        for (int s = 0; s < RINGM - 1; ++s) load_ao(A[s], a.Wof[l], min(s, NGC - 1));
        {   // gate -> z^T (own buffer: no barrier between the conv and the gate)
            const float* b3 = a.b3[l];
            const int ln = opaque(lane);
            float bg[MT][8], bf[MT][8];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int mg = (w * MT + i) * 32 + acc_row(r, ln);
                    bg[i][r] = ldg(b3, (unsigned)mg);
                    bf[i][r] = ldg(b3, (unsigned)(mg + 16));
                }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {      // registers r, r+1 = adjacent channels (even first)
                        const float z0 = cmtts_gate(acc[i][j][r] + bg[i][r], acc[i][j][r + 8] + bf[i][r]);
                        const float z1 = cmtts_gate(acc[i][j][r + 1] + bg[i][r + 1], acc[i][j][r + 9] + bf[i][r + 1]);
                        const int ch = (w * MT + i) * 16 + acc_row(r, ln);
                        put2<MODE>(zt, (j * 32 + (ln & 31)) * RS + ch, z0, z1, true, IMG);
                    }
        }
        LPSTAMP(3);
        // (3) per-wave flags instead of a barrier: the projection's K loop (ordered by source wave: two 16-channel k-groups per wave) acquires
        // the flag of the block it is about to read; a wave writes the next u^T only after it has consumed all eight blocks
        if (lane == 0) __hip_atomic_store(zflag + w, l + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned zready = 0;
        auto need_z = [&](int v) {
            if (v >= NW || ((zready >> v) & 1u)) return;
            unsigned spins = 0;
            for (;;) {
                const int f = __hip_atomic_load(zflag + (lane & (NW - 1)), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                zready = (unsigned)__ballot(f > l) & ((1u << NW) - 1u);
                if ((zready >> v) & 1u) break;
                if (++spins > SPIN_LIMIT) {      // cannot happen; bounded like every wait of this kernel
                    if (lane == 0 && a.tmo) *(volatile unsigned*)a.tmo = 1u;
                    zready = (1u << NW) - 1u;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        };
        LPSTAMP(4);

        // =========================================================== phase C: output projection, 16 k-groups
        {
            zero_acc();
            u32x4 Bv[2][NS][NT];
            need_z(0);
            load_b(Bv[0], zt, 0, 0);
#pragma unroll 1
            for (int it = 0; it < NGC; it += RINGM) {
#pragma unroll
                for (int s = 0; s < RINGM; ++s) {
                    load_ao(A[(s + RINGM - 1) % RINGM], a.Wof[l], min(it + s + RINGM - 1, NGC - 1));
                    if (((it + s + 1) & 1) == 0) need_z((it + s + 1) >> 1);      // k-group it + s + 1 opens the next wave's channels
                    load_b(Bv[(s + 1) & 1], zt, min(it + s + 1, NGC - 1), 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (it + s < NGC) mma_group(A[s], Bv[s & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        LPSTAMP(5);
        // ---- epilogue in fp32 registers: tile 0: x' = (o[:C] + (x + d)) / sqrt(2); tile 1: skip (+)= o[C:]
        {
            const float* bo = a.bo[l];
            const float* dl = dv_b + (long)l * C;
            const int ln = opaque(lane);
            float bor[MT][16], ddr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bor[0][r] = ldg(bo, (unsigned)(mrow0 + acc_row(r, ln)));
                bor[1][r] = ldg(bo, (unsigned)(C + mrow0 + acc_row(r, ln)));
                ddr[r] = ldg(dl, (unsigned)(mrow0 + acc_row(r, ln)));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float o = acc[0][j][r] + bor[0][r];
                    st[0][j][r] = (o + (st[0][j][r] + ddr[r])) * CMTTS_RSQRT2;
                    const float os = acc[1][j][r] + bor[1][r];
                    st[1][j][r] = l > 0 ? os + st[1][j][r] : os;
                }
        }
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        LPSTAMP(6);
        const float* dpn = dp_b + (long)(l + 1) * C;
        const unsigned tag = (unsigned)l + 1;
        unsigned long long* hbase = a.halo + ((((long)(l & 1) * a.B + b) * a.tiles) * 2) * C;
        {   // edge columns of x' (fp32) to the neighbouring tiles
            const int ln = opaque(lane), c31 = ln & 31;
            // through 64 floats of the wave's own LDS scratch (behind the images): one coalesced 64-lane granule store instead of 32
            // two-lane ones (denoiser_persist.hip, round 5)
            float* edge = reinterpret_cast<float*>(lds16 + (MODE == 3 ? 2 : 1) * IMG) + w * 64;
            if (c31 == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) edge[acc_row(r, ln)] = st[0][0][r];
            }
            if (c31 == 31) {
#pragma unroll
                for (int r = 0; r < 16; ++r) edge[32 + acc_row(r, ln)] = st[0][NT - 1][r];
            }
            store_granule(hbase + ((long)tile * 2 + (ln >> 5)) * C + mrow0 + (ln & 31), tag, edge[ln]);
        }
        // ---- halo columns of the next layer's u^T.  Every wave fetches the two halo entries of ITS OWN 32 channels (lanes 0-31: left
        // halo frame t0 - 1, lanes 32-63: right halo frame t0 + FN): one cp value and one granule per lane.  (Until round 4 the last two
        // waves fetched all 256 channels of one side each, four granules per lane, after their own u^T rows: the phase stamps showed
        // them 5 k cycles behind the other six waves at the layer barrier.)  The cp value and a first look at the granule are requested
        // here, before the wave's own rows; the tag is checked after them.
        const int hside = opaque(lane) >> 5, hm = mrow0 + (opaque(lane) & 31);
        const int hth = hside ? t0 + FN : t0 - 1;
        const bool hinside = hth >= 0 && hth < T;
        const unsigned long long* hg = hbase + ((long)(hinside ? (hside ? tile + 1 : tile - 1) : tile) * 2 + (hside ? 0 : 1)) * C + hm;
        const float hcp = (cp_b + (long)(l + 1) * C * T)[(unsigned)(hm * T + min(max(hth, 0), T - 1))];
        unsigned long long hv = __hip_atomic_load((gu64*)hg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {   // next layer's u^T, this wave's 32 channels: cvt(cp + (x' + dp)), channel pairs packed
            const float* cpn = cp_b + (long)(l + 1) * C * T;
            const int ln = opaque(lane), c31 = ln & 31;
            f32x16 cpc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t_c = min(t0 + j * 32 + c31, T - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) cpc[j][r] = ldg(cpn, (unsigned)((mrow0 + acc_row(r, ln)) * T + t_c));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = t0 + j * 32 + c31;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int m = mrow0 + acc_row(r, ln);
                    const float u0 = cpc[j][r] + (st[0][j][r] + ldg(dpn, (unsigned)m));
                    const float u1 = cpc[j][r + 1] + (st[0][j][r + 1] + ldg(dpn, (unsigned)(m + 1)));
                    put2<MODE>(ut, (1 + j * 32 + c31) * RS + m, u0, u1, t < T, IMG);
                    note_range<MODE>(ovf, u0, u1, t < T);
                }
            }
        }
        {   // the halo entries: wait for the neighbours' tags (lanes without a neighbour frame never wait)
            if (!gave_up) {
                unsigned spins = 0;
                while (!__all(!hinside || (unsigned)(hv >> 32) == tag)) {
                    if (++spins > SPIN_LIMIT) {
                        if (lane == 0 && a.tmo) *(volatile unsigned*)a.tmo = 1u;
                        gave_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                    hv = __hip_atomic_load((gu64*)hg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // after a timeout the halo column is poisoned: the utterance's mel comes out NaN (spreading one tile per layer) instead of
            // plausible-but-wrong, and cmtts_poll_error() reports the timeout
            const float xh = gave_up ? __builtin_nanf("") : (hinside ? __uint_as_float((unsigned)hv) : 0.f);
            const float uh = hcp + (xh + dpn[hm]);
            put1<MODE>(ut, (hside ? FN + 1 : 0) * RS + hm, uh, hinside, IMG);
            note_range<MODE>(ovf, uh, 0.f, hinside);
        }
        LPSTAMP(7);
    }

    if (MODE >= 2 && ovf && a.tmo && *(volatile unsigned*)a.tmo == 0u) *(volatile unsigned*)a.tmo = 3u;
    if (a.tail) {
        // skip head + post-scaling in fp32 (persist_tail.h).  Its two fp32 buffers overlay u^T / z^T: wait until every
        // wave has left the last output projection.
        __syncthreads();
        float* f32lds = reinterpret_cast<float*>(lds16);
        persist_tail::run(a, f32lds, f32lds + C * persist_tail::PT_LD, st[1], w, lane, b, t0, T, a.xold, a.noise, a.out, T);
    } else {   // the skip sum leaves the chip once
        float* skip = a.skip + (long)b * C * T;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = t0 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t < T) skip[(unsigned)((mrow0 + acc_row(r, lane)) * T + t)] = st[1][j][r];
        }
    }
}

template <int MODE>
int launch_mode(const PersistArgs& a, int tiles, int max_blocks, hipStream_t stream) {
    static bool attr_set = false;
    // the fp32 tail overlays two [256][68] float buffers on the 16-bit u^T / z^T images
    // (+ 64 floats per wave behind the images: the edge-column scratch of the granule store)
    const size_t lds16b = (size_t)(MODE == 3 ? 2 : 1) * (2 * FN + 2) * RS * sizeof(unsigned short) + NW * 64 * sizeof(float) + NW * sizeof(int), ldstail = (size_t)2 * C * persist_tail::PT_LD * sizeof(float);
    const size_t lds = a.tail && ldstail > lds16b ? ldstail : lds16b;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(denoiser_persist_lp_kernel<MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ldstail > lds16b ? ldstail : lds16b)) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const int B = a.B;
    const int per_launch = max_blocks / tiles;
    const int nchunks = (B + per_launch - 1) / per_launch;
    const int bc = (B + nchunks - 1) / nchunks;
    for (int b0 = 0; b0 < B; b0 += bc) {
        PersistArgs c = a;
#ifdef LP_STAMP
        c.dbg = cmtts_persist_get_debug();
#endif
        const int nb = B - b0 < bc ? B - b0 : bc;
        c.x0 = a.x0 + (long)b0 * C * a.T;
        c.cp = a.cp + (long)b0 * a.cp_bstride;
        c.dp = a.dp + (long)b0 * a.vec_stride;
        c.d = a.d + (long)b0 * a.vec_stride;
        c.skip = a.skip + (long)b0 * C * a.T;
        c.halo = a.halo + (long)b0 * tiles * 2 * C;
        if (a.tail) {
            const long off = (long)b0 * a.T * a.n_mels;
            c.xold = a.xold ? a.xold + off : nullptr;
            c.noise = a.noise ? a.noise + off : nullptr;
            c.out = a.out + off;
        }
        if (cmtts_persist_cooperative(16 + MODE, tiles, nb)) {
            void* params[] = {(void*)&c};
            if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(denoiser_persist_lp_kernel<MODE>), dim3(tiles, nb),
                                           dim3(64 * NW), params, (unsigned)lds, stream) != hipSuccess) return -3;
            cmtts_persist_validated(16 + MODE, tiles, nb);
        } else hipLaunchKernelGGL(denoiser_persist_lp_kernel<MODE>, dim3(tiles, nb), dim3(64 * NW), lds, stream, c);
        if (hipGetLastError() != hipSuccess) return -3;
    }
    return 0;
}

}  // namespace

// mode 1 = bf16, 2 = fp16, 3 = fp16x3 (hi fragments followed by lo fragments); a->W3f / a->Wof point to the 16-bit fragment-order weights of each layer.  Return
// codes and the residency rule are those of cmtts_launch_denoiser_persist.
extern "C" int cmtts_launch_denoiser_persist_lp(const PersistArgs* a_in, int mode, int max_blocks, int force, void* stream_) {
    PersistArgs a = *a_in;
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles = (a.T + FN - 1) / FN;
    if (a.NL < 1 || a.NL > PERSIST_MAX_LAYERS || tiles > max_blocks || (long)C * a.T >= (1L << 30) || mode < 1 || mode > 3) return -2;
    if (!force && (long)tiles * a.B * 2 <= (long)max_blocks) return -2;
    a.tiles = tiles;
    a.dbg = nullptr;
    if (!a.halo_zeroed && hipMemsetAsync(a.halo, 0, cmtts_persist_halo_bytes(a.B, a.T), stream) != hipSuccess) return -3;
    if (mode == 3) return launch_mode<3>(a, tiles, max_blocks, stream);
    return mode == 1 ? launch_mode<1>(a, tiles, max_blocks, stream) : launch_mode<2>(a, tiles, max_blocks, stream);
}
