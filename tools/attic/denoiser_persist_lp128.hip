// Persistent denoiser stack with 16-bit MFMA operands, 128-FRAME TILES (round 6) — denoiser_persist_lp.hip's kernel
// for batches with more 64-frame tiles than the chip has CUs (BASELINE.json configs[2]: B = 64 x 512 frames = 512 tiles; the north-star
// shape 32 x 1024 likewise).
//
// Why: at 16-bit MFMA rates the 64-frame kernel is bound by what a CU can INGEST — a layer's 786 KB of conv weights + 262 KB of projection
// weights must reach every workgroup (44 B/clk per CU: ~24 k cycles where the MFMAs of a 64-frame tile need 16 k), and a batch of 512 tiles
// pays that twice (two rounds of 256 workgroups).  Here a workgroup owns TWO 64-frame sub-tiles and every weight fragment, once in registers,
// multiplies both: four n-tiles per wave instead of two — the weight bytes per frame halve, the loops become MFMA-bound (33 k cycles per
// 128 frames against 24 k of delivery), and the batch runs in ONE round.
//
// What had to give: the register file does not hold the state twice (x + skip sum of 128 frames = 128 registers next to 128 accumulator
// registers), so BOTH state tensors live in memory between layers — `xst`, L2 / Infinity-Cache resident, in the accumulator layout (16-byte
// loads / stores of the lane's own elements: program order, no fences) — and the epilogue, the next layer's u^T rows and the edge columns
// are formed n-tile by n-tile as the state streams through registers.  Everything else is the 64-frame kernel: u^T / z^T images in LDS
// (134 KB for 128 + 2 frames), weights L2 -> VGPR through a register ring, per-wave z flags instead of the mid barrier, granule halo
// exchange, in-kernel tail (persist_tail.h, once per sub-tile).  Same arithmetic per element in the same order: BITWISE equal to the
// 64-frame kernel and to the per-layer 16-bit kernels (tests/test_gpu_parity.py::test_persistent_denoiser_lp128_bitwise).
// bf16 / fp16 (MODE 1 / 2); fp16x3 keeps the 64-frame kernel (two operand sets do not fit next to eight accumulators).
#include <hip/hip_runtime.h>
#include "cvt16.h"
#include "gate.h"
#include "persist_args.h"
#include "persist_tail.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) unsigned long long gu64;

// No floating-point contraction in this file: x' = (o + (x + d)) * 2^-1/2 feeds u = cp + (x' + dp) in the same loop body here, and the compiler fused the
// multiply into that add (one rounding less than the 64-frame kernel and the per-layer kernels, where x' passes through registers of another block / through
// memory): 1 ulp of fp32 that flips a bf16 rounding of u once in ~1e5 elements and then spreads a frame per layer (found by test_persistent_denoiser_lp128_bitwise).
#pragma clang fp contract(off)

namespace {

constexpr int C = 256;
constexpr int NW = 8;           // waves per workgroup, each owning 2 m-tiles x 4 n-tiles
constexpr int MT = 2;
constexpr int RING = 4;         // 16-channel k-groups of weights in flight
constexpr int FN = 128;
constexpr int NT = FN / 32;
constexpr int RS = 260;         // 16-bit elements per LDS row (520 B)
constexpr int IMG = (2 * FN + 2) * RS;
constexpr int STATE = NW * NT * 4 * 64 * 4;      // floats of one state tensor of a tile: [wave][n-tile][quad of registers][lane][4]
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
// fp16: an operand beyond the fp16 range cannot be represented: recorded, reported through the pinned error word (code 3) — denoiser_persist_lp.hip
template <int MODE>
__device__ __forceinline__ void note_range(bool& ovf, float v0, float v1, bool valid) {
    if (MODE >= 2) ovf |= valid && !(fabsf(v0) <= 65504.0f && fabsf(v1) <= 65504.0f);
}
template <int MODE>
__device__ __forceinline__ void put2(unsigned short* img, int off, float v0, float v1, bool valid) {
    *reinterpret_cast<unsigned*>(img + off) = valid ? pack16<MODE>(v0, v1) : 0u;
}
template <int MODE>
__device__ __forceinline__ void put1(unsigned short* img, int off, float v, bool valid) {
    img[off] = valid ? (unsigned short)pack16<MODE>(v, 0.f) : (unsigned short)0;
}

template <int MODE>
__global__ __launch_bounds__(64 * NW, 1) void denoiser_persist_lp128_kernel(const PersistArgs a) {
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
    const float* x0_b = a.x0 + (long)b * C * T;
    const int mrow0 = w * 32;                         // this wave's 32 rows of x and of the skip sum
    // the two state tensors of this tile, [x | skip], each [wave][n-tile][4][64 lanes] float4: element e of vector (j, q) at lane l = accumulator
    // register 4 q + e of n-tile j (row mrow0 + acc_row(4 q + e, l), frame t0 + 32 j + (l & 31))
    float* st_x = a.xst + ((long)b * a.tiles + tile) * (2 * STATE);
    float* st_s = st_x + STATE;
    bool ovf = false;
    int* zflag = reinterpret_cast<int*>(lds16 + IMG) + NW * 64;
    if (tid < NW) zflag[tid] = 0;

    // ---- layer-0 staging: u^T[j][m] = cvt(cp + (x + dp)); lane = frame (two passes of 64), waves over channel pairs
#pragma unroll 1
    for (int h = 0; h < FN / 64; ++h) {
        const int fr = h * 64 + lane;
        const int t = t0 + fr;
        const int t_c = min(t, T - 1);
#pragma unroll 1
        for (int i = 0; i < C / (2 * NW); i += 4) {
            float x0[4], x1[4], c0[4], c1[4], d0[4], d1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = 2 * (w + NW * (i + q));
                x0[q] = x0_b[(unsigned)(m * T + t_c)];
                x1[q] = x0_b[(unsigned)((m + 1) * T + t_c)];
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
                put2<MODE>(ut, (1 + fr) * RS + m, u0, u1, t < T);
                note_range<MODE>(ovf, u0, u1, t < T);
            }
        }
    }
    {
        const int m = tid & (C - 1);
        const bool right = tid >= C;
        const int th = right ? t0 + FN : t0 - 1;
        const int thc = min(max(th, 0), T - 1);
        const float uh = cp_b[(unsigned)(m * T + thc)] + (x0_b[(unsigned)(m * T + thc)] + dp_b[m]);
        put1<MODE>(ut, (right ? FN + 1 : 0) * RS + m, uh, th >= 0 && th < T);
        note_range<MODE>(ovf, uh, 0.f, th >= 0 && th < T);
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
    // weights [group][m-tile (16)][lane][8 x 16-bit]: k = 3 conv tiles 2w, 2w + 1; projection tiles w (residual half), NW + w (skip half)
    auto load_a = [&](u32x4 (&dst)[MT], const void* wfrag, int group) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
            dst[i] = *(reinterpret_cast<const u32x4*>(wfrag) + ((long)group * (2 * C / 32) + w * MT + i) * 64 + lane);
    };
    auto load_ao = [&](u32x4 (&dst)[MT], const void* wfrag, int group) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
            dst[i] = *(reinterpret_cast<const u32x4*>(wfrag) + ((long)group * (2 * C / 32) + i * NW + w) * 64 + lane);
    };
    auto load_b = [&](u32x4 (&dst)[NT], const unsigned short* src, int kg, int row_off) {      // 16 bytes = 8 k-values of one frame
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const unsigned short* p = src + (j * 32 + l31 + row_off) * RS + kg * 16 + khalf * 8;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(p);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(p + 4);
            dst[j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };
    auto mma_group = [&](const u32x4 (&af)[MT], const u32x4 (&bv)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mma16<MODE>(af[i], bv[j], acc[i][j]);
    };

    bool gave_up = false;
    // -DLP_STAMP (timing-only builds loaded through CMTTS_LIB, tools/lp128_phases.py): cycle stamps of layer NL / 2 per wave
#ifdef LP_STAMP
    const int bid_dbg = blockIdx.x + gridDim.x * blockIdx.y;
#define LPSTAMP(slot) do { if (a.dbg && l == a.NL / 2 && lane == 0) a.dbg[((long)bid_dbg * NW + w) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LPSTAMP(slot) do { } while (0)
#endif
    for (int l = 0; l < a.NL; ++l) {
        const bool more = l + 1 < a.NL;
        constexpr int NGB = 3 * (C / 16);        // k = 3 conv: group = tap * 16 + k-group
        constexpr int NGC = C / 16;
        u32x4 A[RING][MT];
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) load_a(A[s], a.W3f[l], s);         // the weight stream does not depend on u
        LPSTAMP(0);
        __syncthreads();   // (1) u^T of layer l complete
        LPSTAMP(1);
        if (more) {        // pull the next layer's cp tile towards L2: one dword per 128-B line
            const float* cpn = cp_b + (long)(l + 1) * C * T;
            const int tl = opaque(tid);
            const float warm = cpn[(unsigned)((tl >> 1) * T + min(t0 + (tl & 1) * 32, T - 1))];
            const float warm2 = cpn[(unsigned)((tl >> 1) * T + min(t0 + 64 + (tl & 1) * 32, T - 1))];
            asm volatile("" ::"v"(warm), "v"(warm2));
        }

        // =========================================================== phase B: gated k = 3 conv, 48 k-groups
        {
            zero_acc();
            u32x4 Bv[2][NT];
            load_b(Bv[0], ut, 0, 0);
#pragma unroll 1
            for (int it = 0; it < NGB; it += RING) {
#pragma unroll
                for (int s = 0; s < RING; ++s) {
                    load_a(A[(s + RING - 1) % RING], a.W3f[l], min(it + s + RING - 1, NGB - 1));
                    const int nx = min(it + s + 1, NGB - 1);
                    load_b(Bv[(s + 1) & 1], ut, nx & 15, nx >> 4);
                    __builtin_amdgcn_sched_barrier(0);
                    if (it + s < NGB) mma_group(A[s], Bv[s & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        LPSTAMP(2);
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) load_ao(A[s], a.Wof[l], min(s, NGC - 1));
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
                    for (int r = 0; r < 8; r += 2) {      // registers r, r + 1 = adjacent channels (even first)
                        const float z0 = cmtts_gate(acc[i][j][r] + bg[i][r], acc[i][j][r + 8] + bf[i][r]);
                        const float z1 = cmtts_gate(acc[i][j][r + 1] + bg[i][r + 1], acc[i][j][r + 9] + bf[i][r + 1]);
                        const int ch = (w * MT + i) * 16 + acc_row(r, ln);
                        put2<MODE>(zt, (j * 32 + (ln & 31)) * RS + ch, z0, z1, true);
                    }
        }
        LPSTAMP(3);
        // (3) per-wave flags instead of a barrier (denoiser_persist_lp.hip): the projection's K loop acquires the flag of the block it is about to read
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
        // =========================================================== phase C: output projection (16 k-groups) + epilogue, as TWO loops of four accumulators:
        // the residual half (rows 32 w .. of o[:C]) with the x state and the next layer's cp tile requested IN FRONT of it — they land under its MFMAs — then
        // x' / the u^T rows / the edge columns; the skip half (rows 32 w .. of o[C:]) with the skip state requested in front of it, then the skip sum.  (As one
        // loop of eight accumulators the state had to be fetched n-tile by n-tile behind it: four exposed round trips to the Infinity Cache, 32 k cycles per
        // layer.)  z^T is read twice; every accumulator's chain is unchanged.
        const float* dpn = dp_b + (long)(l + 1) * C;
        const unsigned tag = (unsigned)l + 1;
        unsigned long long* hbase = a.halo + ((((long)(l & 1) * a.B + b) * a.tiles) * 2) * C;
        float* edge = reinterpret_cast<float*>(lds16 + IMG) + w * 64;
        {
            const int ln = opaque(lane), c31 = ln & 31;
            f32x4* px = reinterpret_cast<f32x4*>(st_x) + (w * NT * 4) * 64 + ln;
            f32x4* ps = reinterpret_cast<f32x4*>(st_s) + (w * NT * 4) * 64 + ln;
            f32x16 (&accp)[NT] = acc[0];
            auto proj_half = [&](auto half) {
                constexpr int I = decltype(half)::value;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accp[j][r] = 0.f;
                u32x4 Bv[2][NT];
                if (I == 0) need_z(0);
                load_b(Bv[0], zt, 0, 0);
#pragma unroll 1
                for (int it = 0; it < NGC; it += RING) {
#pragma unroll
                    for (int s = 0; s < RING; ++s) {
                        A[(s + RING - 1) % RING][I] = *(reinterpret_cast<const u32x4*>(a.Wof[l]) + ((long)min(it + s + RING - 1, NGC - 1) * (2 * C / 32) + I * NW + w) * 64 + lane);
                        if (I == 0 && ((it + s + 1) & 1) == 0) need_z((it + s + 1) >> 1);      // k-group it + s + 1 opens the next wave's channels
                        load_b(Bv[(s + 1) & 1], zt, min(it + s + 1, NGC - 1), 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (it + s < NGC) {
#pragma unroll
                            for (int j = 0; j < NT; ++j) accp[j] = mma16<MODE>(A[s][I], Bv[s & 1][j], accp[j]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            // ---- residual half
            const float* bo = a.bo[l];
            const float* dl = dv_b + (long)l * C;
            const float* cpn = cp_b + (long)(l + 1) * C * T;
            {
                // (the x state — Infinity-Cache latency — is requested in front of the loop; cp, L2-warm since the layer's start, and the per-row vectors behind it,
                // all n-tiles in one batch: with them in flight across the loop as well the wave needs 300 registers)
                float bor[16], ddr[16], dpr[16];
                f32x16 xo[NT], cpc[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int t_c = min(t0 + j * 32 + c31, T - 1);
                    if (l == 0) {      // x enters in the public [C][T] layout
#pragma unroll
                        for (int r = 0; r < 16; ++r) xo[j][r] = ldg(x0_b, (unsigned)((mrow0 + acc_row(r, ln)) * T + t_c));
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 vx = px[(j * 4 + q) * 64];
#pragma unroll
                            for (int e = 0; e < 4; ++e) xo[j][4 * q + e] = vx[e];
                        }
                    }
                }
                proj_half(std::integral_constant<int, 0>{});
                LPSTAMP(5);
                if (more) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        bor[r] = ldg(bo, (unsigned)(mrow0 + acc_row(r, ln)));
                        ddr[r] = ldg(dl, (unsigned)(mrow0 + acc_row(r, ln)));
                    }
                    // pass 1: x' (frees the accumulators), back to `xst`; the tile's edge columns
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float o = accp[j][r] + bor[r];
                            xo[j][r] = (o + (xo[j][r] + ddr[r])) * CMTTS_RSQRT2;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 vx;
#pragma unroll
                            for (int e = 0; e < 4; ++e) vx[e] = xo[j][4 * q + e];
                            px[(j * 4 + q) * 64] = vx;
                        }
                        if (j == 0 && c31 == 0) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) edge[acc_row(r, ln)] = xo[j][r];
                        }
                        if (j == NT - 1 && c31 == 31) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) edge[32 + acc_row(r, ln)] = xo[j][r];
                        }
                    }
                    // the edge columns leave now: the neighbours get the whole skip-half loop to receive them
                    store_granule(hbase + ((long)tile * 2 + (ln >> 5)) * C + mrow0 + (ln & 31), tag, edge[ln]);
                    // pass 2: the next layer's u^T rows = cvt(cp + (x' + dp)): cp (L2-warm) for all n-tiles in one batch
#pragma unroll
                    for (int r = 0; r < 16; ++r) dpr[r] = ldg(dpn, (unsigned)(mrow0 + acc_row(r, ln)));
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
                            const float u0 = cpc[j][r] + (xo[j][r] + dpr[r]);
                            const float u1 = cpc[j][r + 1] + (xo[j][r + 1] + dpr[r + 1]);
                            put2<MODE>(ut, (1 + j * 32 + c31) * RS + m, u0, u1, t < T);
                            note_range<MODE>(ovf, u0, u1, t < T);
                        }
                    }
                }
            }
            // ---- skip half
            {
                float bor[16];
                f32x16 so[NT];
#pragma unroll
                for (int r = 0; r < 16; ++r) bor[r] = ldg(bo, (unsigned)(C + mrow0 + acc_row(r, ln)));
                if (l > 0) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 vs = ps[(j * 4 + q) * 64];
#pragma unroll
                            for (int e = 0; e < 4; ++e) so[j][4 * q + e] = vs[e];
                        }
                }
                proj_half(std::integral_constant<int, 1>{});
#pragma unroll
                for (int j = 0; j < NT; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float os = accp[j][r] + bor[r];
                        so[j][r] = l > 0 ? os + so[j][r] : os;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 vs;
#pragma unroll
                        for (int e = 0; e < 4; ++e) vs[e] = so[j][4 * q + e];
                        ps[(j * 4 + q) * 64] = vs;
                    }
                }
            }
        }
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        LPSTAMP(6);
        // ---- halo columns of the next layer's u^T: every wave fetches the two halo entries of ITS OWN 32 channels (lanes 0-31: frame t0 - 1, lanes 32-63: frame t0 + FN)
        const int hside = opaque(lane) >> 5, hm = mrow0 + (opaque(lane) & 31);
        const int hth = hside ? t0 + FN : t0 - 1;
        const bool hinside = hth >= 0 && hth < T;
        const unsigned long long* hg = hbase + ((long)(hinside ? (hside ? tile + 1 : tile - 1) : tile) * 2 + (hside ? 0 : 1)) * C + hm;
        const float hcp = (cp_b + (long)(l + 1) * C * T)[(unsigned)(hm * T + min(max(hth, 0), T - 1))];
        unsigned long long hv = __hip_atomic_load((gu64*)hg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
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
            const float xh = gave_up ? __builtin_nanf("") : (hinside ? __uint_as_float((unsigned)hv) : 0.f);
            const float uh = hcp + (xh + dpn[hm]);
            put1<MODE>(ut, (hside ? FN + 1 : 0) * RS + hm, uh, hinside);
            note_range<MODE>(ovf, uh, 0.f, hinside);
        }
        LPSTAMP(7);
    }

    if (MODE >= 2 && ovf && a.tmo && *(volatile unsigned*)a.tmo == 0u) *(volatile unsigned*)a.tmo = 3u;
    // the skip sum of this wave's rows sits in `st_s` (its own stores: program order)
    const f32x4* ps = reinterpret_cast<const f32x4*>(st_s) + (w * NT * 4) * 64 + opaque(lane);
    if (a.tail) {
        // skip head + post-scaling in fp32 (persist_tail.h), one 64-frame sub-tile at a time; its two fp32 buffers overlay u^T / z^T
        float* f32lds = reinterpret_cast<float*>(lds16);
#pragma unroll 1
        for (int h = 0; h < FN / 64; ++h) {
            f32x16 sk[persist_tail::PT_NT];
#pragma unroll
            for (int j = 0; j < persist_tail::PT_NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 vs = ps[((2 * h + j) * 4 + q) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) sk[j][4 * q + e] = vs[e];
                }
            __syncthreads();      // every wave has left the last output projection / the previous sub-tile's tail
            if (t0 + 64 * h < T)
                persist_tail::run(a, f32lds, f32lds + C * persist_tail::PT_LD, sk, w, lane, b, t0 + 64 * h, T, a.xold, a.noise, a.out, T);
        }
    } else {   // the skip sum leaves the chip once, in the public layout
        float* skip = a.skip + (long)b * C * T;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = t0 + j * 32 + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 vs = ps[(j * 4 + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t < T) skip[(unsigned)((mrow0 + acc_row(4 * q + e, lane)) * T + t)] = vs[e];
            }
        }
    }
}

template <int MODE>
int launch_mode(const PersistArgs& a, int tiles, int max_blocks, hipStream_t stream) {
    static bool attr_set = false;
    const size_t lds16b = (size_t)IMG * sizeof(unsigned short) + NW * 64 * sizeof(float) + NW * sizeof(int), ldstail = (size_t)2 * C * persist_tail::PT_LD * sizeof(float);
    const size_t lds = ldstail > lds16b ? ldstail : lds16b;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(denoiser_persist_lp128_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    const int B = a.B;
    const int per_launch = max_blocks / tiles;
    const int nchunks = (B + per_launch - 1) / per_launch;
    const int bc = (B + nchunks - 1) / nchunks;
    for (int b0 = 0; b0 < B; b0 += bc) {
        PersistArgs c = a;
        const int nb = B - b0 < bc ? B - b0 : bc;
        c.x0 = a.x0 + (long)b0 * C * a.T;
        c.cp = a.cp + (long)b0 * a.cp_bstride;
        c.dp = a.dp + (long)b0 * a.vec_stride;
        c.d = a.d + (long)b0 * a.vec_stride;
        c.skip = a.skip + (long)b0 * C * a.T;
        c.xst = a.xst + (long)b0 * tiles * (2 * STATE);
        c.halo = a.halo + (long)b0 * tiles * 2 * C;
        if (a.tail) {
            const long off = (long)b0 * a.T * a.n_mels;
            c.xold = a.xold ? a.xold + off : nullptr;
            c.noise = a.noise ? a.noise + off : nullptr;
            c.out = a.out + off;
        }
        if (cmtts_persist_cooperative(24 + MODE, tiles, nb)) {
            void* params[] = {(void*)&c};
            if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(denoiser_persist_lp128_kernel<MODE>), dim3(tiles, nb),
                                           dim3(64 * NW), params, (unsigned)lds, stream) != hipSuccess) return -3;
            cmtts_persist_validated(24 + MODE, tiles, nb);
        } else hipLaunchKernelGGL(denoiser_persist_lp128_kernel<MODE>, dim3(tiles, nb), dim3(64 * NW), lds, stream, c);
        if (hipGetLastError() != hipSuccess) return -3;
    }
    return 0;
}

}  // namespace

// The 128-frame-tile instance (mode 1 = bf16, 2 = fp16).  a->xst must hold cmtts_persist_state_floats(B, T) floats, a->halo the granules of the (B, T) batch, cleared
// by the caller (a->halo_zeroed) or here.  Returns 0, -2 (shape / mode not covered: use cmtts_launch_denoiser_persist_lp) or -3.
extern "C" int cmtts_launch_denoiser_persist_lp128(const PersistArgs* a_in, int mode, int max_blocks, void* stream_) {
    PersistArgs a = *a_in;
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles = (a.T + FN - 1) / FN;
    if (a.NL < 1 || a.NL > PERSIST_MAX_LAYERS || tiles > max_blocks || (long)C * a.T >= (1L << 30) || (mode != 1 && mode != 2) || !a.xst) return -2;
    if (a.tail && a.n_mels > 128) return -2;
    a.tiles = tiles;
#ifdef LP_STAMP
    a.dbg = cmtts_persist_get_debug();
#else
    a.dbg = nullptr;
#endif
    if (!a.halo_zeroed && hipMemsetAsync(a.halo, 0, cmtts_persist_halo_bytes(a.B, a.T), stream) != hipSuccess) return -3;
    return mode == 1 ? launch_mode<1>(a, tiles, max_blocks, stream) : launch_mode<2>(a, tiles, max_blocks, stream);
}
