// The K loop of the 16-bit HiFi-GAN kernels (resblock_pair16.hip, conv_xl16.hip): fragment types, the 16-bit MFMA, the hand-issued
// weight ring (conv_loop16) and its two-m-tile variant.  Included by both translation units; everything lives in an anonymous namespace.
#pragma once
#include <hip/hip_runtime.h>
#include "resblock_pair.h"
#include "cvt16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int N1 = 256;          // columns of xt per workgroup
#ifndef CL16_RING
#define CL16_RING 8
#endif
constexpr int RING = CL16_RING;  // A fragments in flight per wave (by hand, see conv_loop16): an L2 hit takes ~0.7 us = several groups of 2-4 MFMAs
constexpr int R1MAX = 25;
// padding of an LDS image row [column][C + CL16_PAD] in 16-bit elements: 4 = a B fragment is two conflict-free ds_read_b64; 8 (rows a multiple of
// 16 bytes) = one ds_read_b128.  Every kernel that includes this header lays its images out with the same constant.
#ifndef CL16_PAD
#define CL16_PAD 4
#endif

template <int MODE>
__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if (MODE == 1)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int LO, int N, int SEG, class F>
__device__ __forceinline__ void seg_loop(F& body) {
    constexpr int HI = LO + SEG < N ? LO + SEG : N;
#pragma unroll
    for (int it = LO; it < HI; ++it) body(it);
    if constexpr (HI < N) seg_loop<HI, N, SEG>(body);
}

// acc = W * src over K = C * KT in conv_mfma16.hip's order: 32-channel chunk -> tap -> k-group of 16 within the chunk.
// wfrag: [tap][C/16][C/32][64 lanes] u32x4 (A fragments);  src: LDS [cols][RS] 16-bit, output column c reads row c + tap*dil.
template <int C, int KT, int NT, int MODE>
__device__ __forceinline__ void conv_loop16(f32x16 (&acc)[NT], const u32x4* __restrict__ wfrag, const unsigned short* __restrict__ src,
                                            int dil, int mt, int col0, int lane) {
    constexpr int RS = C + CL16_PAD;
    constexpr int G = C / 16, MTn = C / 32;
    constexpr int NG = G * KT;                      // MFMA k-groups: (chunk, tap, k-group-in-chunk)
    const int l31 = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // the K loop is fully unrolled (NG <= 44 groups): chunk / tap / k-group and the ring slots are compile-time, the only
    // runtime term of an operand address is tap * dil
    auto grp = [&](int it, int& chunk, int& tap, int& kgl) {
        chunk = it / (2 * KT);
        const int rr = it - chunk * (2 * KT);
        tap = rr >> 1;
        kgl = rr & 1;
    };
    const unsigned short* bl = src + (col0 + l31) * RS + khalf * 8;
    auto load_b = [&](u32x4 (&dst)[NT], int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const unsigned short* p = bl + (tap * dil) * RS + chunk * 32 + kgl * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (CL16_PAD == 8) {
                dst[j] = *reinterpret_cast<const u32x4*>(p + j * 32 * RS);
                continue;
            }
            const u32x2 lo = *reinterpret_cast<const u32x2*>(p + j * 32 * RS);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(p + j * 32 * RS + 4);
            dst[j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };
    // The weight stream is issued and awaited by hand (inline asm): left to the compiler, every global_load_dwordx4 of the
    // ring was sunk next to its first use behind an s_waitcnt vmcnt(0) — no fragment in flight at all, the whole L2 latency
    // paid per group of NT MFMAs (round 1 and the first half of round 2: 45 % of the 16-bit pipe whatever the ring depth).
    // Loads return in order, so "at most RING-1 younger loads outstanding" is exactly "fragment `it` has landed".
    u32x4 A[RING];
    auto issue_a = [&](u32x4& dst, int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const u32x4* ptr = wfrag + ((long)(tap * G + 2 * chunk + kgl) * MTn + mt) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s)
        if (s < NG) issue_a(A[s], s);
    u32x4 Bf[2][NT];
    load_b(Bf[0], 0);
    auto body = [&](int it) {
        if (it + RING - 1 < NG) {
            issue_a(A[(it + RING - 1) % RING], it + RING - 1);
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(A[it % RING]) : "n"(RING - 1));
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[it % RING]));     // tail: drain
        }
        if (it + 1 < NG) load_b(Bf[(it + 1) & 1], it + 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mma16<MODE>(A[it % RING], Bf[it & 1][j], acc[j]);
        if (it + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, (CL16_PAD == 8 ? 1 : 2) * NT, 0);       // the next group's B fragments
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);                            // under this group's MFMAs
    };
    // full unrolling in segments of 44 groups: one loop of 112 / 176 groups (C = 256) exceeds the compiler's size limit for
    // "#pragma unroll", stays a run-time loop, and the ring then lives behind s_set_gpr_idx register indexing — an asm-issued
    // load whose destination is copied at once (memory faults and garbage: the first version of conv_xl16_kernel at C = 256)
    seg_loop<0, NG, 44>(body);
}

// The same K loop for a wave that owns MT m-tiles x NT n-tiles (conv_xl16_kernel at C = 128: 2 x 4).  Per MFMA of 32 cycles
// a 1 x 4 wave reads 1 KB of B fragments from LDS — 128 B/clk per CU at full rate, all the LDS delivers — and a 2 x 2 wave
// 512 B of A fragments through the L1 (64 B/clk per CU: its limit); 2 x 4 halves both (256 B of A, 512 B of B per MFMA).
template <int C, int KT, int MT, int NT, int MODE>
__device__ __forceinline__ void conv_loop16m(f32x16 (&acc)[MT][NT], const u32x4* __restrict__ wfrag, const unsigned short* __restrict__ src,
                                             int dil, int mt0, int col0, int lane) {
    constexpr int RS = C + CL16_PAD;
    constexpr int G = C / 16, MTn = C / 32;
    constexpr int NG = G * KT;
    constexpr int RINGM = 6;
    const int l31 = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto grp = [&](int it, int& chunk, int& tap, int& kgl) {
        chunk = it / (2 * KT);
        const int rr = it - chunk * (2 * KT);
        tap = rr >> 1;
        kgl = rr & 1;
    };
    const unsigned short* bl = src + (col0 + l31) * RS + khalf * 8;
    auto load_b = [&](u32x4 (&dst)[NT], int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
        const unsigned short* p = bl + (tap * dil) * RS + chunk * 32 + kgl * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (CL16_PAD == 8) {
                dst[j] = *reinterpret_cast<const u32x4*>(p + j * 32 * RS);
                continue;
            }
            const u32x2 lo = *reinterpret_cast<const u32x2*>(p + j * 32 * RS);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(p + j * 32 * RS + 4);
            dst[j] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    };
    u32x4 A[RINGM][MT];
    auto issue_a = [&](u32x4 (&dst)[MT], int it) {
        int chunk, tap, kgl;
        grp(it, chunk, tap, kgl);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const u32x4* ptr = wfrag + ((long)(tap * G + 2 * chunk + kgl) * MTn + mt0 + i) * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[i]) : "v"(ptr) : "memory");
        }
    };
#pragma unroll
    for (int s = 0; s < RINGM - 1; ++s)
        if (s < NG) issue_a(A[s], s);
    u32x4 Bf[2][NT];
    load_b(Bf[0], 0);
    auto body = [&](int it) {
        if (it + RINGM - 1 < NG) {
            issue_a(A[(it + RINGM - 1) % RINGM], it + RINGM - 1);
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(A[it % RINGM][0]) : "n"((RINGM - 1) * MT));
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[it % RINGM][0]));
        }
#pragma unroll
        for (int i = 1; i < MT; ++i) asm volatile("" : "+v"(A[it % RINGM][i]));      // the other fragments of the group: same wait
        if (it + 1 < NG) load_b(Bf[(it + 1) & 1], it + 1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = mma16<MODE>(A[it % RINGM][i], Bf[it & 1][j], acc[i][j]);
        if (it + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, (CL16_PAD == 8 ? 1 : 2) * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
    };
    seg_loop<0, NG, 22>(body);
}

// bf16 epilogues store FOUR channels per LDS instruction (accumulator registers 4q .. 4q+3 = four consecutive channels of a column = 8 contiguous bytes
// of the [column][channel] image): two packed converts + one ds_write_b64 instead of four converts + four ds_write_b16.  fp16 keeps the per-value
// form: there the compiler fuses `leaky multiply -> convert` into a single-rounding v_fma_mixlo_f16 on the two-launch path, and no packed spelling
// tried reproduced its bits (3e-4 on the wav; caught by test_vocoder_pair16_kernel_bitwise).
template <int MODE>
__device__ __forceinline__ u32x2 pack16x4(const float (&v)[4]) {
    return (u32x2){pack16<MODE>(v[0], v[1]), pack16<MODE>(v[2], v[3])};
}

}  // namespace
