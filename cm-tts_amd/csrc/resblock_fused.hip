// Fused denoiser residual block for gfx950 — ResidualBlock.forward (model/blocks.py:667-686) of the
// CM-TTS consistency denoiser as ONE kernel per layer:
//
//   u  = (x + d [+ p]) + cp            cp = conditioner_projection(cond) + bias, precomputed for all 20
//                                      layers by ONE stacked GEMM (it does not depend on the step)
//   y  = W3 (*) u + b3 ; z = sigmoid(y[:C]) * tanh(y[C:])      conv_layer k=3 pad 1 + gate   (phase B)
//   o  = Wo * z + bo ; x' = (o[:C] + (x + d)) / sqrt(2) ; skip (+)= o[C:]                    (phase C)
//
// One workgroup (16 wave64) owns FN = 64 frames of one utterance; u (with its +-1 frame Conv1D halo)
// and z never leave LDS (z re-uses u's buffer): per layer HBM sees x, cp in and x', skip in/out.
// Both contractions run on v_mfma_f32_32x32x2_f32 (exact fp32) in the same (16-channel chunk, tap, k)
// order as conv_mfma.hip, so the result is BITWISE equal to the three-launch form
// (tests/test_gpu_parity.py::test_fused_resblock_bitwise).
//
// Operand delivery (the part that decides the MFMA duty cycle):
//  * WEIGHTS never touch LDS.  Wave w owns ONE 32-row MFMA tile of the output (rows [32w, +32)) over
//    all 64 frames, and no other wave needs those rows, so they are streamed global/L2 -> VGPR already
//    in MFMA A-fragment order: cmtts_finalize() re-packs each layer as [k-group of 8][m-tile][lane][4]
//    so that ONE coalesced global_load_dwordx4 (1 KB per wave) delivers the A operands of four
//    k-steps = 8 MFMAs.  The measured wall (tools/mfma_probe.hip, profiles/) is the per-CU L2->CU fill
//    rate, ~12 B/clk: a 32-frame tile needs 256 B of weights per MFMA and caps the kernel at ~75 % of
//    the MFMA peak whatever the schedule; 64 frames per workgroup halve that.  A 4-deep register ring
//    keeps three groups in flight; there is no ds_write / wave barrier / LDS double buffering.
//  * The gate needs sigmoid and tanh operands in the same lane: each 32-row tile is packed as
//    [16 sigmoid rows | 16 tanh rows] of the same 16 channels, which the 32x32 C layout puts in
//    registers r and r+8 of one lane.
//  * ACTIVATIONS (u, then z) are read from LDS with conflict-free ds_read_b32 (32 consecutive frames
//    per half-wave), one k-group ahead of the MFMAs that consume them.
//  * 16 waves = 4 per SIMD keep the matrix pipe fed while other waves wait on loads (70 KB LDS,
//    <= 128 VGPRs per wave).
// Three s_barriers per workgroup in total (u staged, u dead, z complete).
#include <hip/hip_runtime.h>
#include "gate.h"
#include "resblock_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs

namespace {

constexpr int C = 256;          // residual channels (= encoder hidden)
constexpr int NW = 16;          // waves per workgroup (one 32-row MFMA tile each)
constexpr int RING = 4;         // register ring depth for the weight stream

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int FN>
__global__ __launch_bounds__(64 * NW, 4) void resblock_fused_kernel(const ResArgs a) {
    static_assert(64 * NW == 1024, "1024-thread workgroups");
    constexpr int NT = FN / 32;         // 32-frame MFMA column tiles per wave
    constexpr int U_LD = FN + 4;        // LDS row stride of u / z (FN + 2 columns used by u)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* u_lds = smem;                // [C][U_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FN;
    const int T = a.T;
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* xin = a.x_in + (long)b * C * T;
    const float* cp = a.cp + (long)b * a.cp_bstride;
    const float* dp = a.dp + (long)b * a.vec_stride;
    const float* dv = a.d + (long)b * a.vec_stride;

    const int bid_dbg = blockIdx.x + gridDim.x * blockIdx.y;
    auto stamp = [&](int slot) {
        if (a.dbg && tid == 0) a.dbg[(long)bid_dbg * 8 + slot] = (long long)__builtin_readcyclecounter();
    };
    stamp(0);

    // ---- stage u[m][j] = cp + (x + dp) for frames t0-1+j, j in [0, FN+2); zero outside [0,T) (conv padding).
    // Unconditional clamped loads, selects at LDS-store time, 8 rows in flight per thread.
    {
        constexpr int RPI = 64 / FN;                 // rows covered by one wave-wide access (2 or 1)
        const int n = lane & (FN - 1);
        const int rsub = RPI == 2 ? khalf : 0;
        const int t = t0 + n;
        const int t_c = min(t, T - 1);
        constexpr int ROWS_PER_WAVE = C / NW;        // 32 rows per wave
#pragma unroll 1
        for (int i = 0; i < ROWS_PER_WAVE / RPI; i += 8) {
            float xv[8], cv[8], dq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = w * ROWS_PER_WAVE + (i + q) * RPI + rsub;
                xv[q] = xin[(unsigned)(m * T + t_c)];
                cv[q] = cp[(unsigned)(m * T + t_c)];
                dq[q] = dp[m];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = w * ROWS_PER_WAVE + (i + q) * RPI + rsub;
                const float uv = cv[q] + (xv[q] + dq[q]);
                u_lds[m * U_LD + 1 + n] = t < T ? uv : 0.f;
            }
        }
        if (tid < 2 * C) {                           // halo columns: thread = (side, row)
            const int m = tid & (C - 1);
            const bool right = tid >= C;
            const int th = right ? t0 + FN : t0 - 1;
            const int thc = min(max(th, 0), T - 1);
            const float uh = cp[(unsigned)(m * T + thc)] + (xin[(unsigned)(m * T + thc)] + dp[m]);
            u_lds[m * U_LD + (right ? FN + 1 : 0)] = (th >= 0 && th < T) ? uh : 0.f;
        }
    }
    __syncthreads();   // (1) u staged
    stamp(1);

    f32x16 acc[NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    };
    // A fragments of one k-group (8 input channels = 4 k-steps) for this wave's 32-row tile
    auto load_a = [&](f32x4& dst, const float* wfrag, int group) {
        dst = *reinterpret_cast<const f32x4*>(wfrag + ((long)group * (2 * C / 32) + w) * 256 + lane * 4);
    };
    // B operands of one k-group: rows krow0 + 2kk + khalf of the LDS tile at column offset `col`
    auto load_b = [&](float (&dst)[4][NT], int krow0, int col) {
        const float* bs = u_lds + (krow0 + khalf) * U_LD + l31 + col;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j) dst[kk][j] = bs[2 * kk * U_LD + j * 32];
    };
    auto mma_group = [&](const f32x4& af, const float (&bv)[4][NT]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bv[kk][j], acc[j], 0, 0, 0);
    };

    // =============================================================== phase B: gated k=3 conv
    {
        zero_acc();
        constexpr int NG = (C / 8) * 3;                       // 96 k-groups: (16-chunk, tap, 8-half)
        auto kgrp = [&](int it, int& g8, int& tap) {           // iteration -> (k-group of 8 channels, tap)
            it = min(it, NG - 1);
            const int q = it / 6, rr = it - q * 6;
            tap = rr >> 1;
            g8 = 2 * q + (rr & 1);
        };
        f32x4 A[RING];
        float Bv[2][4][NT];
        int g8, tap;
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) {
            kgrp(s, g8, tap);
            load_a(A[s], a.W3f, tap * (C / 8) + g8);
        }
        kgrp(0, g8, tap);
        load_b(Bv[0], g8 * 8, tap);
        for (int it = 0; it < NG; it += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                kgrp(it + s + RING - 1, g8, tap);
                load_a(A[(s + RING - 1) % RING], a.W3f, tap * (C / 8) + g8);
                kgrp(it + s + 1, g8, tap);
                load_b(Bv[(s + 1) & 1], g8 * 8, tap);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(A[s], Bv[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    stamp(2);
    __syncthreads();   // (2) every wave is done reading u: its buffer becomes z
    stamp(3);
    {
        // this wave's tile = [16 sigmoid rows | 16 tanh rows] of channels [16w, +16): registers r (rows 0-15)
        // and r+8 (rows 16-31) of the same lane pair up
        float bg[8], bf[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int mg = w * 32 + acc_row(r, lane);
            bg[r] = a.b3[mg];
            bf[r] = a.b3[mg + 16];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float zv = cmtts_gate(acc[j][r] + bg[r], acc[j][r + 8] + bf[r]);
                u_lds[(w * 16 + acc_row(r, lane)) * U_LD + j * 32 + l31] = zv;
            }
    }
    __syncthreads();   // (3) z complete
    stamp(4);

    // =============================================================== phase C: output projection
    {
        zero_acc();
        constexpr int NG = C / 8;                              // 32 k-groups
        f32x4 A[RING];
        float Bv[2][4][NT];
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) load_a(A[s], a.Wof, s);
        load_b(Bv[0], 0, 0);
        for (int it = 0; it < NG; it += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                load_a(A[(s + RING - 1) % RING], a.Wof, min(it + s + RING - 1, NG - 1));
                load_b(Bv[(s + 1) & 1], min(it + s + 1, NG - 1) * 8, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(A[s], Bv[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(5);
        float* xout = a.x_out + (long)b * C * T;
        float* skip = a.skip + (long)b * C * T;
        const bool res_half = w < NW / 2;         // wave-uniform: rows [0,256) -> x', rows [256,512) -> skip
        const float* src = res_half ? xin : skip;
        float* dst = res_half ? xout : skip;
        const bool need_src = res_half || a.accum_skip;
        const int mrow0 = (w % (NW / 2)) * 32;                   // row inside the half
        float bo[16], dd[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bo[r] = a.bo[w * 32 + acc_row(r, lane)];
            dd[r] = res_half ? dv[mrow0 + acc_row(r, lane)] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = t0 + j * 32 + l31;
            const int t_c = min(t, T - 1);
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)             // all gathers first: 16 loads in flight
                sv[r] = need_src ? src[(unsigned)((mrow0 + acc_row(r, lane)) * T + t_c)] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = mrow0 + acc_row(r, lane);
                const float o = acc[j][r] + bo[r];
                float v;
                if (res_half) v = (o + (sv[r] + dd[r])) * CMTTS_RSQRT2;
                else v = a.accum_skip ? o + sv[r] : o;
                if (t < T) dst[(unsigned)(mr * T + t)] = v;
            }
        }
    }
    stamp(6);
}

long long* g_dbg = nullptr;
int g_force_fn = 0;

template <int FN>
int launch_fn(const ResArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    const size_t lds = (size_t)(C * (FN + 4)) * sizeof(float);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_fused_kernel<FN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.T + FN - 1) / FN, a.B);
    hipLaunchKernelGGL(resblock_fused_kernel<FN>, grid, dim3(64 * NW), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

extern "C" void cmtts_resblock_set_debug(long long* dbg) { g_dbg = dbg; }
extern "C" void cmtts_resblock_set_tile(int frames) { g_force_fn = frames; }

extern "C" int cmtts_launch_resblock(const ResArgs* a_in, void* stream) {
    ResArgs a = *a_in;
    a.dbg = g_dbg;
    if ((long)C * a.T >= (1L << 31)) return -2;
    // 64-frame tiles halve the weight bytes per MFMA (the measured wall); 32-frame tiles only when 64-frame
    // tiles could not even give every CU one workgroup
    const long tiles64 = (long)((a.T + 63) / 64) * a.B;
    const bool use32 = g_force_fn ? g_force_fn == 32 : tiles64 < 256;
    return use32 ? launch_fn<32>(a, (hipStream_t)stream) : launch_fn<64>(a, (hipStream_t)stream);
}
