// X-resident Conv1D for short sequences on gfx950: the k=9 FFN conv of the FFT blocks (model/blocks.py:539-546,
// Conv1d(256 -> 1024, k=9) + scale + GELU over L <= ~100 phonemes).  The generic kernel (conv_mfma.hip) runs an
// 85-phoneme utterance as two 64-column tiles (a third of the MFMAs multiply padding) and re-stages X per m-tile.
// Here a workgroup (4 waves) stages the whole X tile [K <= 256][96 + halo] of one utterance in LDS once, each wave
// owns one 32-row m-tile over three 32-column n-tiles, and the weights stream L2 -> VGPR in MFMA A-fragment order
// (iteration order [chunk][tap][half][m-tile][lane][4], one dwordx4 per 12 MFMAs) through a register ring — no weight staging,
// no barrier in the K loop.  Same (16-channel chunk, tap, k) accumulation order and the same epilogue arithmetic as
// the generic kernel: BITWISE equal (tests/test_gpu_parity.py::test_xres_conv_bitwise).
#include <hip/hip_runtime.h>
#include "conv_args.h"
#include "conv_epilogue.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KMAX = 256;
constexpr int BN = 96;          // columns per workgroup (3 MFMA n-tiles)
constexpr int NT = 3;
constexpr int HALO = 8;         // (taps - 1) * dil <= 8
constexpr int X_LD = BN + HALO; // 104
constexpr int RING = 4;

__global__ __launch_bounds__(256, 1) void conv_xres_kernel(const ConvArgs a, const float* __restrict__ wfrag) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [K][X_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * BN;
    const int mt = blockIdx.y * 4 + w;             // this wave's m-tile
    const int z = blockIdx.z;
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* Xb = a.X + z * a.x_zs0;
    const int MTn = (a.M + 31) / 32;
    const int total = (a.K / 16) * a.taps * 2;     // k-groups: (16-channel chunk, tap, 8-channel half)

    // weights: A fragments in ITERATION order ([chunk][tap][half][m-tile][lane][4], cmtts_finalize): the stream of k-group
    // it = chunk * nq + tap * 2 + half is one linear walk (clamped m-tile: idle waves load valid memory)
    const int mtc = min(mt, MTn - 1);
    const float* wl = wfrag + ((long)mtc * 64 + lane) * 4;
    auto load_a = [&](f32x4& dst, int it) {
        dst = *reinterpret_cast<const f32x4*>(wl + (long)min(it, total - 1) * MTn * 256);
    };
    f32x4 A[RING];
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load_a(A[s], s);

    {   // stage X[k][n0 - pad + c], c in [0, X_LD): zero outside [0, Tin); lane = column (two passes), a quarter of the
        // rows per wave, 32 unconditional clamped loads in flight per lane
        const int xw = BN + (a.taps - 1) * a.dil;
        const int rows = a.K / 4;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int c = lane + 64 * cp;
            const int t = n0 - a.pad + c;
            const bool ok = c < xw && t >= 0 && t < a.Tin;
            const int tc = min(max(t, 0), a.Tin - 1);
            if (c < X_LD) {
#pragma unroll 1
                for (int k0 = w * rows; k0 < (w + 1) * rows; k0 += 32) {
                    float v[32];
#pragma unroll
                    for (int q = 0; q < 32; ++q) v[q] = Xb[(long)min(k0 + q, a.K - 1) * a.ldx + tc];
#pragma unroll
                    for (int q = 0; q < 32; ++q)
                        if (k0 + q < (w + 1) * rows) xs[(k0 + q) * X_LD + c] = ok ? v[q] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // scalar walk over (chunk, tap, half): boff = float offset of the group's first row / tap column in the X tile
    const float* bl = xs + khalf * X_LD + l31;
    int boff = 0, tap = 0, half = 0;
    const int boff_max = (a.K - 8) * X_LD + (a.taps - 1) * a.dil;
    auto advance = [&]() {
        if (half == 0) { half = 1; boff += 8 * X_LD; }
        else {
            half = 0; boff -= 8 * X_LD; ++tap; boff += a.dil;
            if (tap == a.taps) { tap = 0; boff += 16 * X_LD - a.taps * a.dil; }
        }
    };
    float Bv[2][4][NT];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < NT; ++j) Bv[0][kk][j] = bl[2 * kk * X_LD + j * 32];
    // loads are threaded between the MFMAs (one k-step's three ds_reads + the A load of a later group per k-step): a wave
    // issues in order, so loads bunched between two groups of MFMAs leave the matrix pipe idle while they issue
#pragma unroll 1
    for (int it = 0; it < total; it += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            advance();
            const float* bs = bl + min(boff, boff_max);
            load_a(A[(s + RING - 1) % RING], it + s + RING - 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int j = 0; j < NT; ++j) Bv[(s + 1) & 1][kk][j] = bs[2 * kk * X_LD + j * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][kk], Bv[s & 1][kk][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
                if (kk == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            }
        }
    }
    if (mt >= MTn) return;
    const ConvOut& o = a.out[0];
#pragma unroll
    for (int j = 0; j < NT; ++j) epi_tile(o, acc[j], mt * 32, 4 * khalf, n0 + j * 32 + l31, a.M, a.N, z, 0);
}

}  // namespace

// Conv1d with fp32 weights as MFMA A fragments in iteration order [K/16][taps][2][ceil(M/32)][64][4]; zdiv == 1, split == INT_MAX,
// dil > 0, K % 32 == 0, K <= 256, (taps-1)*dil <= 8, no pre-activation.  Meant for short sequences (N of a few 96-column tiles)
// with many output rows.  Returns 0, -2 (unsupported: use cmtts_launch_conv) or -3.
extern "C" int cmtts_launch_conv_xres(const ConvArgs* ap, const float* wfrag, int nbatch, void* stream_) {
    const ConvArgs& a = *ap;
    if (a.M <= 0 || a.N <= 0 || nbatch <= 0) return 0;
    if (!wfrag || a.zdiv != 1 || a.split != INT_MAX || a.dil <= 0 || a.K % 32 != 0 || a.K > KMAX || (a.taps - 1) * a.dil > HALO ||
        a.pre_div != 1.0f || a.pre_slope != 1.0f)
        return -2;
    static bool attr_set = false;
    const size_t lds = (size_t)a.K * X_LD * sizeof(float);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xres_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)KMAX * X_LD * sizeof(float))) != hipSuccess)
            return -3;
        attr_set = true;
    }
    dim3 grid((a.N + BN - 1) / BN, ((a.M + 31) / 32 + 3) / 4, nbatch);
    hipLaunchKernelGGL(conv_xres_kernel, grid, dim3(256), lds, (hipStream_t)stream_, a, wfrag);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
