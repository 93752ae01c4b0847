// Persistent denoiser stack for gfx950: ALL residual layers of Denoiser.forward (model/modules.py:626-633,
// ResidualBlock model/blocks.py:667-686) in ONE launch.  A workgroup (8 wave64) keeps its 64-frame tile of
// one utterance for the whole stack:
//
//   * every wave keeps 32 rows of x (the residual stream) and 32 rows of the skip sum in registers, in the MFMA
//     accumulator layout: per layer HBM sees only cp (the precomputed conditioner projection) — x, x', skip are
//     neither re-read nor re-written (resblock_fused.hip moves 5 KB/frame/layer, this kernel 1 KB);
//   * LDS holds u and z in separate buffers, so a wave gates its rows as soon as ITS k=3 conv is done (two
//     workgroup barriers per layer instead of three) and every wave writes its 32 rows of the NEXT layer's u in place,
//     straight from registers, after the projection (the cp tile was pulled into L2 during the conv);
//   * the +-1 frame Conv1D halo is the only inter-workgroup traffic: the two edge columns of x' (2 x 256 values)
//     go to the neighbouring tiles as 8-byte {layer tag, value} granules — write-through agent-scope stores, the
//     consumer re-reads until every tag matches (cdna_hip_programming.md §6 Guideline 16, form R2: the data is
//     the flag, no fences).  Granule slots alternate by layer parity: a workgroup can be at most one layer ahead
//     of a neighbour, because finishing layer l+1 needs the neighbour's layer-l columns.
//
//   * after the last layer the skip head (skip_projection, ReLU, output_projection) and the sampler's post-scaling run
//     in the same launch (persist_tail.h): the skip sum never leaves the chip.
//
// Every workgroup must be resident: the launcher keeps the grid <= the CU count (one 512-thread, 139-KB-LDS workgroup
// per CU) and splits larger batches into balanced utterance chunks; spins are bounded and report through a timeout word.
// Arithmetic and accumulation order are those of resblock_fused.hip: BITWISE equal to the per-layer kernels
// (tests/test_gpu_parity.py::test_persistent_denoiser_bitwise).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gate.h"
#include "persist_args.h"
#include "persist_tail.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned long long gu64;

#ifndef WINO_ABL
#define WINO_ABL 0
#endif
#ifndef WINO_L2PF
#define WINO_L2PF 0        // 1: touch the next layer's transformed weights into L2 during the projection (measured: no effect on the conv loop)
#endif
#ifndef WINO_THREAD
#define WINO_THREAD 1
#endif
#ifndef WINO43_RING
#define WINO43_RING 3      // k-steps of F(4,3) transformed weights in flight per wave + the one in use (24 registers each)
#endif
#ifndef WINO_RING
#define WINO_RING 3        // half-groups of transformed weights in flight per wave + the one in use (WINO instances)
#endif

namespace {

constexpr int C = 256;
constexpr int NW = 8;           // waves per workgroup: 2 per SIMD, each owning 2 m-tiles x 2 n-tiles (4 accumulators)
constexpr int MT = 2;           // 32-row MFMA tiles per wave
constexpr int RING = 6;         // register ring depth of the weight stream (k-groups in flight)
constexpr int FN = 64;
constexpr int NT = FN / 32;
constexpr int U_LD = FN + 4;
constexpr int IDX_LDS_BYTES = (FN + 2) * 2 * 4 + NW * 64 * 4 + NW * 4;      // behind the u / z buffers: FACT's (phoneme, pitch bucket) table of frames t0 - 1 .. t0 + FN, and
                                                                   // 64 floats per wave to turn its two edge columns into ONE 64-lane granule store
constexpr unsigned SPIN_LIMIT = 1u << 20;     // bounded wait for a neighbour (~2 s); then the timeout word is set and
                                              // this wave stops waiting for the rest of the launch (results invalid)

// An opaque copy of a lane-dependent value: address arithmetic derived from it cannot be hoisted out of the layer
// loop (hoisted per-lane offsets of the staging / epilogue sections would push the resident tile into scratch).
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// base[idx] with the byte offset formed in 32 bits, so the load takes the (SGPR base + 32-bit VGPR offset) form instead of
// a 64-bit VGPR address per element (64 of those in flight would not fit the register file).  idx < 2^30.
__device__ __forceinline__ float ldg(const float* base, unsigned idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 4u));
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned tag, float v) {
    __hip_atomic_store((gu64*)g, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// DBG: cycle stamps of the middle layer (tools/persist_timing.py); the production instance has none.
// RAGGED (round 3, BASELINE.json configs[3]): a 1-D grid over a tile-descriptor list — the utterances of SEVERAL frame buckets
// (groups: own buffers, own padded T) fill the chip together, and an utterance may be TRIMMED to its first `active` tiles:
// frames at and beyond Tc = active * 64 behave exactly like frames beyond T (u = 0 there, no neighbour, nothing stored).  Frames
// closer than NL to Tc differ from the untrimmed result; the caller keeps Tc far enough beyond the frames it uses
// (cmtts_api.hip: sample_ragged).  The arithmetic of a computed frame is unchanged: every value is bit-identical to the uniform
// launch of its own bucket as long as no trimmed frame lies within its receptive field.
// FACT (round 4): the conditioner projections are gathered from their factors (persist_args.h) wherever cp would be read.
// WINO == 2 (round 5, the default): the same conv as Winograd F(4,3) over QUADS of output frames with the points 0, +-1, +-2, inf — six
// products per quad instead of twelve (98 k pipe cycles per layer).  With d0..d5 = u(4q-1 .. 4q+4): V0 = 4 d0 - 5 d2 + d4,
// V1 = (d4 - 4 d2) + (d3 - 4 d1), V2 = (d4 - 4 d2) - (d3 - 4 d1), V3 = (d4 - d2) + 2 (d3 - d1), V4 = (d4 - d2) - 2 (d3 - d1),
// V5 = 4 d1 - 5 d3 + d5; m_p = U_p V_p with U0 = g0/4, U1 = -(g0+g1+g2)/6, U2 = -(g0-g1+g2)/6, U3 = g0/24 + g1/12 + g2/6,
// U4 = g0/24 - g1/12 + g2/6, U5 = g2 (cmtts_api.hip: to_wino43_fragments); y0 = m0 + (m1+m2) + (m3+m4), y1 = (m1-m2) + 2 (m3-m4),
// y2 = (m1+m2) + 4 (m3+m4), y3 = (m1-m2) + 8 (m3-m4) + m5.  One n-tile of v_mfma_f32_16x16x4_f32 = one transform of the tile's 16 quads
// (all six transforms of a quad in the same lane), a wave carries 4 sixteen-row m-tiles x 6 transforms = 24 accumulators of 4 registers;
// u in natural frame order.  State, barriers, halo protocol, FACT, RAGGED, tail: as WINO == 1.  Restated in oracle/winograd_ref.py
// (conv1d_f43) and checked against the plain conv on the CPU (tests/test_winograd_tables.py).
// WINO == 1 (round 4): the gated k = 3 conv as a Winograd F(2,3) convolution along the frame axis — per PAIR of output frames four products
// instead of six: m0 = (d0 - d2) g0, m1 = (d1 + d2) (g0 + g1 + g2)/2, m2 = (d2 - d1) (g0 - g1 + g2)/2, m3 = (d1 - d3) g2,
// y(2p) = m0 + m1 + m2, y(2p+1) = m1 - m2 - m3 with d0..d3 = u(2p-1 .. 2p+2).  Each m_i is its own K = 256 contraction over the
// channels (transformed weights W3f = [64 half-groups][16 m-tiles][2][64 lanes][4], packed by cmtts_api.hip: to_wino_fragments), so a
// wave carries 2 m-tiles x 4 transforms = 8 accumulators over ONE 32-pair n-tile and the conv costs 2/3 of the direct form's MFMAs
// (131 k instead of 197 k pipe cycles per layer).  u lives in LDS split by frame parity (odd frames at row offset (f + 1) / 2,
// even frames at 33 + f / 2) so that the four d_i of a pair are unit-stride reads; the residual stream x makes room in the register
// file by living in `xst` memory between layers (read-modify-write by the lane that owns the element, L2-resident); the skip sum stays.
// NOT bitwise equal to the direct form (fp32 Winograd: ~1e-6 relative per layer); everything else in the kernel is unchanged.
template <bool DBG, bool RAGGED, bool FACT = false, int WINO = 0>
__global__ __launch_bounds__(64 * NW, 2) void denoiser_persist_kernel(const PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int tile_, b_, T_, Tc_, gi = 0;
    if (RAGGED) {
        const unsigned d = a.desc[blockIdx.x];
        gi = d & 7;
        b_ = (d >> 3) & 1023;
        tile_ = (d >> 13) & 127;
        T_ = a.grp[gi].T;
        Tc_ = min(T_, (int)((d >> 20) & 255) * FN);
    } else {
        tile_ = blockIdx.x; b_ = blockIdx.y; T_ = a.T; Tc_ = a.T;
    }
    const int tile = tile_, b = b_;
    const int t0 = tile * FN;
    const int T = T_;                                  // row stride / address clamp
    const int Tc = Tc_;                                // frames that exist for this launch (== T unless trimmed)
    const int l31 = lane & 31, khalf = lane >> 5;
    const float* x0_b = (RAGGED ? a.grp[gi].x0 : a.x0) + (long)b * C * T;
    const float* cp_b = (RAGGED ? a.grp[gi].cp + (long)b * a.grp[gi].cp_bstride : a.cp + (long)b * a.cp_bstride);
    const float* dp_b = (RAGGED ? a.grp[gi].dp : a.dp) + (long)b * a.vec_stride;
    const float* dv_b = (RAGGED ? a.grp[gi].d : a.d) + (long)b * a.vec_stride;
    unsigned long long* halo_g = RAGGED ? a.grp[gi].halo : a.halo;
    const int B_g = RAGGED ? a.grp[gi].B : a.B, tiles_g = RAGGED ? a.grp[gi].tiles : a.tiles;
    const int mrow0 = w * 32;                          // this wave's 32 rows of x (state tile 0) and of the skip sum (tile 1)
    // FACT: cp of (row m of layer l, frame t) from the factors; (ph, ix) = frame_idx(t)
    const float *p1_b = nullptr, *p1t_b = nullptr;
    const long long *m2p_b = nullptr, *pix_b = nullptr;
    int ldp = 0, Lph = 0;
    if (FACT) {
        p1_b = (RAGGED ? a.grp[gi].p1 : a.p1);
        ldp = RAGGED ? a.grp[gi].ldp : a.ldp;
        Lph = RAGGED ? a.grp[gi].Lph : a.Lph;
        p1_b += (long)b * a.NL * C * ldp;
        p1t_b = (RAGGED ? a.grp[gi].p1t : a.p1t) + (long)b * a.NL * C * ldp;
        m2p_b = (RAGGED ? a.grp[gi].mel2ph : a.mel2ph) + (long)b * T;
        pix_b = (RAGGED ? a.grp[gi].pidx : a.pidx) + (long)b * T;
    }
    auto frame_idx = [&](int t_c, int& ph, int& ix) {
        const long long p = m2p_b[t_c], q = pix_b[t_c];
        ph = (int)(p > Lph ? Lph : p);
        ix = q < 0 ? 0 : (q >= a.ld2 ? a.ld2 - 1 : (int)q);
    };
    // (32-bit offsets through ldg() with the row term hoisted out of the 16-register loop were tried: 2.74 -> 2.75 ms per launch, slower)
    auto cp_fact = [&](int row, int ph, int ix) -> float {      // row = l * C + m; cond_expand_kernel's expression
        const float av = p1_b[(long)row * ldp + (ph > 0 ? ph - 1 : 0)];
        const float qv = a.p2[(long)row * a.ld2 + ix];
        return (ph > 0 ? av : 0.f) + qv;
    };

    // column of frame f (-1 .. FN) within a u row: f + 1, or (WINO) odd frames first, then even frames
    auto uidx = [](int f) { return WINO == 1 ? ((f & 1) ? (f + 1) >> 1 : 33 + (f >> 1)) : f + 1; };
    // WINO: the residual stream x of this tile between layers, in a kernel-private layout [wave][j * 4 + q][lane][4] (element e of that
    // float4 = accumulator register 4 q + e of n-tile j): every load / store of the state is one fully coalesced 16-byte-per-lane instruction
    constexpr int PST_TILE = NW * 8 * 64 * 4;       // floats of one tile (64 KB)
    float* pst_b = nullptr;
    if (WINO) pst_b = (RAGGED ? a.grp[gi].xst : a.xst) + ((long)b * (RAGGED ? a.grp[gi].tiles : a.tiles) + tile) * PST_TILE;

    // ---- layer-0 staging (as resblock_fused.hip): u = cp + (x + dp), halo columns straight from x0
    {
        const float* xin = x0_b;
        const int t = t0 + lane;
        const int t_c = min(t, T - 1);
        int ph0 = 0, ix0 = 0;
        if (FACT) frame_idx(t_c, ph0, ix0);
        constexpr int ROWS_PER_WAVE = C / NW;
#pragma unroll 1
        for (int i = 0; i < ROWS_PER_WAVE; i += 8) {
            float xv[8], cv[8], dq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = w * ROWS_PER_WAVE + i + q;
                xv[q] = xin[(unsigned)(m * T + t_c)];
                cv[q] = FACT ? cp_fact(m, ph0, ix0) : cp_b[(unsigned)(m * T + t_c)];
                dq[q] = dp_b[m];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = w * ROWS_PER_WAVE + i + q;
                const float uv = cv[q] + (xv[q] + dq[q]);
                smem[m * U_LD + uidx(lane)] = t < Tc ? uv : 0.f;
            }
        }
        if (tid < 2 * C) {
            const int m = tid & (C - 1);
            const bool right = tid >= C;
            const int th = right ? t0 + FN : t0 - 1;
            const int thc = min(max(th, 0), T - 1);
            float cph0;
            if (FACT) { int ph, ix; frame_idx(thc, ph, ix); cph0 = cp_fact(m, ph, ix); }
            else cph0 = cp_b[(unsigned)(m * T + thc)];
            const float uh = cph0 + (xin[(unsigned)(m * T + thc)] + dp_b[m]);
            smem[m * U_LD + uidx(right ? FN : -1)] = (th >= 0 && th < Tc) ? uh : 0.f;
        }
    }
    // FACT: the factor indices of the tile's frames (and of its two halo frames) are the same for every layer: one table behind the u / z
    // buffers, filled here (visible after the first layer barrier), read by the publish phase with one ds_read_b64 per n-tile — until round 5
    // every layer re-loaded them from HBM / L2 in front of its gathers: two of that phase's three dependent round trips (~3 k cycles each
    // with every wave of the chip asking at once)
    int* idx_lds = reinterpret_cast<int*>(smem + 2 * C * U_LD);
    // one word per wave behind the edge scratch: the last layer whose gate output (z rows 32 w ..) this wave has written — what the output
    // projection waits for, k-block by k-block, instead of a workgroup barrier (phase C)
    int* zflag = idx_lds + (FN + 2) * 2 + NW * 64;
    if (tid < NW) zflag[tid] = 0;
    if (FACT && tid < FN + 2) {
        int ph, ix;
        frame_idx(min(max(t0 - 1 + tid, 0), T - 1), ph, ix);
        idx_lds[2 * tid] = ph;
        idx_lds[2 * tid + 1] = ix;
    }
    // resident state: st[0] = this wave's 32 rows of x, st[1] = its 32 rows of the skip sum; MFMA C layout: [j][r] = row
    // acc_row(r), frame j*32 + l31
    f32x16 st[MT][NT];
    {
        const float* xin = x0_b;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t_c = min(t0 + j * 32 + l31, T - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[i][j][r] = (i == 0 && !WINO) ? ldg(xin, (unsigned)((mrow0 + acc_row(r, lane)) * T + t_c)) : 0.f;
            }
    }

    f32x16 acc[MT][NT];
    // A fragments of one k-group (8 input channels = 4 k-steps) for this wave's MT 32-row tiles
    auto load_a = [&](f32x4 (&dst)[MT], const float* wfrag, int group) {        // k=3 conv: packed gate tiles 2w, 2w+1
#pragma unroll
        for (int i = 0; i < MT; ++i)
            dst[i] = *reinterpret_cast<const f32x4*>(wfrag + ((long)group * (2 * C / 32) + w * MT + i) * 256 + lane * 4);
    };
    // output projection: tile w (rows 32w.. of the residual half) and tile NW + w (rows 32w.. of the skip half), so that
    // every wave owns 32 rows of x AND 32 rows of the skip sum and the epilogue work is spread over all waves
    auto load_ao = [&](f32x4 (&dst)[MT], const float* wfrag, int group) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
            dst[i] = *reinterpret_cast<const f32x4*>(wfrag + ((long)group * (2 * C / 32) + i * NW + w) * 256 + lane * 4);
    };
    // FIRST: the group opens the accumulators — its first k-step takes the inline constant 0 as C (`v_mfma ..., 0`: the same operation as
    // accumulating into zeroed registers without the 64 v_mov per loop that, next to fp32 MFMAs, cost their own issue time)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // I0: the first m-tile this group computes (1 in the LAST layer's output projection: nobody reads that layer's x')
    auto mma_group = [&](auto first, const f32x4 (&af)[MT], const float (&bv)[4][NT], auto i0) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr int I0 = decltype(i0)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = I0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bv[kk][j], FIRST && kk == 0 ? zero16 : acc[i][j], 0, 0, 0);
    };
    using first_t = std::integral_constant<bool, true>;
    using later_t = std::integral_constant<bool, false>;

    const int bid_dbg = blockIdx.x + gridDim.x * blockIdx.y;
    // -DPUB_STAMP (timing-only builds, tools/persist_timing.py PUB=1): slots 0..5 take the publish phase's own steps instead of the layer's
#ifdef PUB_STAMP
    constexpr bool PUBS = true;
#else
    constexpr bool PUBS = false;
#endif
    auto stamp = [&](int l, int slot) {
        if (DBG && !(PUBS && slot < 6) && a.dbg && l == a.NL / 2 && lane == 0)
            a.dbg[((long)bid_dbg * NW + w) * 8 + slot] = (long long)__builtin_readcyclecounter();
    };
    auto pstamp = [&](int l, int slot) {
        if (DBG && PUBS && a.dbg && l == a.NL / 2 && lane == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            a.dbg[((long)bid_dbg * NW + w) * 8 + slot] = (long long)__builtin_readcyclecounter();
        }
    };
    bool gave_up = false;
    for (int l = 0; l < a.NL; ++l) {
        float* u_lds = smem;                  // u of the current layer, then assembled in place for the next one
        float* z_lds = smem + C * U_LD;       // gate output
        const bool more = l + 1 < a.NL;
        constexpr int NGB = (C / 8) * 3;         // k-groups of the k=3 conv: (16-channel chunk, tap, 8-half)
        constexpr int NGC = C / 8;               // k-groups of the output projection
        auto kgrp = [&](int it, int& g8, int& tap) {
            it = min(it, NGB - 1);
            const int q = it / 6, rr = it - q * 6;
            tap = rr >> 1;
            g8 = 2 * q + (rr & 1);
        };
        // the weight stream does not depend on u: its first RING-1 k-groups are requested before the barrier
        f32x4 A[RING][MT];
        // WINO: a stage = one half-group (4 channels = 2 k-steps) of all four transforms for this wave's two m-tiles: 4 x 16 bytes per lane,
        // element q of fragment (i, ps) = transform 2 ps + (q >> 1), k-step q & 1
        constexpr int WR = WINO_RING, NH = C / 4;
        f32x4 Aw[WINO == 1 ? WR : 1][MT][2];
        // WINO == 2, F(4,3): a stage = one k-step of FOUR channels of all six transforms for this wave's four 16-row m-tiles: 6 x 16 bytes per lane,
        // element e of fragment p = transform p, m-tile e (cmtts_api.hip: to_wino43_fragments)
        constexpr int WR4 = WINO43_RING, NS4 = C / 4;
        f32x4 A4[WINO == 2 ? WR4 : 1][6];
        // (buffer loads: the layer's array as a descriptor, the lane's fragment offset in a VGPR that never changes, the k-step's offset
        //  in an SGPR — no per-load VALU address arithmetic next to the MFMAs; flat loads took two 64-bit VALU adds per k-step)
        auto load_a4 = [&](f32x4 (&dst)[6], const float* wfrag, int ks) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, (NS4 + 4) * (NW * 6 * 64 * 16), 0x00020000);
            const int voff = (w * 6 * 64 + lane) * 16;
#pragma unroll
            for (int p = 0; p < 6; ++p)
                dst[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + p * 1024, ks * (NW * 6 * 64 * 16), 0));
        };
        // (uniform base + 32-bit lane offset: the saddr form, no per-step VALU address arithmetic; the ring reads up to WR - 1 half-groups past
        //  the layer's last one — the packer pads every layer's array by that much, cmtts_api.hip: to_wino_fragments)
        auto load_aw = [&](f32x4 (&dst)[MT][2], const float* wfrag, int hg) {
            const char* base = reinterpret_cast<const char*>(wfrag) + (size_t)hg * ((2 * C / 32) * 2 * 64 * 16);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps)
                    dst[i][ps] = *reinterpret_cast<const f32x4*>(base + (size_t)((unsigned)((((w * MT + i) * 2 + ps) * 64 + lane) * 16)));
        };
        if (WINO == 2) {
#pragma unroll
            for (int s = 0; s < WR4 - 1; ++s) load_a4(A4[s], a.W3f[l], s);
        } else if (WINO == 1) {
#pragma unroll
            for (int s = 0; s < WR - 1; ++s) load_aw(Aw[s], a.W3f[l], s);
        } else {
            int g8, tap;
#pragma unroll
            for (int s = 0; s < RING - 1; ++s) {
                kgrp(s, g8, tap);
                load_a(A[s], a.W3f[l], tap * (C / 8) + g8);
            }
        }
        stamp(l, 0);
        __syncthreads();   // (1) u of layer l complete (interior, halo columns)
        stamp(l, 1);
        if (!FACT && more && tid < 2 * C) {
            // pull the next layer's cp tile (256 rows x 256 B) towards this XCD's L2 now: one dword per 128-B line;
            // the x waves read it in the accumulator layout after the output projection and would otherwise pay the
            // HBM latency there
            const float* cpn = cp_b + (long)(l + 1) * C * T;
            const int tl = opaque(tid);
            const float warm = cpn[(unsigned)((tl >> 1) * T + min(t0 + (tl & 1) * 32, T - 1))];
            asm volatile("" ::"v"(warm));
        }

        // =========================================================== phase B: gated k=3 conv
        f32x16 accw[WINO == 1 ? MT : 1][WINO == 1 ? 4 : 1];      // WINO: m-tile x transform, one n-tile of 32 frame PAIRS
        f32x4 acc4[WINO == 2 ? 4 : 1][WINO == 2 ? 6 : 1];        // F(4,3): 16-row m-tile x transform, one n-tile of the 16 frame QUADS
        if constexpr (WINO == 2) {
            // F(4,3) along the frame axis (header comment): per quad of output frames six products instead of twelve.  One n-tile of
            // v_mfma_f32_16x16x4_f32 is one transform of the tile's 16 quads, so lane (q = l & 15, k = l >> 4) reads the six inputs
            // u(4q-1 .. 4q+4) of its quad in channel 4 ks + k (one 16-byte + one 8-byte LDS read, natural frame order), forms the six
            // transformed inputs (12 VALU) and feeds 6 transforms x 4 m-tiles = 24 MFMAs of 32 cycles.
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[i][p][r] = 0.f;
            const float* W3f = a.W3f[l];
            float V4[6];
            f32x4 Da;
            float2 Db;
            auto load_d4 = [&](int ks) {      // (the read past the last k-step lands in the z buffer: discarded)
                const float* rr = u_lds + (4 * ks + (lane >> 4)) * U_LD + 4 * (lane & 15);
                Da = *reinterpret_cast<const f32x4*>(rr);
                Db = *reinterpret_cast<const float2*>(rr + 4);
            };
            // (written on float pairs: the compiler then issues v_pk_fma_f32 / v_pk_add_f32 — 8 VALU operations per k-step instead of 12 + moves;
            //  the same fused operations on the same values)
            auto transform4 = [&]() {
                const f32x2 P01 = {Da[0], Da[1]}, P23 = {Da[2], Da[3]}, P45 = {Db.x, Db.y};
                const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, c2 = {2.f, -2.f};
                const f32x2 V05 = __builtin_elementwise_fma(c4, P01, __builtin_elementwise_fma(cm5, P23, P45));
                const float t0 = __builtin_fmaf(-4.f, Da[2], Db.x), t1 = __builtin_fmaf(-4.f, Da[1], Da[3]);
                const float t2 = Db.x - Da[2], t3 = Da[3] - Da[1];
                const f32x2 a0 = {t0, t0}, a1 = {t1, -t1}, b0 = {t2, t2}, b1 = {t3, t3};
                const f32x2 V12 = a0 + a1;
                const f32x2 V34 = __builtin_elementwise_fma(c2, b1, b0);
                V4[0] = V05.x; V4[1] = V12.x; V4[2] = V12.y; V4[3] = V34.x; V4[4] = V34.y; V4[5] = V05.y;
            };
            load_d4(0);
#pragma unroll 1
            for (int s0 = 0; s0 < NS4; s0 += WR4) {
#pragma unroll
                for (int s = 0; s < WR4; ++s) {
                    const int ks = s0 + s;
                    transform4();
                    __builtin_amdgcn_sched_barrier(0);
                    load_a4(A4[(s + WR4 - 1) % WR4], W3f, ks + WR4 - 1);
                    load_d4(ks + 1);
                    if (ks < NS4) {
#pragma unroll
                        for (int p = 0; p < 6; ++p)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc4[i][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A4[s][p][i], V4[p], acc4[i][p], 0, 0, 0);
                    }
                    // (the stage's loads are placed by the compiler — it clusters them behind the stage's MFMAs: 7.97 ms per T = 4 sample; threaded
                    // between the MFMAs with sched_group_barrier as in the F(2,3) loop 8.16, all in front 8.06, in the first half 8.20.  Timing-only
                    // ablations of this loop mislead: with the weight ring or the inputs left constant the chip clocks differently, and removing
                    // work made the launch SLOWER.  Doubling the transform's 12 VALU operations costs 3 %: one VALU operation per stage = 0.25 %.)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (WINO == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int tr = 0; tr < 4; ++tr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accw[i][tr][r] = 0.f;
            const float* W3f = a.W3f[l];
            // inputs of pair p = l31 for the two k-steps of a half-group (channel 4 hg + 2 kk + khalf): the raw reads of the NEXT half-group
            // are in flight during this one's MFMAs and are transformed right in front of their own (transforming them a step ahead
            // into a second buffer costs 8 registers this loop does not have: half of the x rows went to scratch)
            float Vb[2][4], Dn[2][4];
            auto load_d = [&](int hg) {
                const float* bs = u_lds + (4 * hg + khalf) * U_LD + l31;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const float* rr = bs + 2 * kk * U_LD;
                    Dn[kk][0] = rr[0]; Dn[kk][2] = rr[1]; Dn[kk][1] = rr[33]; Dn[kk][3] = rr[34];     // u(2p-1), u(2p+1) | u(2p), u(2p+2)
                }
            };
            auto transform = [&]() {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    Vb[kk][0] = Dn[kk][0] - Dn[kk][2];
                    Vb[kk][1] = Dn[kk][1] + Dn[kk][2];
                    Vb[kk][2] = Dn[kk][2] - Dn[kk][1];
                    Vb[kk][3] = Dn[kk][1] - Dn[kk][3];
                }
            };
            load_d(0);
#if WINO_ABL & 2
            transform();
#endif
#pragma unroll 1
            for (int h0 = 0; h0 < NH; h0 += WR) {
#pragma unroll
                for (int s = 0; s < WR; ++s) {
                    const int hg = h0 + s;
                    // -DWINO_ABL=n (timing-only builds, wrong results): 1 = no weight loads in the loop, 2 = no LDS reads / input transform
#if !(WINO_ABL & 2)
                    transform();
                    __builtin_amdgcn_sched_barrier(0);      // all eight values in front of the MFMAs: computed just in time (a VALU + s_nop in front of every
                                                            // MFMA pair) the loop measured 3 % slower
#endif
#if !(WINO_ABL & 1)
                    load_aw(Aw[(s + WR - 1) % WR], W3f, hg + WR - 1);
#endif
#if !(WINO_ABL & 2)
                    load_d(hg + 1);           // (the read past the last half-group lands in the z buffer: discarded)
#endif
                    if (hg < NH) {
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int tr = 0; tr < 4; ++tr)
#pragma unroll
                                for (int i = 0; i < MT; ++i)
                                    accw[i][tr] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aw[s][i][tr >> 1][(tr & 1) * 2 + kk], Vb[kk][tr], accw[i][tr], 0, 0, 0);
                    }
#if WINO_THREAD
                    // the step's 4 weight loads and 4 LDS reads go BETWEEN its MFMAs (a wave has ~60 idle issue cycles behind each one): in one
                    // block in front of them, both waves of a SIMD reach their ~45 non-MFMA instructions together and the pipe idles
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // Round 5: one ring round = one 16-channel chunk (tap-major, two 8-channel halves: six k-groups = the ring depth), so every offset inside
            // a round is a compile-time constant — the weights through a buffer descriptor (lane offset constant, group offset scalar), u through
            // one LDS pointer per round with immediate offsets: no VALU address arithmetic and no clamps next to the MFMAs (the projection loop's
            // form; same loads of the same values in the same order, same bits)
            static_assert(RING == 6, "a ring round is one chunk of six k-groups");
            const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W3f[l]), 0, 3 * (C / 8) * (2 * C / 32) * 1024, 0x00020000);
            const int wvo3 = (w * MT * 64 + lane) * 16;
            // stage n of the round that starts at chunk q: chunk q + n / 6, tap (n % 6) >> 1, half n & 1
            auto load_a3 = [&](f32x4 (&dst)[MT], int q, int n) {
                const int qq = q + n / 6, ss = n % 6;
                const int group = (ss >> 1) * (C / 8) + 2 * qq + (ss & 1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w3rs, wvo3 + i * 1024, group * ((2 * C / 32) * 1024), 0));
            };
            float Bv[2][4][NT];
            const float* ub0 = u_lds + khalf * U_LD + l31;
            auto load_b3 = [&](float (&dst)[4][NT], const float* ub, int n) {
                const int ro = (n / 6) * 16 + (n & 1) * 8, col = (n % 6) >> 1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NT; ++j) dst[kk][j] = ub[(ro + 2 * kk) * U_LD + col + j * 32];
            };
            load_b3(Bv[0], ub0, 0);
            auto ring_round = [&](int q, auto first) {
                const float* ub = ub0 + q * 16 * U_LD;
#pragma unroll
                for (int s = 0; s < RING; ++s) {
                    load_a3(A[(s + RING - 1) % RING], q, s + RING - 1);
                    load_b3(Bv[(s + 1) & 1], ub, s + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (decltype(first)::value && s == 0) mma_group(first_t{}, A[s], Bv[s & 1], std::integral_constant<int, 0>{});
                    else mma_group(later_t{}, A[s], Bv[s & 1], std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            ring_round(0, first_t{});
#pragma unroll 1
            for (int q = 1; q < C / 16; ++q) ring_round(q, later_t{});
        }
        stamp(l, 2);
        if (!WINO) {
#pragma unroll
            for (int s = 0; s < RING - 1; ++s) load_ao(A[s], a.Wof[l], min(s, NGC - 1));   // output projection: same, before the gate
        }
        if constexpr (WINO == 2) {
            // output transform + gate: y0 = m0 + (m1 + m2) + (m3 + m4), y1 = (m1 - m2) + 2 (m3 - m4), y2 = (m1 + m2) + 4 (m3 + m4),
            // y3 = (m1 - m2) + 8 (m3 - m4) + m5 — all six transforms of a quad sit in the same lane and register; m-tiles 2 cb / 2 cb + 1 are
            // the sigmoid / tanh rows of the same 16 channels, and the lane writes its quad of z as one 16-byte store
            const float* b3 = a.b3[l];
            const int ln = opaque(lane), q4 = ln & 15, rb = ln >> 4;
            float bg[2][4], bf[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bg[cb][r] = ldg(b3, (unsigned)(64 * w + 32 * cb + 4 * rb + r));
                    bf[cb][r] = ldg(b3, (unsigned)(64 * w + 32 * cb + 16 + 4 * rb + r));
                }
            // (on ROW PAIRS: registers r, r + 1 of an accumulator are adjacent, so the output transform's adds and the gate's multiplies / adds run as
            //  v_pk_*_f32 on two rows at once — the same operations on the same values, half the VALU instructions next to the other waves' MFMAs)
            auto out4 = [&](int i, int h, f32x2 bias, f32x2 (&y)[4]) {
                auto P = [&](int p) { return f32x2{acc4[i][p][2 * h], acc4[i][p][2 * h + 1]}; };
                const f32x2 m0 = P(0), m1 = P(1), m2 = P(2), m3 = P(3), m4 = P(4), m5 = P(5);
                const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
                y[0] = ((m0 + s12) + s34) + bias;
                y[1] = __builtin_elementwise_fma(c2, d34, d12) + bias;
                y[2] = __builtin_elementwise_fma(c4, s34, s12) + bias;
                y[3] = (__builtin_elementwise_fma(c8, d34, d12) + m5) + bias;
            };
            auto gate2 = [&](f32x2 g, f32x2 f) -> f32x2 {      // cmtts_gate (gate.h) on two elements: the same multiplies, v_exp / v_rcp, adds and fma per element
                const f32x2 nl2e = {-1.44269504088896340736f, -1.44269504088896340736f}, l2e = {1.44269504088896340736f, 1.44269504088896340736f};
                const f32x2 one = {1.f, 1.f}, m2c = {-2.f, -2.f};
                const f32x2 ag = g * nl2e;
                const f32x2 eg = {__builtin_amdgcn_exp2f(ag.x), __builtin_amdgcn_exp2f(ag.y)};
                const f32x2 dg = one + eg;
                const f32x2 sg = {__builtin_amdgcn_rcpf(dg.x), __builtin_amdgcn_rcpf(dg.y)};
                const f32x2 af = (f + f) * l2e;
                const f32x2 ef = {__builtin_amdgcn_exp2f(af.x), __builtin_amdgcn_exp2f(af.y)};
                const f32x2 df = one + ef;
                const f32x2 rf = {__builtin_amdgcn_rcpf(df.x), __builtin_amdgcn_rcpf(df.y)};
                const f32x2 th = __builtin_elementwise_fma(m2c, rf, one);
                return sg * th;
            };
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 yg[4], yf[4];
                    out4(2 * cb, h, f32x2{bg[cb][2 * h], bg[cb][2 * h + 1]}, yg);
                    out4(2 * cb + 1, h, f32x2{bf[cb][2 * h], bf[cb][2 * h + 1]}, yf);
                    f32x4 z0, z1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2 zz = gate2(yg[e], yf[e]);
                        z0[e] = zz.x; z1[e] = zz.y;
                    }
                    float* zr = z_lds + (32 * w + 16 * cb + 4 * rb + 2 * h) * U_LD + 4 * q4;
                    *reinterpret_cast<f32x4*>(zr) = z0;
                    *reinterpret_cast<f32x4*>(zr + U_LD) = z1;
                }
        } else {   // gate: z goes to its own buffer, so a wave gates as soon as ITS k=3 conv is done (VALU under the
            // other waves' MFMAs); nobody reads z before barrier (3)
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
            if constexpr (WINO == 1) {
                // output transform: y(2p) = (m0 + m1) + m2, y(2p+1) = (m1 - m2) - m3; lane p writes the frame pair as one 8-byte store
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float ge = (accw[i][0][r] + accw[i][1][r]) + accw[i][2][r];
                        const float go = (accw[i][1][r] - accw[i][2][r]) - accw[i][3][r];
                        const float fe = (accw[i][0][r + 8] + accw[i][1][r + 8]) + accw[i][2][r + 8];
                        const float fo = (accw[i][1][r + 8] - accw[i][2][r + 8]) - accw[i][3][r + 8];
                        float2 zz;
                        zz.x = cmtts_gate(ge + bg[i][r], fe + bf[i][r]);
                        zz.y = cmtts_gate(go + bg[i][r], fo + bf[i][r]);
                        *reinterpret_cast<float2*>(z_lds + ((w * MT + i) * 16 + acc_row(r, ln)) * U_LD + 2 * (ln & 31)) = zz;
                    }
            } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float zv = cmtts_gate(acc[i][j][r] + bg[i][r], acc[i][j][r + 8] + bf[i][r]);
                        z_lds[((w * MT + i) * 16 + acc_row(r, ln)) * U_LD + j * 32 + (ln & 31)] = zv;
                    }
            }
        }
        float l2touch = 0.f;
        if constexpr (WINO) {      // eight accumulators + the gate's operands leave no room for the projection's ring before this point
#pragma unroll
            for (int s = 0; s < RING - 1; ++s) load_ao(A[s], a.Wof[l], min(s, NGC - 1));
            // the residual stream of this wave's elements (see the epilogue) is requested BEHIND the ring's first groups:
            // loads return in order, so the projection loop never waits for it and it has landed long before the loop's end
            const int ln = opaque(lane);
            if (l == 0) {      // x enters in the public [C][T] layout
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int t_c = min(t0 + j * 32 + (ln & 31), T - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[0][j][r] = ldg(x0_b, (unsigned)((mrow0 + acc_row(r, ln)) * T + t_c));
                }
            } else {
                const f32x4* px = reinterpret_cast<const f32x4*>(pst_b) + (w * 8) * 64 + ln;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 vx = px[(j * 4 + q) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) st[0][j][4 * q + e] = vx[e];
                    }
            }
#if WINO_L2PF
            if (more) {
                // pull the NEXT layer's transformed conv weights (2 MB) towards this XCD's L2 while the projection runs: the 32 workgroups of
                // an XCD (block id mod 8 — a placement guess that only costs speed when wrong) x 8 waves x 64 lanes touch one dword of each
                // of its 16384 128-byte lines.  Without it every stage of the conv loop's weight ring is a first touch that all 32 CUs,
                // in lockstep, wait out at Infinity-Cache latency (the loop ran at 82 % of its MFMA time).
                const int bid = blockIdx.x + gridDim.x * blockIdx.y;
                const unsigned line = (unsigned)((((bid >> 3) & 31) * NW + w) * 64 + ln);
                l2touch = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.W3f[l + 1]) + (size_t)line * 128u);
            }
#endif
        }
        stamp(l, 3);
        // (3) no barrier here since round 5.  The two waves of a SIMD do not finish the conv together (the older one is served first: 87 k
        // against 116 k cycles, -DDBG stamps), and behind a barrier the early wave idled ~27 k cycles per layer while the late one ran the
        // pipe alone at half its rate.  Now every wave announces its z rows (release store of the layer number) and the projection's
        // K loop — ordered by source wave — only waits for the block it is about to read: the early waves start the projection on the
        // early waves' rows under the late waves' conv tail.  Everything the barrier also ordered still holds: a wave writes the next u
        // (publish) only after it has consumed all eight blocks, i.e. after every wave has left the conv; z is rewritten behind barrier (1).
        if (lane == 0) __hip_atomic_store(zflag + w, l + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned zready = 0;      // bit v: wave v's z rows of this layer are written (wave-uniform)
        auto need_z = [&](int v) {
            if (v >= NW || ((zready >> v) & 1u)) return;
            unsigned spins = 0;
            for (;;) {
                const int f = __hip_atomic_load(zflag + (lane & (NW - 1)), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                zready = (unsigned)__ballot(f > l) & ((1u << NW) - 1u);
                if ((zready >> v) & 1u) break;
                if (++spins > SPIN_LIMIT) {      // cannot happen (every wave of the workgroup runs this code); bounded like every wait of this kernel
                    if (lane == 0 && a.tmo) *(volatile unsigned*)a.tmo = 1u;
                    zready = (1u << NW) - 1u;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        };
        stamp(l, 4);

        // =========================================================== phase C: output projection
        {
            constexpr int NG = NGC;
            // Round 5: no VALU address arithmetic in this loop either — the weights through a buffer descriptor (lane offset constant, k-group
            // offset scalar; groups past the end are out of range and read as zeros: no clamps), z through one LDS pointer per ring round
            // with immediate offsets.  The ring's look-ahead reads k-groups 32 .. 36 of z: up to 40 rows (~10.9 KB) past the z tile, of which only the
            // index table (2.6 KB) lies inside the workgroup's allocation — the rest is beyond it.  Those reads are never USED (the groups do
            // not exist: `it + s < NG` guards the MFMAs), and an LDS read beyond the workgroup's allocation is bounds-checked by the hardware and returns
            // zero; clamping them costs a v_min per read in the loop next to fp32 MFMAs, where every VALU instruction is issue time (0.25 % per instruction)
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wof[l]), 0, NG * (2 * C / 32) * 256 * 4, 0x00020000);
            const int wvo = (w * 64 + lane) * 16;
            auto load_aob = [&](f32x4 (&dst)[MT], int group) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo + i * (NW * 1024), group * ((2 * C / 32) * 1024), 0));
            };
            float Bv[2][4][NT];
            const float* zb0 = z_lds + khalf * U_LD + l31;
            auto load_bz = [&](float (&dst)[4][NT], const float* zb, int goff) {      // goff: groups beyond zb's
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NT; ++j) dst[kk][j] = zb[(goff * 8 + 2 * kk) * U_LD + j * 32];
            };
            need_z(0);
            load_bz(Bv[0], zb0, 0);
            auto ring_round = [&](int it, auto first, auto i0) {
                const float* zb = zb0 + it * 8 * U_LD;
#pragma unroll
                for (int s = 0; s < RING; ++s) {
                    load_aob(A[(s + RING - 1) % RING], it + s + RING - 1);
                    if (((it + s + 1) & 3) == 0) need_z((it + s + 1) >> 2);      // k-group it + s + 1 opens the z rows of the next wave
                    load_bz(Bv[(s + 1) & 1], zb, s + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (it + s < NG) {       // NG need not be a multiple of the ring depth
                        if (decltype(first)::value && s == 0) mma_group(first_t{}, A[s], Bv[s & 1], i0);
                        else mma_group(later_t{}, A[s], Bv[s & 1], i0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (more) {
                ring_round(0, first_t{}, std::integral_constant<int, 0>{});
#pragma unroll 1
                for (int it = RING; it < NG; it += RING) ring_round(it, later_t{}, std::integral_constant<int, 0>{});
            } else {      // the last layer: only the skip half of the projection (its x' has no reader): half of this loop's MFMAs, 0.6 % of a launch
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[0][j] = zero16;
                ring_round(0, first_t{}, std::integral_constant<int, 1>{});
#pragma unroll 1
                for (int it = RING; it < NG; it += RING) ring_round(it, later_t{}, std::integral_constant<int, 1>{});
            }
        }
        stamp(l, 5);
        // ---- epilogue in registers: tile 0: x' = (o[:C] + (x + d)) / sqrt(2); tile 1: skip (+)= o[C:]
        {
            const float* bo = a.bo[l];
            const float* dl = dv_b + (long)l * C;
            const int ln = opaque(lane);
            float bor[MT][16], ddr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {            // all loads in flight before the first use
                bor[0][r] = ldg(bo, (unsigned)(mrow0 + acc_row(r, ln)));
                bor[1][r] = ldg(bo, (unsigned)(C + mrow0 + acc_row(r, ln)));
                ddr[r] = ldg(dl, (unsigned)(mrow0 + acc_row(r, ln)));
            }
            // WINO: eight accumulators leave the conv loop no room for all 64 registers of state: the skip sum stays resident, the residual
            // stream x waits in memory (`xst`: L2 / Infinity Cache resident) between layers — read-modify-write by the lane that owns
            // the element (its own earlier stores: program order), requested ahead of the projection loop (st[0] holds the loaded x
            // here).  x' stays in registers for the publish phase below.
            if (WINO) asm volatile("" ::"v"(l2touch));      // (the L2-warming load's destination stays reserved until here)
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
        stamp(l, 6);
        pstamp(l, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand the edge columns of x' to the neighbouring tiles first (their latency is what the neighbours wait for)
        const float* dpn = dp_b + (long)(l + 1) * C;
        const unsigned tag = (unsigned)l + 1;
        unsigned long long* hbase = halo_g + ((((long)(l & 1) * B_g + b) * tiles_g) * 2) * C;    // [parity][b][tile][side][C]
        {
            const int ln = opaque(lane), c31 = ln & 31;
            // column 0 of this tile -> slot (tile, side 0); column FN-1 -> slot (tile, side 1).  The two columns sit in four lanes (16 registers
            // each): stored from there they were 32 store instructions of two active lanes — ~100 cycles of issue apiece, 3-5 k per wave and
            // layer at the head of the phase the neighbours wait for (round 5, -DPUB_STAMP).  Through 64 floats of LDS (a wave's own LDS
            // operations execute in order: no barrier) every lane owns one granule and the wave stores them with ONE coalesced instruction.
            float* edge = reinterpret_cast<float*>(smem + 2 * C * U_LD) + (FN + 2) * 2 + w * 64;
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
        // ---- halo columns of the next layer's u.  Every wave fetches the two halo entries of ITS OWN 32 rows (lanes 0-31: left halo
        // frame t0 - 1, lanes 32-63: right halo frame t0 + FN): one cp value and one granule per lane, requested here — before the wave's
        // own u rows — and checked after them.  (Until round 4 the last two waves fetched all 256 rows of one side each, four granules
        // per lane, after their own rows: the phase stamps showed them 5-7 k cycles behind the other six waves at the layer barrier.)
        const int hside = opaque(lane) >> 5, hm = mrow0 + (opaque(lane) & 31);
        const int hth = hside ? t0 + FN : t0 - 1;
        const bool hinside = hth >= 0 && hth < Tc;
        // neighbour's slot: its right edge (side 1) feeds our left halo, its left edge (side 0) our right halo
        const unsigned long long* hg = hbase + ((long)(hinside ? (hside ? tile + 1 : tile - 1) : tile) * 2 + (hside ? 0 : 1)) * C + hm;
        float hcp = 0.f;
        int hph = 0, hix = 0;
        {
            const int thc = min(max(hth, 0), T - 1);
            if (FACT) { hph = idx_lds[2 * (hside ? FN + 1 : 0)]; hix = idx_lds[2 * (hside ? FN + 1 : 0) + 1]; }   // (its two gathers ride with the wave's below)
            else hcp = (cp_b + (long)(l + 1) * C * T)[(unsigned)(hm * T + thc)];
        }
        unsigned long long hv = __hip_atomic_load((gu64*)hg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pstamp(l, 1);
        // ---- next layer's u rows of this wave: cp (L2-warm, accumulator layout) + (x' + dp)
        {
            const float* cpn = cp_b + (long)(l + 1) * C * T;
            const int ln = opaque(lane), c31 = ln & 31;
            f32x16 cpc[NT];
            float dpr[16];
            if constexpr (FACT) {
                // Round 5: every load of this phase is ISSUED BY HAND before the first wait — the frame indices of both n-tiles and of the
                // halo frame first (one round trip), then the wave's 64 gathers, the halo entry's two and the 16 per-row dp values (one more).
                // Left to the compiler (at 253 of 256 registers) every element became mad_i64 / shift / add / load / s_waitcnt vmcnt(0): 32
                // dependent L2 round trips per wave and layer, then the index loads of the second n-tile behind the first tile's gathers, then
                // the dp loads one by one between the LDS stores — most of the publish phase's 15-20 k cycles.  The byte offset of an element
                // is formed in 32 bits in the register the load then overwrites (saddr form: uniform layer base + lane offset); row
                // (r & 3) + 8 (r >> 2) of the tile is a uniform multiple of the row pitch.  Same values as cp_fact: the same two loads, the
                // same add.
                const float* p1l = p1_b + (long)(l + 1) * C * ldp;
                const float* p2l = a.p2 + (long)(l + 1) * C * a.ld2;
                // the wave's own elements come from the channel-contiguous copies (persist_args.h: p1t [NL][ldp][C], p2t [NL][ld2][C]): a lane's 16 rows
                // of a tile are four runs of four consecutive channels = four 16-byte loads per factor and n-tile (the row-major tables took 16
                // scattered dwords each, and the phase was bound by the cache lines its gathers touch)
                const float* p1tl = p1t_b + (long)(l + 1) * ldp * C;
                const float* p2tl = a.p2t + (long)(l + 1) * a.ld2 * C;
                int phj[NT], ixj[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) { phj[j] = idx_lds[2 * (1 + j * 32 + c31)]; ixj[j] = idx_lds[2 * (1 + j * 32 + c31) + 1]; }
                // Round 6: the same 34 loads as compiler-VISIBLE buffer loads (one descriptor per table, the lane's byte offset in a
                // VGPR, the run's offset an immediate) instead of one `asm volatile` per load with a hand-written s_waitcnt behind them: the compiler
                // knew nothing of those loads' latency and was free to copy or spill their destination registers before the wait.  A buffer load
                // needs no address arithmetic (what had serialised the plain-pointer form at 253 registers), and the compiler places the waits.
                // Same loads of the same values, the same adds => the same bits (tests: test_cond_factored, FACT == expanded factors bitwise).
                const __amdgpu_buffer_rsrc_t r1t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p1tl), 0, ldp * C * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t r2t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p2tl), 0, a.ld2 * C * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p1l), 0, C * ldp * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p2l), 0, C * a.ld2 * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dpn), 0, C * 4, 0x00020000);
                f32x4 g1[NT][4], g2[NT][4];
                const float h1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (int)((unsigned)(hm * ldp + (hph > 0 ? hph - 1 : 0)) * 4u), 0, 0));
                const float h2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, (int)((unsigned)(hm * a.ld2 + hix) * 4u), 0, 0));
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int o1 = (int)((unsigned)((phj[j] > 0 ? phj[j] - 1 : 0) * C + mrow0 + 4 * (ln >> 5)) * 4u);
                    const int o2 = (int)((unsigned)(ixj[j] * C + mrow0 + 4 * (ln >> 5)) * 4u);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {      // rows 8 q + 4 khalf + {0, 1, 2, 3} = accumulator registers 4 q .. 4 q + 3
                        g1[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1t, o1 + q * 32, 0, 0));
                        g2[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2t, o2 + q * 32, 0, 0));
                    }
                }
                {
                    const int od = (int)((unsigned)(mrow0 + 4 * (ln >> 5)) * 4u);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dpr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, od + ((r & 3) + 8 * (r >> 2)) * 4, 0, 0));
                }
                pstamp(l, 2);
                hcp = (hph > 0 ? h1 : 0.f) + h2;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) cpc[j][4 * q + e] = (phj[j] > 0 ? g1[j][q][e] : 0.f) + g2[j][q][e];
                    }
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int t_c = min(t0 + j * 32 + c31, T - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) cpc[j][r] = ldg(cpn, (unsigned)((mrow0 + acc_row(r, ln)) * T + t_c));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dpr[r] = ldg(dpn, (unsigned)(mrow0 + acc_row(r, ln)));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = t0 + j * 32 + c31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow0 + acc_row(r, ln);
                    const float uv = cpc[j][r] + (st[0][j][r] + dpr[r]);
                    u_lds[m * U_LD + uidx(j * 32 + c31)] = t < Tc ? uv : 0.f;
                }
            }
        }
        pstamp(l, 3);
        {   // the halo entries: wait for the neighbours' tags (lanes without a neighbour frame never wait)
            if (!gave_up) {
                unsigned spins = 0;
                while (!__all(!hinside || (unsigned)(hv >> 32) == tag)) {
                    if (++spins > SPIN_LIMIT) {      // wave-uniform: a neighbour never arrived
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
            u_lds[hm * U_LD + uidx(hside ? FN : -1)] = hinside ? uh : 0.f;
        }
        pstamp(l, 4);
        if constexpr (WINO) {      // x' goes back to memory LAST: in front of the publish phase's loads, every wait of that phase also waited for
                                   // the acknowledgement of these stores (one counter for loads and stores)
            f32x4* px = reinterpret_cast<f32x4*>(pst_b) + (w * 8) * 64 + opaque(lane);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 vx;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vx[e] = st[0][j][4 * q + e];
                    px[(j * 4 + q) * 64] = vx;
                }
        }
        pstamp(l, 5);
        stamp(l, 7);
    }

    if (a.tail) {   // skip head + post-scaling in-kernel (persist_tail.h); the u buffer is free since barrier (3), z after barrier (A) inside
        persist_tail::run(a, smem, smem + C * U_LD, st[1], w, lane, b, t0, T, RAGGED ? a.grp[gi].xold : a.xold,
                          RAGGED ? a.grp[gi].noise : a.noise, RAGGED ? a.grp[gi].out : a.out, Tc);
    } else {   // ---- the skip sum leaves the chip once
        float* skip = (RAGGED ? a.grp[gi].skip : a.skip) + (long)b * C * T;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = t0 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t < Tc) skip[(unsigned)((mrow0 + acc_row(r, lane)) * T + t)] = st[1][j][r];
        }
    }
}

long long* g_pdbg = nullptr;
// -1 = automatic (default), 0 / 1 = forced.  Automatic: once the process has a communicator (RCCL kernels may then share the GPU
// with the persistent grid) the FIRST launch of every (kernel variant, grid) goes through hipLaunchCooperativeKernel — the runtime
// validates that the whole grid can be co-resident and fails the launch otherwise — and later launches of a validated shape are
// plain.  Why not every launch: with RCCL loaded a cooperative launch drains EVERY queue of the device first (measured on MI355X:
// 12.80 -> 15.9 ms per bench step, persistent launch 2.71 -> 2.93 ms, profiles/r03_cooperative.md), which also serialises the
// branch streams; co-residency is a static property of (kernel, grid, LDS), the ordering against RCCL's kernels is the host's
// (the all-gather is completed on the stream before the next persistent launch), and bounded waits + NaN poisoning remain.
int g_coop = -1;
int g_process_group = 0;
// per kernel instance: the LARGEST grid (workgroups) whose co-residency the runtime has confirmed.  Instances differ in registers and
// threads, so every one has its own record: fp32 uniform 0..7 = (direct | 8-wave F(2,3) | 8-wave F(4,3) | one wave per SIMD) x (cp | factors),
// fp32 ragged 8..15 likewise, the 16-bit modes 16 + MODE (shared slots let one kernel's validation vouch for another)
constexpr int N_VARIANTS = 24;
int g_validated_wg[N_VARIANTS] = {};

}  // namespace

extern "C" int cmtts_persist_set_cooperative(int on) { const int p = g_coop; if (on >= -1 && on <= 1) g_coop = on; return p; }
extern "C" int cmtts_persist_note_process_group(int on) { const int p = g_process_group; if (on == 0 || on == 1) g_process_group = on; return p; }
// Should this launch of kernel instance `variant` (the table above) with grid (gx, gy) be cooperative?
// Co-residency is a property of (kernel instance, LDS, workgroup COUNT): every grid no larger than one the runtime has accepted is
// resident too, so only the maximum is remembered (a ring of exact shapes made most ragged launches — n_wg changes
// with every trim — cooperative again, +24 % per step with RCCL loaded).
extern "C" int cmtts_persist_cooperative(int variant, int gx, int gy) {
    if (g_coop >= 0) return g_coop;
    if (!g_process_group) return 0;
    const long wg = (long)gx * (gy > 0 ? gy : 1);
    if (variant < 0 || variant >= N_VARIANTS) return 1;
    return wg > g_validated_wg[variant] ? 1 : 0;
}
// Record a grid only AFTER hipLaunchCooperativeKernel has returned hipSuccess for it.
extern "C" void cmtts_persist_validated(int variant, int gx, int gy) {
    const long wg = (long)gx * (gy > 0 ? gy : 1);
    if (variant >= 0 && variant < N_VARIANTS && wg > g_validated_wg[variant]) g_validated_wg[variant] = (int)wg;
}

extern "C" void cmtts_persist_set_debug(long long* dbg) { g_pdbg = dbg; }
extern "C" long long* cmtts_persist_get_debug(void) { return g_pdbg; }

extern "C" int cmtts_persist_chunks(int B, int T, int max_blocks) {
    const int tiles = (T + FN - 1) / FN;
    if (tiles > max_blocks) return 0;
    const int per_launch = max_blocks / tiles;
    return (B + per_launch - 1) / per_launch;
}

// Workgroups the largest launch of one call keeps resident (0 = the call would not take the persistent path): what the
// caller's cross-stream guard has to reserve.  Mirrors the decisions of cmtts_launch_denoiser_persist(_lp).
extern "C" int cmtts_persist_plan(int B, int T, int NL, int max_blocks, int force) {
    const int tiles = (T + FN - 1) / FN;
    if (B < 1 || NL < 1 || NL > PERSIST_MAX_LAYERS || tiles > max_blocks || (long)C * T >= (1L << 30)) return 0;
    if (!force && (long)tiles * B * 2 <= (long)max_blocks) return 0;
    const int per_launch = max_blocks / tiles;
    const int nchunks = (B + per_launch - 1) / per_launch;
    const int bc = (B + nchunks - 1) / nchunks;
    return tiles * (bc < B ? bc : B);
}

extern "C" size_t cmtts_persist_state_floats(int B, int T) {
    const long tiles = (T + FN - 1) / FN;
    return (size_t)B * tiles * (NW * 8 * 64 * 4);
}

extern "C" size_t cmtts_persist_halo_bytes(int B, int T) {
    const long tiles = (T + FN - 1) / FN;
    return (size_t)2 * B * tiles * 2 * C * sizeof(unsigned long long);
}

// Returns 0, -2 (shape not supported: use the per-layer kernels) or -3 (HIP error).
extern "C" int cmtts_launch_denoiser_persist(const PersistArgs* a_in, int max_blocks, int force, void* stream_) {
    PersistArgs a = *a_in;
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles = (a.T + FN - 1) / FN;
    if (a.NL < 1 || a.NL > PERSIST_MAX_LAYERS || tiles > max_blocks || (long)C * a.T >= (1L << 30)) return -2;
    // a workgroup walks the whole stack alone (~128 us per layer): while the per-layer kernels can still spread the batch
    // over the chip in ONE round of 32-frame tiles (tiles <= CUs / 2) they finish sooner (2.4 vs 3.2 ms per evaluation);
    // above that they need two rounds (4.0 ms) and the persistent stack wins (measured, tools/mid_bench.py)
    if (!force && (long)tiles * a.B * 2 <= (long)max_blocks) return -2;
    a.tiles = tiles;
    a.dbg = g_pdbg;
    if ((a.wino == 1 || a.wino == 3) && !a.xst) return -2;     // the 8-wave Winograd instances keep the residual stream in `xst` between layers
    const int wsel = a.wino == 3 ? 2 : a.wino ? 1 : 0;         // instance: direct | F(2,3) | F(4,3)
    if (a.wino == 2) return -2;               // (round 5's one-wave-per-SIMD stack: measured slower, out of the product build since round 6 — tools/attic/)
    const int threads = 64 * NW;
    // instance table: [dbg][fact][wino]
#define KFN(D, R, F, W) reinterpret_cast<const void*>(denoiser_persist_kernel<D, R, F, W>)
    static const void* const kfns[2][2][3] = {
        {{KFN(false, false, false, 0), KFN(false, false, false, 1), KFN(false, false, false, 2)}, {KFN(false, false, true, 0), KFN(false, false, true, 1), KFN(false, false, true, 2)}},
        {{KFN(true, false, false, 0), KFN(true, false, false, 1), KFN(true, false, false, 2)}, {KFN(true, false, true, 0), KFN(true, false, true, 1), KFN(true, false, true, 2)}}};
    static bool attr_set = false;
    const size_t lds = (size_t)2 * C * U_LD * sizeof(float) + IDX_LDS_BYTES;
    if (!attr_set) {
        for (int d = 0; d < 2; ++d)
            for (int f = 0; f < 2; ++f) {
                for (int wn = 0; wn < 3; ++wn)
                    if (hipFuncSetAttribute(kfns[d][f][wn], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
            }
        attr_set = true;
    }
    if (a.fact && (!a.p1 || !a.p2 || !a.mel2ph || !a.pidx || a.ldp < 1 || a.ld2 < 1)) return -2;
    if (a.fact && (!a.p1t || !a.p2t)) return -2;      // the 8-wave FACT instances gather the channel-contiguous tables in their publish phase
    // every granule tag must be stale (0) when a launch starts
    if (!a.halo_zeroed && hipMemsetAsync(a.halo, 0, cmtts_persist_halo_bytes(a.B, a.T), stream) != hipSuccess) return -3;
    // utterance chunks: all workgroups of a launch must be resident (one per CU); chunks are balanced so that the last
    // one does not run on a sliver of the chip
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
        if (a.xst) c.xst = a.xst + (long)b0 * tiles * (NW * 8 * 64 * 4);      // [B][tiles][16384]
        c.halo = a.halo + (long)b0 * tiles * 2 * C;      // [parity][B][tiles][2][C]: the parity stride keeps a.B
        if (a.fact) {
            c.p1 = a.p1 + (long)b0 * a.NL * C * a.ldp;
            if (a.p1t) c.p1t = a.p1t + (long)b0 * a.NL * C * a.ldp;
            c.mel2ph = a.mel2ph + (long)b0 * a.T;
            c.pidx = a.pidx + (long)b0 * a.T;
        }
        if (a.tail) {
            const long off = (long)b0 * a.T * a.n_mels;
            c.xold = a.xold ? a.xold + off : nullptr;
            c.noise = a.noise ? a.noise + off : nullptr;
            c.out = a.out + off;
        }
        const void* kfn = kfns[a.dbg ? 1 : 0][a.fact ? 1 : 0][wsel];
        void* params[] = {(void*)&c};
        const int variant = wsel * 2 + (a.fact ? 1 : 0);        // one record per kernel instance (persist_args.h)
        if (!a.dbg && cmtts_persist_cooperative(variant, tiles, nb)) {
            if (hipLaunchCooperativeKernel(kfn, dim3(tiles, nb), dim3(threads), params, (unsigned)lds, stream) != hipSuccess) return -3;
            cmtts_persist_validated(variant, tiles, nb);
        } else if (hipLaunchKernel(kfn, dim3(tiles, nb), dim3(threads), params, lds, stream) != hipSuccess) return -3;
        if (hipGetLastError() != hipSuccess) return -3;
    }
    return 0;
}

// Ragged form: one workgroup per descriptor (persist_args.h).  The caller has cleared every group's halo granules on `stream`.
extern "C" int cmtts_launch_denoiser_persist_ragged(const PersistArgs* a_in, void* stream_) {
    const PersistArgs& a = *a_in;
    hipStream_t stream = (hipStream_t)stream_;
    if (a.NL < 1 || a.NL > PERSIST_MAX_LAYERS || a.n_groups < 1 || a.n_groups > PERSIST_MAX_GROUPS || a.n_wg < 1 || a.n_wg > PERSIST_MAX_WG)
        return -2;
    for (int g = 0; g < a.n_groups; ++g)
        if ((long)C * a.grp[g].T >= (1L << 30) || a.grp[g].tiles > 127 || a.grp[g].B > 1023) return -2;
    // instance table: [fact][wino]
    static const void* const kfns[2][3] = {{KFN(false, true, false, 0), KFN(false, true, false, 1), KFN(false, true, false, 2)},
                                           {KFN(false, true, true, 0), KFN(false, true, true, 1), KFN(false, true, true, 2)}};
    static bool attr_set = false;
    const size_t lds = (size_t)2 * C * U_LD * sizeof(float) + IDX_LDS_BYTES;
    if (!attr_set) {
        for (int f = 0; f < 2; ++f) {
            for (int wn = 0; wn < 3; ++wn)
                if (hipFuncSetAttribute(kfns[f][wn], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
        }
        attr_set = true;
    }
    if (a.fact) {      // every group brings its factors, or none does
        if (!a.p2 || a.ld2 < 1) return -2;
        for (int g = 0; g < a.n_groups; ++g)
            if (a.grp[g].B > 0 && (!a.grp[g].p1 || !a.grp[g].mel2ph || !a.grp[g].pidx || a.grp[g].ldp < 1)) return -2;
        if (!a.p2t) return -2;
        for (int g = 0; g < a.n_groups; ++g)
            if (a.grp[g].B > 0 && !a.grp[g].p1t) return -2;
    }
    if (a.wino == 1 || a.wino == 3)   // the 8-wave Winograd instances keep the residual stream of every group in its `xst` buffer
        for (int g = 0; g < a.n_groups; ++g)
            if (a.grp[g].B > 0 && !a.grp[g].xst) return -2;
    if (a.wino == 2) return -2;
    const int threads = 64 * NW;
    const int wsel = a.wino == 3 ? 2 : a.wino ? 1 : 0;
    const void* kfn = kfns[a.fact ? 1 : 0][wsel];
    const int variant = 8 + wsel * 2 + (a.fact ? 1 : 0);      // one record per kernel instance
    void* params[] = {(void*)a_in};
    if (cmtts_persist_cooperative(variant, a.n_wg, -1)) {
        if (hipLaunchCooperativeKernel(kfn, dim3(a.n_wg), dim3(threads), params, (unsigned)lds, stream) != hipSuccess) return -3;
        cmtts_persist_validated(variant, a.n_wg, -1);
    } else if (hipLaunchKernel(kfn, dim3(a.n_wg), dim3(threads), params, lds, stream) != hipSuccess) return -3;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
