// Persistent denoiser stack for gfx950, ONE WAVE PER SIMD (round 5): all residual layers of Denoiser.forward
// (model/modules.py:626-633, ResidualBlock model/blocks.py:667-686) in one launch, the gated k = 3 conv in its Winograd F(2,3)
// form (denoiser_persist.hip explains the protocol: LDS u / z buffers, {tag, value} halo granules, bounded waits, the in-kernel tail;
// this file keeps all of it and changes WHO holds WHAT):
//
//   * a workgroup is 4 wave64 — one per SIMD, 512 registers each (256 VGPR + 256 AGPR) — instead of 8;
//   * conv: a wave owns 4 packed m-tiles (64 z rows) x 4 transforms over the tile's 32 frame pairs = 16 accumulators = all 256
//     AGPRs; one transformed input value (4 per k-step and lane) now feeds FOUR MFMAs instead of two, and one 16-byte weight load per
//     m-tile and k-step brings the four transforms of (row, channel) (packing: cmtts_api.hip to_wino4_fragments): 10 non-MFMA
//     instructions per 16 MFMAs where the 8-wave form has 15;
//   * output projection: 4 m-tiles (two of the residual half, two of the skip half) x 2 n-tiles = 8 accumulators;
//   * state: every wave keeps its 64 rows of the residual stream x AND its 64 rows of the skip sum in VGPRs for the whole stack (128
//     registers, MFMA accumulator layout) — the 8-wave Winograd form had to park x in an L2 buffer between layers (`xst`: 34 MB
//     written and re-read per layer at the bench shape).  Per layer the fabric sees the layer's weights, the conditioner factors
//     and the halo granules, nothing else.
//
// Same arithmetic per element as the 8-wave Winograd instances (transforms, accumulation order over the channels, gate, epilogue):
// the two forms agree bit for bit (tests/test_gpu_parity.py::test_persist4_matches_8wave_winograd); against the direct form the
// Winograd bound of tests/test_gpu_precision.py applies.
#include <hip/hip_runtime.h>
#include "gate.h"
#include "persist_args.h"
#include "persist_tail.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;

#ifndef P4_ABL
#define P4_ABL 0           // timing-only builds (wrong results): 1 = no weight loads in the conv loop, 2 = no LDS reads / input transform
#endif
#ifndef P4_CMT
#define P4_CMT 2           // conv m-tiles per wave and pass: 2 (two passes of 8 accumulators) or 4 (one pass of 16 = all AGPRs)
#endif
#ifndef P4_WR
#define P4_WR (16 / P4_CMT)
#endif
#ifndef P4_SCHED
#define P4_SCHED 1         // 0: no sched_group_barrier pattern in the conv loop (the compiler places the loads)
#endif

namespace {

constexpr int C = 256;
constexpr int NW = 4;           // waves per workgroup: one per SIMD
constexpr int TPW = 2;          // 32-row tiles of x (and of the skip sum) per wave
constexpr int CMT = P4_CMT;          // packed conv m-tiles per wave and PASS (16 gate + 16 filter rows each); two passes per layer
constexpr int WR = P4_WR;           // conv weight ring: k-steps in use + in flight (8 registers each)
constexpr int RP = 4;           // projection weight ring: k-groups of 4 k-steps (16 registers each)
constexpr int FN = 64;
constexpr int NPASS = 4 / CMT;
constexpr int NKS = C / 2;      // k-steps of one Winograd transform (2 channels each)
constexpr int NT = FN / 32;
constexpr int U_LD = FN + 4;
constexpr unsigned SPIN_LIMIT = 1u << 20;

__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float ldg(const float* base, unsigned idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 4u));
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
// An accumulator element out of its AGPR at THIS point of the program: left to the compiler, every accumulator of a finished MFMA loop is
// copied to VGPRs at the top of the block that follows (256 + 128 copies) and the resident state goes to scratch.
__device__ __forceinline__ float acc_rd(float v) {
    float r;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(v));
    return r;
}
// MFMA results are read by VALU instructions the compiler does not see (acc_rd): wait out the last 16-pass MFMA by hand
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }
__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned tag, float v) {
    __hip_atomic_store((gu64*)g, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// DBG / RAGGED / FACT: as denoiser_persist.hip (cycle stamps of the middle layer; tile-descriptor grid over several frame buckets;
// conditioner projections gathered from their factors).
template <bool DBG, bool RAGGED, bool FACT>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void denoiser_persist4_kernel(const PersistArgs a) {
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
    const int mrow0 = w * (32 * TPW);                  // this wave's 64 rows of x and of the skip sum
    const float* p1_b = nullptr;
    const long long *m2p_b = nullptr, *pix_b = nullptr;
    int ldp = 0, Lph = 0;
    if (FACT) {
        p1_b = (RAGGED ? a.grp[gi].p1 : a.p1);
        ldp = RAGGED ? a.grp[gi].ldp : a.ldp;
        Lph = RAGGED ? a.grp[gi].Lph : a.Lph;
        p1_b += (long)b * a.NL * C * ldp;
        m2p_b = (RAGGED ? a.grp[gi].mel2ph : a.mel2ph) + (long)b * T;
        pix_b = (RAGGED ? a.grp[gi].pidx : a.pidx) + (long)b * T;
    }
    auto frame_idx = [&](int t_c, int& ph, int& ix) {
        const long long p = m2p_b[t_c], q = pix_b[t_c];
        ph = (int)(p > Lph ? Lph : p);
        ix = q < 0 ? 0 : (q >= a.ld2 ? a.ld2 - 1 : (int)q);
    };
    auto cp_fact = [&](int row, int ph, int ix) -> float {      // row = l * C + m; cond_expand_kernel's expression
        const float av = p1_b[(long)row * ldp + (ph > 0 ? ph - 1 : 0)];
        const float qv = a.p2[(long)row * a.ld2 + ix];
        return (ph > 0 ? av : 0.f) + qv;
    };
    // column of frame f (-1 .. FN) within a u row: odd frames first, then even frames (the four inputs of a frame pair are unit-stride reads)
    auto uidx = [](int f) { return (f & 1) ? (f + 1) >> 1 : 33 + (f >> 1); };

    // ---- layer-0 staging: u = cp + (x + dp), halo columns straight from x0
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
#pragma unroll 1
        for (int h = tid; h < 2 * C; h += 64 * NW) {
            const int m = h & (C - 1);
            const bool right = h >= C;
            const int th = right ? t0 + FN : t0 - 1;
            const int thc = min(max(th, 0), T - 1);
            float cph0;
            if (FACT) { int ph, ix; frame_idx(thc, ph, ix); cph0 = cp_fact(m, ph, ix); }
            else cph0 = cp_b[(unsigned)(m * T + thc)];
            const float uh = cph0 + (xin[(unsigned)(m * T + thc)] + dp_b[m]);
            smem[m * U_LD + uidx(right ? FN : -1)] = (th >= 0 && th < Tc) ? uh : 0.f;
        }
    }
    // resident state, MFMA C layout: [i][j][r] = row mrow0 + 32 i + acc_row(r), frame j * 32 + l31
    f32x16 xs[TPW][NT], sk[TPW][NT];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t_c = min(t0 + j * 32 + l31, T - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xs[i][j][r] = ldg(x0_b, (unsigned)((mrow0 + 32 * i + acc_row(r, lane)) * T + t_c));
                sk[i][j][r] = 0.f;
            }
        }

    // FACT: the factor indices of this lane's frames are the same for every layer: own frames (j = 0, 1) and its halo frame
    int fph[NT] = {}, fix[NT] = {}, hph = 0, hix = 0;
    if (FACT) {
#pragma unroll
        for (int j = 0; j < NT; ++j) frame_idx(min(t0 + j * 32 + l31, T - 1), fph[j], fix[j]);
        const int hth0 = khalf ? t0 + FN : t0 - 1;
        frame_idx(min(max(hth0, 0), T - 1), hph, hix);
    }
    const int bid_dbg = blockIdx.x + gridDim.x * blockIdx.y;
    auto stamp = [&](int l, int slot) {
        if (DBG && a.dbg && l == a.NL / 2 && lane == 0) a.dbg[((long)bid_dbg * NW + w) * 8 + slot] = (long long)__builtin_readcyclecounter();
    };
    bool gave_up = false;
    for (int l = 0; l < a.NL; ++l) {
        float* u_lds = smem;                  // u of the current layer, then assembled in place for the next one
        float* z_lds = smem + C * U_LD;       // gate output
        const bool more = l + 1 < a.NL;
        constexpr int NGC = C / 8;            // k-groups of the output projection

        // ---- conv weights: this wave's stream of the layer, [pass][k-step][m-tile of the pass][64 lanes][4 transforms] (element tr of a
        // fragment = transform tr of (row 32 mt + l31, channel 2 ks + khalf); packing: cmtts_api.hip to_wino4_fragments), read through a
        // ring of WR stages that runs straight through both passes; uniform base + 32-bit lane offset.  The ring reads up to WR - 1 stages
        // past the stream's end (the next wave's stream / the array's padding).  The first stages do not depend on u: requested before the
        // barrier.
        f32x4 Aw[WR][CMT];
        const char* W3s = reinterpret_cast<const char*>(a.W3f[l]) + (size_t)w * (NPASS * NKS * CMT * 64 * 16);
        auto load_aw = [&](f32x4 (&dst)[CMT], int q) {
            const char* base = W3s + (size_t)q * (CMT * 64 * 16);
#pragma unroll
            for (int i = 0; i < CMT; ++i)
                dst[i] = *reinterpret_cast<const f32x4*>(base + (size_t)((unsigned)((i * 64 + lane) * 16)));
        };
#pragma unroll
        for (int s = 0; s < WR - 1; ++s) load_aw(Aw[s], s);
        stamp(l, 0);
        __syncthreads();   // (1) u of layer l complete (interior, halo columns)
        stamp(l, 1);
        if (!FACT && more) {
            // pull the next layer's cp tile (256 rows x 256 B) towards this XCD's L2 now: one dword per 128-B line
            const float* cpn = cp_b + (long)(l + 1) * C * T;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tl = opaque(tid) + h * 64 * NW;
                const float warm = cpn[(unsigned)((tl >> 1) * T + min(t0 + (tl & 1) * 32, T - 1))];
                asm volatile("" ::"v"(warm));
            }
        }

        // =========================================================== phase B: gated k = 3 conv, Winograd F(2,3), in two passes of CMT
        // m-tiles (8 accumulators = 128 AGPRs each: with all 16 accumulators of a wave's four m-tiles at once the AGPRs are full and the
        // 128 registers of resident state do not fit the VGPRs next to the weight ring — they went to scratch)
#pragma unroll 1
        for (int ps = 0; ps < NPASS; ++ps) {
            f32x16 accw[CMT][4];                   // m-tile x transform over the tile's 32 frame pairs
#pragma unroll
            for (int i = 0; i < CMT; ++i)
#pragma unroll
                for (int tr = 0; tr < 4; ++tr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accw[i][tr][r] = 0.f;
            {
                // inputs of pair p = l31, channel 2 ks + khalf: d0..d3 = u(2p-1), u(2p), u(2p+1), u(2p+2); the raw reads run two k-steps
                // ahead of their MFMAs, the transform one
                float Dn[2][4], Vb[2][4];
                auto load_d = [&](float (&d)[4], int ks) {
                    const float* rr = u_lds + (2 * ks + khalf) * U_LD + l31;
                    d[0] = rr[0]; d[2] = rr[1]; d[1] = rr[33]; d[3] = rr[34];
                };
                auto transform = [&](float (&v)[4], const float (&d)[4]) {
                    v[0] = d[0] - d[2];
                    v[1] = d[1] + d[2];
                    v[2] = d[2] - d[1];
                    v[3] = d[1] - d[3];
                };
                load_d(Dn[0], 0);
                load_d(Dn[1], 1);
                transform(Vb[0], Dn[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
                for (int k0 = 0; k0 < NKS; k0 += WR) {
#pragma unroll
                    for (int s = 0; s < WR; ++s) {
                        const int ks = k0 + s;
#if !(P4_ABL & 1)
                        load_aw(Aw[(s + WR - 1) % WR], ps * NKS + ks + WR - 1);
#endif
#if !(P4_ABL & 2)
                        transform(Vb[(s + 1) & 1], Dn[(s + 1) & 1]);      // k-step ks + 1 (read during ks - 1)
                        load_d(Dn[s & 1], ks + 2);                         // (reads past the last k-step land in the z buffer: discarded)
#endif
#pragma unroll
                        for (int i = 0; i < CMT; ++i)
#pragma unroll
                            for (int tr = 0; tr < 4; ++tr)
                                accw[i][tr] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aw[s][i][tr], Vb[s & 1][tr], accw[i][tr], 0, 0, 0);
                        // the step's 2 weight loads, 2 LDS reads and 4 transform instructions go BETWEEN its 8 MFMAs (the wave is alone on its
                        // SIMD: whatever it issues in one block in front of the MFMAs is time the matrix pipe idles)
#if P4_SCHED
#pragma unroll
                        for (int q = 0; q < CMT; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (ps == NPASS - 1) stamp(l, 2);
            mfma_drain();
            {   // gate: z goes to its own buffer; nobody reads z before barrier (3).  Output transform: y(2p) = (m0 + m1) + m2,
                // y(2p+1) = (m1 - m2) - m3; lane p writes the frame pair as one 8-byte store
                const float* b3 = a.b3[l];
                const int ln = opaque(lane);
#pragma unroll
                for (int i = 0; i < CMT; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int mt = (w * NPASS + ps) * CMT + i;          // packed m-tile: 16 gate rows | 16 filter rows of channels 16 mt ..
                    float bg[8], bf[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int mg = mt * 32 + acc_row(r, ln);
                        bg[r] = ldg(b3, (unsigned)mg);
                        bf[r] = ldg(b3, (unsigned)(mg + 16));
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float m0 = acc_rd(accw[i][0][r]), m1 = acc_rd(accw[i][1][r]), m2 = acc_rd(accw[i][2][r]), m3 = acc_rd(accw[i][3][r]);
                        const float n0 = acc_rd(accw[i][0][r + 8]), n1 = acc_rd(accw[i][1][r + 8]), n2 = acc_rd(accw[i][2][r + 8]), n3 = acc_rd(accw[i][3][r + 8]);
                        const float ge = (m0 + m1) + m2;
                        const float go = (m1 - m2) - m3;
                        const float fe = (n0 + n1) + n2;
                        const float fo = (n1 - n2) - n3;
                        float2 zz;
                        zz.x = cmtts_gate(ge + bg[r], fe + bf[r]);
                        zz.y = cmtts_gate(go + bg[r], fo + bf[r]);
                        *reinterpret_cast<float2*>(z_lds + (mt * 16 + acc_row(r, ln)) * U_LD + 2 * (ln & 31)) = zz;
                    }
                }
            }
        }
        // ---- output projection: wave w owns residual-half tiles 2w, 2w+1 (rows mrow0 ..) and skip-half tiles 8 + 2w, 8 + 2w + 1; its first
        // ring groups are requested behind the gate
        f32x4 A[RP][2 * TPW];
        const float* Wof = a.Wof[l];
        auto load_ao = [&](f32x4 (&dst)[2 * TPW], int group) {
#pragma unroll
            for (int i = 0; i < 2 * TPW; ++i) {
                const int mt = (i < TPW ? 0 : C / 32) + w * TPW + (i % TPW);
                dst[i] = *reinterpret_cast<const f32x4*>(Wof + ((long)group * (2 * C / 32) + mt) * 256 + lane * 4);
            }
        };
#pragma unroll
        for (int s = 0; s < RP - 1; ++s) load_ao(A[s], min(s, NGC - 1));      // (in front of the gate these 48 registers push state into scratch)
        stamp(l, 3);
        __syncthreads();   // (3) z complete, u of this layer dead
        stamp(l, 4);

        // ---- the next layer's conditioner projection for this wave's elements (accumulator layout) does not depend on anything this layer
        // computes: requested HERE, in front of the projection loop, it has landed long before the publish phase needs it (in that phase
        // its latency — four batches of L2 gathers — was 8 k cycles of a 20 k phase with nothing to overlap it)
        f32x16 cpn_v[TPW][NT];
        if (more) {
            const float* cpn = cp_b + (long)(l + 1) * C * T;
            const int ln = opaque(lane), c31 = ln & 31;
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (FACT) {
                        const int row0 = (l + 1) * C + mrow0 + 32 * i;
#pragma unroll
                        for (int r = 0; r < 16; ++r) cpn_v[i][j][r] = cp_fact(row0 + acc_row(r, ln), fph[j], fix[j]);
                    } else {
                        const int t_c = min(t0 + j * 32 + c31, T - 1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) cpn_v[i][j][r] = ldg(cpn, (unsigned)((mrow0 + 32 * i + acc_row(r, ln)) * T + t_c));
                    }
                }
        }
        // =========================================================== phase C: output projection
        f32x16 acc[2 * TPW][NT];
#pragma unroll
        for (int i = 0; i < 2 * TPW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        {
            float Bv[2][4][NT];
            auto load_b = [&](float (&dst)[4][NT], int krow0) {
                const float* bs = z_lds + (krow0 + khalf) * U_LD + l31;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < NT; ++j) dst[kk][j] = bs[2 * kk * U_LD + j * 32];
            };
            load_b(Bv[0], 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int it = 0; it < NGC; it += RP) {
#pragma unroll
                for (int s = 0; s < RP; ++s) {
                    load_ao(A[(s + RP - 1) % RP], min(it + s + RP - 1, NGC - 1));
                    load_b(Bv[(s + 1) & 1], min(it + s + 1, NGC - 1) * 8);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < 2 * TPW; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][i][kk], Bv[s & 1][kk][j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        stamp(l, 5);
        mfma_drain();
        // ---- epilogue in registers: x' = (o[:C] + (x + d)) / sqrt(2); skip (+)= o[C:]
        {
            const float* bo = a.bo[l];
            const float* dl = dv_b + (long)l * C;
            const int ln = opaque(lane);
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // half a tile's rows at a time (24 bias / projection values in flight next to the 128 registers of state)
                    __builtin_amdgcn_sched_barrier(0);
                    float box[8], bos[8], ddr[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int m = mrow0 + 32 * i + acc_row(8 * h + q, ln);
                        box[q] = ldg(bo, (unsigned)m);
                        bos[q] = ldg(bo, (unsigned)(C + m));
                        ddr[q] = ldg(dl, (unsigned)m);
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int r = 8 * h + q;
                            const float o = acc_rd(acc[i][j][r]) + box[q];
                            xs[i][j][r] = (o + (xs[i][j][r] + ddr[q])) * CMTTS_RSQRT2;
                            const float os = acc_rd(acc[TPW + i][j][r]) + bos[q];
                            sk[i][j][r] = l > 0 ? os + sk[i][j][r] : os;
                        }
                }
        }
        if (!more) break;
        stamp(l, 6);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand the edge columns of x' to the neighbouring tiles first (their latency is what the neighbours wait for)
        const float* dpn = dp_b + (long)(l + 1) * C;
        const unsigned tag = (unsigned)l + 1;
        unsigned long long* hbase = halo_g + ((((long)(l & 1) * B_g + b) * tiles_g) * 2) * C;    // [parity][b][tile][side][C]
        {
            const int ln = opaque(lane), c31 = ln & 31;
            // column 0 of this tile -> slot (tile, side 0); column FN-1 -> slot (tile, side 1); through the wave's LDS scratch (behind the
            // u / z buffers and the 8-wave kernel's index table) so that every lane stores one granule per row tile (denoiser_persist.hip)
            float* edge = smem + 2 * C * U_LD + (FN + 2) * 2 + w * (64 * TPW);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (c31 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) edge[64 * i + acc_row(r, ln)] = xs[i][0][r];
                }
                if (c31 == 31) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) edge[64 * i + 32 + acc_row(r, ln)] = xs[i][NT - 1][r];
                }
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                store_granule(hbase + ((long)tile * 2 + (ln >> 5)) * C + mrow0 + 32 * i + (ln & 31), tag, edge[64 * i + ln]);
        }
        // ---- halo columns of the next layer's u: every wave fetches the two halo entries of ITS OWN 64 rows (lanes 0-31: left halo frame
        // t0 - 1, lanes 32-63: right halo frame t0 + FN; two rows per lane), requested here — before the wave's own u rows — and checked after
        const int hside = opaque(lane) >> 5, hm0 = mrow0 + (opaque(lane) & 31);
        const int hth = hside ? t0 + FN : t0 - 1;
        const bool hinside = hth >= 0 && hth < Tc;
        // neighbour's slot: its right edge (side 1) feeds our left halo, its left edge (side 0) our right halo
        const unsigned long long* hg = hbase + ((long)(hinside ? (hside ? tile + 1 : tile - 1) : tile) * 2 + (hside ? 0 : 1)) * C + hm0;
        float hcp[TPW];
        {
            const int thc = min(max(hth, 0), T - 1);
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                if (FACT) hcp[q] = cp_fact((l + 1) * C + hm0 + 32 * q, hph, hix);
                else hcp[q] = (cp_b + (long)(l + 1) * C * T)[(unsigned)((hm0 + 32 * q) * T + thc)];
            }
        }
        unsigned long long hv[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) hv[q] = __hip_atomic_load((gu64*)(hg + 32 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- next layer's u rows of this wave: cp (requested before the projection) + (x' + dp)
        {
            const int ln = opaque(lane), c31 = ln & 31;
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                float dpr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dpr[r] = ldg(dpn, (unsigned)(mrow0 + 32 * i + acc_row(r, ln)));
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int t = t0 + j * 32 + c31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mrow0 + 32 * i + acc_row(r, ln);
                        const float uv = cpn_v[i][j][r] + (xs[i][j][r] + dpr[r]);
                        u_lds[m * U_LD + uidx(j * 32 + c31)] = t < Tc ? uv : 0.f;
                    }
                }
            }
        }
        {   // the halo entries: wait for the neighbours' tags (lanes without a neighbour frame never wait)
            if (!gave_up) {
                unsigned spins = 0;
                while (!__all(!hinside || ((unsigned)(hv[0] >> 32) == tag && (unsigned)(hv[TPW - 1] >> 32) == tag))) {
                    if (++spins > SPIN_LIMIT) {      // wave-uniform: a neighbour never arrived
                        if (lane == 0 && a.tmo) *(volatile unsigned*)a.tmo = 1u;
                        gave_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
#pragma unroll
                    for (int q = 0; q < TPW; ++q) hv[q] = __hip_atomic_load((gu64*)(hg + 32 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // after a timeout the halo column is poisoned: the utterance's mel comes out NaN instead of plausible-but-wrong
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int hm = hm0 + 32 * q;
                const float xh = gave_up ? __builtin_nanf("") : (hinside ? __uint_as_float((unsigned)hv[q]) : 0.f);
                const float uh = hcp[q] + (xh + dpn[hm]);
                u_lds[hm * U_LD + uidx(hside ? FN : -1)] = hinside ? uh : 0.f;
            }
        }
        stamp(l, 7);
    }

    if (a.tail) {   // skip head + post-scaling in-kernel (persist_tail.h); the u buffer is free since barrier (3), z after barrier (A) inside
        persist_tail::run<TPW>(a, smem, smem + C * U_LD, &sk[0][0], w, lane, b, t0, T, RAGGED ? a.grp[gi].xold : a.xold,
                               RAGGED ? a.grp[gi].noise : a.noise, RAGGED ? a.grp[gi].out : a.out, Tc);
    } else {   // ---- the skip sum leaves the chip once
        float* skip = (RAGGED ? a.grp[gi].skip : a.skip) + (long)b * C * T;
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = t0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t < Tc) skip[(unsigned)((mrow0 + 32 * i + acc_row(r, lane)) * T + t)] = sk[i][j][r];
            }
    }
}

}  // namespace

// The instance a launcher of denoiser_persist.hip picks for PersistArgs.wino == 2 (256 threads per workgroup, the same dynamic LDS).
extern "C" const void* cmtts_persist4_kernel(int dbg, int ragged, int fact) {
    if (ragged) return fact ? reinterpret_cast<const void*>(denoiser_persist4_kernel<false, true, true>)
                            : reinterpret_cast<const void*>(denoiser_persist4_kernel<false, true, false>);
    if (dbg) return fact ? reinterpret_cast<const void*>(denoiser_persist4_kernel<true, false, true>)
                         : reinterpret_cast<const void*>(denoiser_persist4_kernel<true, false, false>);
    return fact ? reinterpret_cast<const void*>(denoiser_persist4_kernel<false, false, true>)
                : reinterpret_cast<const void*>(denoiser_persist4_kernel<false, false, false>);
}
extern "C" int cmtts_persist4_threads(void) { return 64 * NW; }
