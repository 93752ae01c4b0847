// C ABI of the MI355X-native CM-TTS inference hot path (include/cmtts_hip.h): weight import
// (re-pack + upload), workspace carving and the host-side launch sequences.  No allocation and no
// host synchronisation after cmtts_finalize(); everything is enqueued on the caller's stream.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/cmtts_hip.h"
#include "internal_hooks.h"
#include "conv_args.h"
#include "kernels.h"
#include "resblock_args.h"
#include "persist_args.h"
#include "cond_gemm.h"
#include "resblock_pair.h"
#include "inproj.h"
#include "attention.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) return fail(CMTTS_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define CHK(x)                 \
    do {                       \
        int r_ = (x);          \
        if (r_ != 0) return r_; \
    } while (0)

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t dim(int i) const { return i < (int)shape.size() ? shape[i] : 1; }
};

struct PackedConv {
    float* w = nullptr;     // device, [phase][tap][cin][ld]
    float* bias = nullptr;  // device, [cout] (packed row order)
    int cout = 0, cin = 0, taps = 0, ld = 0;
    long tap_stride = 0, phase_stride = 0;
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Allocs {
    std::vector<void*> ptrs;
    int upload(const std::vector<float>& h, float** out) {
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, h.size() * sizeof(float) + 256));
        HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        ptrs.push_back(p);
        *out = (float*)p;
        return 0;
    }
    int upload_bytes(const void* h, size_t nbytes, void** out) {
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, nbytes + 256));
        HIPCHK(hipMemcpy(p, h, nbytes, hipMemcpyHostToDevice));
        ptrs.push_back(p);
        *out = p;
        return 0;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

// W [Cout][Cin][K] -> k-major [K][Cin][ld] with ld = round_up(Cout, 4); perm[p] = original row of packed row p
int pack_conv(Allocs& al, const HostTensor& W, const HostTensor* bias, const std::vector<int>* perm, PackedConv* out,
              std::vector<float>* host_copy = nullptr) {
    const int Cout = (int)W.dim(0), Cin = (int)W.dim(1), K = (int)W.dim(2);
    const int ld = round_up(Cout, 4);
    std::vector<float> p((size_t)K * Cin * ld, 0.f);
    for (int k = 0; k < K; ++k)
        for (int ci = 0; ci < Cin; ++ci)
            for (int r = 0; r < Cout; ++r) {
                const int co = perm ? (*perm)[r] : r;
                p[((size_t)k * Cin + ci) * ld + r] = W.data[((size_t)co * Cin + ci) * K + k];
            }
    CHK(al.upload(p, &out->w));
    if (host_copy) *host_copy = p;
    out->bias = nullptr;
    if (bias) {
        std::vector<float> b(Cout);
        for (int r = 0; r < Cout; ++r) b[r] = bias->data[perm ? (*perm)[r] : r];
        CHK(al.upload(b, &out->bias));
    }
    out->cout = Cout; out->cin = Cin; out->taps = K; out->ld = ld;
    out->tap_stride = (long)Cin * ld;
    out->phase_stride = 0;
    return 0;
}

// ConvTranspose1d W [Cin][Cout][K], stride s -> polyphase [s][K/s][Cin][ld]: phase r uses taps k = r + s*q
int pack_conv_transpose(Allocs& al, const HostTensor& W, const HostTensor& bias, int s, PackedConv* out,
                        std::vector<float>* two_tap = nullptr) {
    const int Cin = (int)W.dim(0), Cout = (int)W.dim(1), K = (int)W.dim(2);
    const int Q = K / s, ld = round_up(Cout, 4);
    if (two_tap && Q == 2) {   // the same weights as ONE two-tap conv with s * Cout stacked rows (row = phase * Cout + co): [2][Cin][s * Cout]
        two_tap->assign((size_t)2 * Cin * s * Cout, 0.f);
        for (int q = 0; q < 2; ++q)
            for (int ci = 0; ci < Cin; ++ci)
                for (int r = 0; r < s; ++r)
                    for (int co = 0; co < Cout; ++co)
                        (*two_tap)[((size_t)q * Cin + ci) * s * Cout + (size_t)r * Cout + co] = W.data[((size_t)ci * Cout + co) * K + (r + s * q)];
    }
    std::vector<float> p((size_t)s * Q * Cin * ld, 0.f);
    for (int r = 0; r < s; ++r)
        for (int q = 0; q < Q; ++q)
            for (int ci = 0; ci < Cin; ++ci)
                for (int co = 0; co < Cout; ++co)
                    p[(((size_t)r * Q + q) * Cin + ci) * ld + co] = W.data[((size_t)ci * Cout + co) * K + (r + s * q)];
    CHK(al.upload(p, &out->w));
    CHK(al.upload(bias.data, &out->bias));
    out->cout = Cout; out->cin = Cin; out->taps = Q; out->ld = ld;
    out->tap_stride = (long)Cin * ld;
    out->phase_stride = (long)Q * Cin * ld;
    return 0;
}

// k-major packed weights [taps][K][M] -> MFMA A-fragment order [taps][K/8][M/32][64 lanes][4]:
// element (lane, j) of k-group g, m-tile mt is P[tap][8g + 2j + (lane >> 5)][32 mt + (lane & 31)], i.e. the A
// operand of v_mfma_f32_32x32x2_f32 for k-step j of that group (lane l supplies A[m = l & 31][k = l >> 5]).
std::vector<float> to_fragment_order(const std::vector<float>& p, int taps, int K, int M) {
    std::vector<float> f((size_t)taps * K * M);
    const int G = K / 8, MTn = M / 32;
    for (int tap = 0; tap < taps; ++tap)
        for (int g = 0; g < G; ++g)
            for (int mt = 0; mt < MTn; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j)
                        f[((((size_t)tap * G + g) * MTn + mt) * 64 + lane) * 4 + j] =
                            p[((size_t)tap * K + 8 * g + 2 * j + (lane >> 5)) * M + 32 * mt + (lane & 31)];
    return f;
}

// Winograd F(2,3) form of a k = 3 conv for the persistent denoiser's WINO instances (denoiser_persist.hip): k-major packed weights
// [3][K][M] -> transformed weights G0 = g0, G1 = (g0 + g1 + g2) / 2, G2 = (g0 - g1 + g2) / 2, G3 = g2 (formed in double, rounded once) as MFMA
// A fragments [K/4 half-groups][M/32][2][64 lanes][4]: element q of fragment (hg, mt, ps) at lane l is transform 2 ps + (q >> 1) of
// input channel 4 hg + 2 (q & 1) + (l >> 5), output row 32 mt + (l & 31).
constexpr int WINO_PAD_HG = 4;        // half-groups of zero padding behind a layer's array: the kernel's weight ring runs a few stages past the end
std::vector<float> to_wino_fragments(const std::vector<float>& p, int K, int M) {
    std::vector<float> f((size_t)4 * K * M + (size_t)WINO_PAD_HG * (M / 32) * 2 * 64 * 4, 0.0f);
    const int MTn = M / 32;
    for (int hg = 0; hg < K / 4; ++hg)
        for (int mt = 0; mt < MTn; ++mt)
            for (int ps = 0; ps < 2; ++ps)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 4; ++q) {
                        const int tr = 2 * ps + (q >> 1), k = 4 * hg + 2 * (q & 1) + (lane >> 5), mrow = 32 * mt + (lane & 31);
                        const double g0 = p[((size_t)0 * K + k) * M + mrow], g1 = p[((size_t)1 * K + k) * M + mrow], g2 = p[((size_t)2 * K + k) * M + mrow];
                        const double v = tr == 0 ? g0 : tr == 1 ? 0.5 * (g0 + g1 + g2) : tr == 2 ? 0.5 * (g0 - g1 + g2) : g2;
                        f[((((size_t)hg * MTn + mt) * 2 + ps) * 64 + lane) * 4 + q] = (float)v;
                    }
    return f;
}

// Winograd F(4,3) form of the same conv for the persistent denoiser's WINO == 2 instances (points 0, +-1, +-2, inf): transformed weights
// U0 = g0 / 4, U1 = -(g0 + g1 + g2) / 6, U2 = -(g0 - g1 + g2) / 6, U3 = g0 / 24 + g1 / 12 + g2 / 6, U4 = g0 / 24 - g1 / 12 + g2 / 6, U5 = g2
// (formed in double, rounded once) as A fragments of v_mfma_f32_16x16x4_f32 in the kernel's iteration order
// [K/4 k-steps][M/64 waves][6 transforms][64 lanes][4]: element e at lane l is input channel 4 ks + (l >> 4), output row 64 w + 16 e + (l & 15).
constexpr int WINO43_PAD_KS = 4;      // k-steps of zero padding behind a layer's array (the weight ring runs a few stages past the end)
std::vector<float> to_wino43_fragments(const std::vector<float>& p, int K, int M) {
    const int NWV = M / 64;
    std::vector<float> f((size_t)(K / 4 + WINO43_PAD_KS) * NWV * 6 * 64 * 4, 0.0f);
    for (int ks = 0; ks < K / 4; ++ks)
        for (int w = 0; w < NWV; ++w)
            for (int tr = 0; tr < 6; ++tr)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * ks + (lane >> 4), mrow = 64 * w + 16 * e + (lane & 15);
                        const double g0 = p[((size_t)0 * K + k) * M + mrow], g1 = p[((size_t)1 * K + k) * M + mrow], g2 = p[((size_t)2 * K + k) * M + mrow];
                        double v;
                        switch (tr) {
                            case 0: v = g0 / 4.0; break;
                            case 1: v = -(g0 + g1 + g2) / 6.0; break;
                            case 2: v = -(g0 - g1 + g2) / 6.0; break;
                            case 3: v = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                            case 4: v = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                            default: v = g2; break;
                        }
                        f[((((size_t)ks * NWV + w) * 6 + tr) * 64 + lane) * 4 + e] = (float)v;
                    }
    return f;
}

// Winograd form of a k-tap conv for conv_xlw_kernel (resblock_pair.h: WinoTab<KT>): the transformed weights of every table entry (formed in
// double, rounded once) as A fragments in the kernel's iteration order [K/16 chunks][entries][2 halves][M/32][64 lanes][4].
template <int KT>
std::vector<float> to_wino_iter_fragments_k(const std::vector<float>& p, int K, int M) {
    using TAB = WinoTab<KT>;
    std::vector<float> f((size_t)TAB::N * K * M);
    const int MTn = M / 32;
    size_t o = 0;
    for (int c = 0; c < K / 16; ++c)
        for (int e = 0; e < TAB::N; ++e)
            for (int h = 0; h < 2; ++h)
                for (int mt = 0; mt < MTn; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const int k = 16 * c + 8 * h + 2 * j + (lane >> 5), mrow = 32 * mt + (lane & 31), tau = TAB::e[e].tau;
                            auto g = [&](int tap) -> double { return tap < KT ? (double)p[((size_t)tap * K + k) * M + mrow] : 0.0; };
                            double v = 0.0;
                            switch (TAB::e[e].wkind) {
                                case 0: v = g(tau); break;
                                case 1: v = 0.5 * (g(tau) + g(tau + 1) + g(tau + 2)); break;
                                case 2: v = 0.5 * (g(tau) - g(tau + 1) + g(tau + 2)); break;
                                case 3: v = g(tau + 2); break;
                                case 4: v = -g(tau); break;
                                case 5: v = g(tau) + g(tau + 1); break;
                                case 6: v = g(tau + 1); break;
                            }
                            f[o++] = (float)v;
                        }
    return f;
}
std::vector<float> to_wino_iter_fragments(const std::vector<float>& p, int taps, int K, int M) {
    if (taps == 3) return to_wino_iter_fragments_k<3>(p, K, M);
    if (taps == 7) return to_wino_iter_fragments_k<7>(p, K, M);
    if (taps == 11) return to_wino_iter_fragments_k<11>(p, K, M);
    return {};
}

// F(4,3) form of the FFT blocks' k = 9 FFN conv for conv_xres_kernel<.., WQ = true> (conv_xres.hip): three groups of three taps, six transforms each (U0 .. U5 as in
// to_wino43_fragments), as A fragments of v_mfma_f32_16x16x4_f32 in the kernel's iteration order [K/4 k-steps][M/32 m-tiles][9][64 lanes][4]: vector j of a (k-step,
// m-tile) holds transforms 2 j and 2 j + 1 (of tap group j / 3) for the m-tile's two 16-row halves — element (pt & 1) * 2 + i at lane l = input channel 4 ks + (l >> 4),
// output row 32 mt + 16 i + (l & 15).
std::vector<float> to_wino43_xres_fragments(const std::vector<float>& p, int taps, int K, int M) {
    if (taps != 9 || K % 4 || M % 32) return {};
    const int MTn = M / 32;
    std::vector<float> f((size_t)(K / 4) * MTn * 9 * 256);
    for (int ks = 0; ks < K / 4; ++ks)
        for (int mt = 0; mt < MTn; ++mt)
            for (int pt = 0; pt < 18; ++pt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 2; ++i) {
                        const int k = 4 * ks + (lane >> 4), mrow = 32 * mt + 16 * i + (lane & 15), tau = 3 * (pt / 6);
                        const double g0 = p[((size_t)tau * K + k) * M + mrow], g1 = p[((size_t)(tau + 1) * K + k) * M + mrow], g2 = p[((size_t)(tau + 2) * K + k) * M + mrow];
                        double v;
                        switch (pt % 6) {
                            case 0: v = g0 / 4.0; break;
                            case 1: v = -(g0 + g1 + g2) / 6.0; break;
                            case 2: v = -(g0 - g1 + g2) / 6.0; break;
                            case 3: v = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                            case 4: v = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                            default: v = g2; break;
                        }
                        f[((((size_t)ks * MTn + mt) * 9 + pt / 2) * 64 + lane) * 4 + (pt & 1) * 2 + i] = (float)v;
                    }
    return f;
}

// The same conv as three F(2,3) tap groups over output PAIRS (conv_xres.hip, WQ == 2 instances; round 6): per (k-step of four channels, 32-row m-tile, tap group) the four
// transformed weights U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (formed in double, rounded once) for the m-tile's two 16-row halves, as two 16-byte
// vectors per lane: [K/4][M/32][3][2][64 lanes][4], element (tr & 1) * 2 + i of vector tr / 2 at lane l = transform tr, input channel 4 ks + (l >> 4), output row 32 mt + 16 i + (l & 15).
std::vector<float> to_wino23_xres_fragments(const std::vector<float>& p, int taps, int K, int M) {
    if (taps != 9 || K % 4 || M % 32) return {};
    const int MTn = M / 32;
    std::vector<float> f((size_t)(K / 4) * MTn * 6 * 256);
    for (int ks = 0; ks < K / 4; ++ks)
        for (int mt = 0; mt < MTn; ++mt)
            for (int g = 0; g < 3; ++g)
                for (int tr = 0; tr < 4; ++tr)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 2; ++i) {
                            const int k = 4 * ks + (lane >> 4), mrow = 32 * mt + 16 * i + (lane & 15), tau = 3 * g;
                            const double g0 = p[((size_t)tau * K + k) * M + mrow], g1 = p[((size_t)(tau + 1) * K + k) * M + mrow], g2 = p[((size_t)(tau + 2) * K + k) * M + mrow];
                            const double v = tr == 0 ? g0 : tr == 1 ? 0.5 * (g0 + g1 + g2) : tr == 2 ? 0.5 * (g0 - g1 + g2) : g2;
                            f[(((((size_t)ks * MTn + mt) * 3 + g) * 2 + tr / 2) * 64 + lane) * 4 + (tr & 1) * 2 + i] = (float)v;
                        }
    return f;
}

// F(4,3) form of a k-tap, dilation-1 conv for conv_xlq_kernel (conv_xlq.hip: QTab<KT>): per k-step of four input channels the transformed weights of
// every group of three taps (U0 = g0/4, U1 = -(g0+g1+g2)/6, U2 = -(g0-g1+g2)/6, U3 = g0/24 + g1/12 + g2/6, U4 = g0/24 - g1/12 + g2/6, U5 = g2; a tap beyond the
// kernel is zero) and, for k = 7, of the seventh tap on its own (g, g/2, g/2, g) — formed in double, rounded once — as A fragments of v_mfma_f32_16x16x4_f32 in
// the kernel's iteration order [K/4 k-steps][M/64 waves][points][64 lanes][4]: element i at lane l = input channel 4 ks + (l >> 4), output row 64 w + 16 i + (l & 15).
std::vector<float> to_wino43_iter_fragments(const std::vector<float>& p, int taps, int K, int M) {
    if ((taps != 3 && taps != 5 && taps != 7 && taps != 11) || K % 4 || M % 64) return {};
    const int ngrp = taps == 3 ? 1 : taps <= 7 ? 2 : 4, npt = ngrp * 6 + (taps == 7 ? 4 : 0), NWV = M / 64;
    std::vector<float> f((size_t)(K / 4) * NWV * npt * 256);
    for (int ks = 0; ks < K / 4; ++ks)
        for (int w = 0; w < NWV; ++w)
            for (int pt = 0; pt < npt; ++pt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                        const int k = 4 * ks + (lane >> 4), mrow = 64 * w + 16 * i + (lane & 15);
                        auto g = [&](int tap) -> double { return tap < taps ? (double)p[((size_t)tap * K + k) * M + mrow] : 0.0; };
                        double v;
                        if (pt < ngrp * 6) {
                            const int tau = 3 * (pt / 6);
                            const double g0 = g(tau), g1 = g(tau + 1), g2 = g(tau + 2);
                            switch (pt % 6) {
                                case 0: v = g0 / 4.0; break;
                                case 1: v = -(g0 + g1 + g2) / 6.0; break;
                                case 2: v = -(g0 - g1 + g2) / 6.0; break;
                                case 3: v = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                                case 4: v = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                                default: v = g2; break;
                            }
                        } else {
                            const int q = pt - ngrp * 6;
                            v = (q == 1 || q == 2) ? 0.5 * g(6) : g(6);
                        }
                        f[((((size_t)ks * NWV + w) * npt + pt) * 64 + lane) * 4 + i] = (float)v;
                    }
    return f;
}

// The same fragments in the ITERATION order of the fused ResBlock pair kernels (resblock_pair.hip): the K loop walks
// (16-channel chunk, tap, 8-channel half), so [K/16][taps][2][M/32][64 lanes][4] makes the weight stream one linear walk.
std::vector<float> to_fragment_iter_order(const std::vector<float>& p, int taps, int K, int M) {
    std::vector<float> f((size_t)taps * K * M);
    const int MTn = M / 32;
    size_t o = 0;
    for (int c = 0; c < K / 16; ++c)
        for (int tap = 0; tap < taps; ++tap)
            for (int h = 0; h < 2; ++h)
                for (int mt = 0; mt < MTn; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            f[o++] = p[((size_t)tap * K + 16 * c + 8 * h + 2 * j + (lane >> 5)) * M + 32 * mt + (lane & 31)];
    return f;
}

inline unsigned short host_cvt16(float f, int mode) {   // mode 1 = bf16 (round to nearest even), 2 = fp16
    if (mode == 1) {
        unsigned u;
        memcpy(&u, &f, 4);
        return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    const _Float16 h = (_Float16)f;
    unsigned short r;
    memcpy(&r, &h, 2);
    return r;
}

// k-major packed weights [taps][K][M] -> 16-bit MFMA A-fragment order for v_mfma_f32_32x32x16_{bf16,f16}:
// [taps][K/16][M/32][64 lanes][8]: element (lane, j) = P[tap][16g + 8 (lane >> 5) + j][32 mt + (lane & 31)].
std::vector<unsigned short> to_fragment16(const std::vector<float>& p, int taps, int K, int M, int mode) {
    std::vector<unsigned short> f((size_t)taps * K * M);
    const int G = K / 16, MTn = M / 32;
    for (int tap = 0; tap < taps; ++tap)
        for (int g = 0; g < G; ++g)
            for (int mt = 0; mt < MTn; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        f[((((size_t)tap * G + g) * MTn + mt) * 64 + lane) * 8 + j] =
                            host_cvt16(p[((size_t)tap * K + 16 * g + 8 * (lane >> 5) + j) * M + 32 * mt + (lane & 31)], mode);
    return f;
}

// The same fragments in the ITERATION order of conv_mfma16.hip's deep-ring variant: its K loop walks (32-channel chunk, tap,
// k-group of the chunk), so [K/32][taps][2][M/32][64 lanes][8] makes the weight stream one linear walk (K % 32 == 0).
std::vector<unsigned short> to_fragment16_iter(const std::vector<float>& p, int taps, int K, int M, int mode) {
    std::vector<unsigned short> f((size_t)taps * K * M);
    const int MTn = M / 32;
    size_t o = 0;
    for (int c = 0; c < K / 32; ++c)
        for (int tap = 0; tap < taps; ++tap)
            for (int h = 0; h < 2; ++h)
                for (int mt = 0; mt < MTn; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j)
                            f[o++] = host_cvt16(p[((size_t)tap * K + 32 * c + 16 * h + 8 * (lane >> 5) + j) * M + 32 * mt + (lane & 31)], mode);
    return f;
}

// fp16x3 operands: every weight as hi = fp16(w) and lo = fp16(w - hi); the lo fragment set follows the hi set
std::vector<unsigned short> to_fragment16_split(const std::vector<float>& p, int taps, int K, int M) {
    std::vector<float> hi(p.size()), lo(p.size());
    for (size_t i = 0; i < p.size(); ++i) {
        const _Float16 h = (_Float16)p[i];
        hi[i] = (float)h;
        lo[i] = p[i] - (float)h;
    }
    std::vector<unsigned short> f = to_fragment16(hi, taps, K, M, 2);
    const std::vector<unsigned short> fl = to_fragment16(lo, taps, K, M, 2);
    f.insert(f.end(), fl.begin(), fl.end());
    return f;
}

// nn.Linear weight [N][K] -> transposed [K][N] (dense_small operand / X operand of a GEMM)
std::vector<float> transpose2d(const float* w, int N, int K) {
    std::vector<float> t((size_t)K * N);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) t[(size_t)k * N + n] = w[(size_t)n * K + k];
    return t;
}

std::vector<float> omega_table(int C) {
    // SinusoidalPositionalEmbedding.get_embedding (model/blocks.py:50-54): exp(arange(half) * -ln(1e4)/(half-1)) in fp32
    const int half = C / 2;
    const float e = (float)(log(10000.0) / (half - 1));
    std::vector<float> w(half);
    for (int j = 0; j < half; ++j) w[j] = (float)exp((double)((float)j * -e));
    return w;
}

// Sinusoid table rows 0..rows-1 (row 0 = padding = zeros): fp32 argument p*w_j, sine/cosine in f64 rounded to
// fp32 — the same recipe the kernels use on the fly beyond the table.
constexpr int PE_ROWS = 4096;
std::vector<float> pe_table(int C, int rows) {
    const std::vector<float> w = omega_table(C);
    const int half = C / 2;
    std::vector<float> t((size_t)rows * C, 0.f);
    for (int p = 1; p < rows; ++p)
        for (int c = 0; c < C; ++c) {
            const float arg = (float)p * w[c < half ? c : c - half];
            t[(size_t)p * C + c] = c < half ? (float)sin((double)arg) : (float)cos((double)arg);
        }
    return t;
}

ConvArgs conv_args(const PackedConv& w, const float* X, int Tin, int ldx, long x_bs, float* Y, int ldy, long y_bs, int N) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.A = w.w; a.X = X;
    a.M = w.cout; a.N = N; a.K = w.cin;
    a.taps = w.taps; a.dil = 1; a.pad = (w.taps - 1) / 2;
    a.Tin = Tin; a.a_ld = w.ld; a.a_cols = w.ld; a.a_tap_stride = w.tap_stride; a.ldx = ldx;
    a.zdiv = 1; a.x_zs0 = x_bs;
    a.pre_div = 1.f; a.pre_slope = 1.f;
    a.split = INT_MAX;
    for (int i = 0; i < 2; ++i) {
        ConvOut& o = a.out[i];
        o.Y = Y; o.y_zs0 = y_bs; o.ldy = ldy; o.Tout = N; o.ostride = 1;
        o.bias = w.bias; o.alpha = 1.f; o.act = ACT_NONE; o.div = 1.f;
    }
    return a;
}

int launch(const ConvArgs& a, int epi, int nbatch, hipStream_t s) {
    const int r = cmtts_launch_conv(&a, epi, nbatch, (void*)s);
    if (r != 0) return fail(r, "conv launch failed (unsupported shape or HIP launch error)");
    return 0;
}

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// Optional HIP-event instrumentation of the dominant kernel (the gated k=3 conv of the denoiser
// residual block), recorded on the launch stream: bench.py's live roofline figure.
struct Profile {
    bool on = false;
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;
    int stride = 1;               // every stride-th launch is bracketed (event records cost launch-queue time)
    long seen = 0;
} g_prof;

bool g_fused_resblock = true;
int g_persist_tail = 1;      // skip head + post-scaling inside the persistent denoiser launch (false: separate launches)
extern "C" int cmtts_launch_conv_xresq(const ConvArgs* a, const float* wq, int nbatch, void* stream, int form);      // conv_xres.hip
int g_ffn_wino = 1;             // FFT blocks, fp32: the k = 9 FFN conv of the fused launch as three Winograd tap groups (conv_xres.hip WQ instances; NOT bitwise the direct form: fp32
                                // rounding): 1 (default since round 6) = F(2,3) over output pairs, 2 = F(4,3) over output quads (round 5's default), 0 = direct.  Every fp32 FFN
                                // whose shape the X-resident kernel covers then takes it — whatever L and B — so that the text side's bits, and with them durations and lengths,
                                // still do not depend on the batch.
int g_ffn_fused = 1;            // FFT blocks: the FFN linear's K-segment partial products formed inside the k = 9 conv's launch (conv_xres.hip; same bits); 0 = its own launch
int g_inproj_fused = 1;         // denoiser input: c_in scaling + transpose + input projection + halo clearing in one launch (same bits); 0 = three launches
int g_step_cache = 1;           // cmtts_sample: reuse the timestep-only part of the step embedding across calls (same bits); 0 = recompute every call
int g_persist_wino = 3;         // fp32 persistent denoiser: the k = 3 conv in a Winograd form (NOT bitwise the direct form): 3 = F(4,3), the 8-wave WINO == 2 instances
                                // of denoiser_persist.hip (default since round 5: half of the conv's MFMAs), 1 = F(2,3), the 8-wave WINO == 1 instances (2/3), 0 = direct
                                // (2 was round 5's one-wave-per-SIMD F(2,3) stack — same bits as 1, measured 4-10 % slower, out of the build since round 6: tools/attic/ — and runs as 1)
int g_voc_wino = 1;             // fp32 HiFi-GAN, C >= 128: ResBlock convs in their Winograd form (conv_xlw_kernel; NOT bitwise the direct form): 0 never, 1 launches of >= 1024 column tiles, 2 always (tests)
int g_voc_wino64_k = 7;          // smallest kernel size of the C = 64 stage that takes the two-launch Winograd form (measurement switch voc_wino64_k)
int g_voc_wino43 = 1;           // fp32 HiFi-GAN: the convs of the Winograd path in the F(4,3) form (conv_xlq_kernel) instead of F(2,3) tap groups (measurement switch; 1 = dilation 1 and 3 everywhere + dilation 5 at C = 256 or k = 3 (default), 2 = only dilation 1, 3 = every dilation)
int g_voc_qpair = 1;            // fp32 HiFi-GAN, k = 3 pairs at C = 64 / 128 of the Winograd path: both convs F(4,3) in ONE launch, xt on the CU (conv_xlq_pair.hip; NOT bitwise the two conv_xlq launches: the quads of its conv1 start one frame earlier): 0 never, 1 launches of >= 1024 column tiles, 2 always (tests)
int g_voc_wino64 = 1;           // fp32 HiFi-GAN, C = 64, k >= 7: two Winograd launches per pair instead of the pair kernel (measurement switch)
int g_voc_pair3 = 1;            // fp16x3 HiFi-GAN, C <= 128: ResBlock pair as ONE X-resident launch (resblock_pair16x3.hip; same bits); 0 = two conv16 launches per pair
int g_voc_pairw = 1;            // 16-bit HiFi-GAN, C = 128: ResBlock pair as ONE launch with one in-place LDS image, two workgroups per CU (resblock_pairw16.hip; same bits); 0 = two conv_xl16 launches
int g_voc_pair128 = 1;          // 16-bit HiFi-GAN, C = 128: pair kernel (1) or two conv_xl16 launches (0); same bits
int g_voc_rb16 = 1;             // 16-bit HiFi-GAN, C <= 64: a whole ResBlock (three pairs) per launch (same bits); 0 = one launch per pair
int g_voc_upsT = 1;             // HiFi-GAN upsamplers: all phases of a ConvTranspose1d in one X-resident launch (same bits); 0 = generic kernel, one z per phase
int g_voc_xl16 = 1;             // 16-bit HiFi-GAN convs at C >= 128 on the X-resident conv_xl16 kernel (same bits); 0 = chunked conv_mfma16 kernel
int g_pred_xl = 1;              // frame-level 256 -> 256 predictor convs on the X-resident conv_xl kernel (bitwise equal); 0 = generic kernel
int g_pred_head = 1;            // predictors: last LayerNorm + linear head as one launch (ln_linear_kernel); 0 = layernorm_ct + chan_linear
int g_text_xres = 7;            // FFT blocks, bit mask: 1 = LayerNorm1 + in-projection in one X-resident launch, 2 = out-projection on that kernel (round 4: its 32-column instance, default on), 4 = LayerNorm2 as the prologue of the FFN conv; 0 = separate LayerNorm launches
int g_attn_fused = 1;           // FFT-block attention as QKV projection + ONE fused kernel (attention.hip; key-chunked with an online softmax above L = 192): 0 = three-launch path
int g_voc_xl = 1;               // HiFi-GAN ResBlock convs of the C >= 128 stages through the X-resident kernel (conv_xl): 0 never, 1 yes
int g_voc_pair = 1;             // HiFi-GAN ResBlock pairs of the C <= 64 stages as one launch (resblock_pair{,16}.hip): 0 never, 1 where it pays, 2 always
int g_qkv_nt = 0;              // internal switch "qkv_nt": conv_xres tile width of the in-projection (0 = launcher's rule, 1, 3) — measurements
int g_stats_mlp = 1;           // round 6: cwt_stats_layers as one launch (kernels.hip: stats_mlp_kernel; same bits); 0 = three dense_small launches
int g_cwt_in_phoneme = 1;      // round 4: the pitch predictor's input projection applied before the length regulator (same bits); 0 = over the frames
int g_xres_small = 1;          // round 4: conv_xres with 32-column tiles for text-side launches that cannot fill the chip (same bits); 0 = the generic kernel there
int g_ffn_xres = 1;            // encoder k=9 FFN conv through conv_xres.hip when the shape suits it (false: generic kernel)
int g_split_resblock = 1;       // fp32 residual block as two launches over 4x the CUs (resblock_split.hip): 0 never, 1 small batches, 2 always
int g_cond_gemm16 = 1;          // bf16 / fp16 / fp16x3 models: conditioner GEMM with 16-bit operands (cond_gemm16.hip); 0 = fp32 operands as until round 3 (different numerics)
int g_cond_gemm = 1;            // stacked conditioner GEMM through cond_gemm.hip: 0 never (generic kernel), 1 when it pays, 2 whenever supported
int g_persist = 1;              // residual layers in one persistent launch (denoiser_persist.hip): 0 never, 1 when it pays, 2 whenever supported
unsigned* g_tmo_host = nullptr;  // pinned, device-visible: 1 = a neighbour wait of the persistent kernel expired, 2 = a denoiser evaluation wrote a
                                 // non-finite mel value (persist_tail.h, mel_post_kernel: the sampler's post-scaling sees every output element), 3 = a conv
                                 // input of the fp16 / fp16x3 residual blocks left the fp16 range (denoiser_persist_lp.hip, resblock_fused_lp.hip)
// Reads and clears the device flag word (one atomic exchange: a device store between a read and a clear cannot be lost).
// cmtts_poll_error() reports every code.  A denoiser call checks the word before it launches (before_launch): only code 1 — a neighbour
// wait that expired, i.e. a chip that did not hold the whole grid — refuses the launch; codes 2 / 3 describe the numerics of ONE earlier
// request (possibly another model's, on another stream) and stay in the word for cmtts_poll_error(): they do not fail an unrelated call
// (ADVICE r05).
int check_device_flag(bool before_launch = false) {
    if (!g_tmo_host) return 0;
    unsigned v = __atomic_load_n(g_tmo_host, __ATOMIC_RELAXED);
    if (!v) return 0;
    if (before_launch && v != 1) return 0;
    v = __atomic_exchange_n(g_tmo_host, 0u, __ATOMIC_ACQ_REL);
    if (!v) return 0;
    if (v == 1) return fail(CMTTS_E_HIP, "persistent denoiser: a neighbour wait timed out (the affected utterances' mel is NaN)");
    if (v == 3) return fail(CMTTS_E_HIP, "denoiser, fp16 / fp16x3 operands: a conv input left the fp16 range (|u| > 65504) in an earlier evaluation — the "
                                         "mel is finite but wrong; use bf16 or fp32 for this model / input scale");
    return fail(CMTTS_E_HIP, "denoiser: non-finite mel values in an earlier evaluation (fp16 / fp16x3 operands overflow at 65504: "
                             "use bf16 or fp32 for this model / input scale; fp32: non-finite weights or inputs)");
}

// Persistent launches need ALL of their workgroups resident (one per CU): two grids that together exceed the CU count
// would split the chip and wait for each other's missing neighbours.  Launches of this process on different streams
// are therefore admitted by capacity: a launch waits (stream-side, through events) for the oldest launches in flight on
// OTHER streams until the workgroups still possibly running plus its own fit the chip.  Small grids (bucket groups of a
// ragged shard on separate streams) then run side by side; full-chip launches still run one after the other.
struct PersistInFlight { hipEvent_t ev; hipStream_t stream; int blocks; };
std::vector<PersistInFlight> g_persist_inflight;    // oldest first
std::vector<hipEvent_t> g_persist_free_events;

// Round 6: as long as ONE stream has ever launched persistent grids, its launches run one after the other by themselves and need neither
// admission nor an event behind every launch (an event record is a barrier packet: ~5 us between a launch and the next evaluation's input
// projection, four times per T = 4 sample).  The first launch from a second stream drains the device once and switches the tracking on.
hipStream_t g_persist_only_stream = (hipStream_t)(intptr_t)-1;
bool g_persist_multi = false;
int persist_admit(hipStream_t s, int blocks, int capacity) {
    if (!g_persist_multi) {
        if (g_persist_only_stream == (hipStream_t)(intptr_t)-1) g_persist_only_stream = s;
        if (s == g_persist_only_stream) return 0;
        HIPCHK(hipDeviceSynchronize());          // launches so far carry no event: nothing of them may still be running when tracking starts
        g_persist_multi = true;
    }
    // drop launches that have certainly finished
    for (size_t i = 0; i < g_persist_inflight.size();) {
        if (hipEventQuery(g_persist_inflight[i].ev) == hipSuccess) {
            g_persist_free_events.push_back(g_persist_inflight[i].ev);
            g_persist_inflight.erase(g_persist_inflight.begin() + i);
        } else {
            ++i;
        }
    }
    (void)hipGetLastError();          // hipErrorNotReady from the queries is not an error
    // launches of one stream run one after the other: a stream holds at most its largest launch in flight, and it is
    // certainly idle once its NEWEST launch has finished
    struct PerStream { hipStream_t stream; int blocks; hipEvent_t newest; };
    std::vector<PerStream> per;          // in order of each stream's first launch in flight (oldest first)
    for (const auto& f : g_persist_inflight) {
        if (f.stream == s) continue;     // launches on `s` itself are ordered before this one by the stream
        size_t i = 0;
        while (i < per.size() && per[i].stream != f.stream) ++i;
        if (i == per.size()) per.push_back({f.stream, 0, f.ev});
        per[i].blocks = std::max(per[i].blocks, f.blocks);
        per[i].newest = f.ev;
    }
    int others = 0;
    for (const auto& q : per) others += q.blocks;
    for (const auto& q : per) {
        if (others + blocks <= capacity) break;
        HIPCHK(hipStreamWaitEvent(s, q.newest, 0));
        others -= q.blocks;
    }
    return 0;
}
int persist_launched(hipStream_t s, int blocks) {
    if (!g_persist_multi) return 0;
    hipEvent_t ev;
    if (!g_persist_free_events.empty()) { ev = g_persist_free_events.back(); g_persist_free_events.pop_back(); }
    else HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, s));
    g_persist_inflight.push_back({ev, s, blocks});
    return 0;
}

// Independent branches of the text side (duration / energy predictors, ...) are short launches that cannot fill the chip:
// a branch runs on a side stream forked from and joined back into the caller's stream with events, so the two overlap.
// One side stream per caller stream (callers that run bucket groups on several streams keep their concurrency).
int g_branch_streams = 1;
struct SideStream {
    hipStream_t user, side;
    hipEvent_t fork, join;
    hipStream_t side2 = nullptr;                      // second side stream + chain events: the three ResBlocks of an MRF stage
    hipEvent_t join2 = nullptr, done0 = nullptr, done1 = nullptr;
};
std::vector<SideStream> g_sides;
SideStream* side_for(hipStream_t s) {
    if (!g_branch_streams) return nullptr;
    if (g_sides.capacity() < 16) g_sides.reserve(16);      // callers keep SideStream pointers across nested side_for() calls: never reallocate (only grows while empty)
    for (auto& x : g_sides)
        if (x.user == s) return &x;
    if (g_sides.size() >= 16) return nullptr;          // more caller streams than that: branches run in line
    SideStream x{s, nullptr, nullptr, nullptr};
    // (round 4: a LOW-priority branch stream was tried — the chip-filling conditioner factor GEMM stretches the latency-bound kernels of the
    // main chain — headline 12.66 ms either way, and the ragged shard's set-aside sampler, which lives on this stream, starves: 18 -> 28 ms)
    if (hipStreamCreateWithFlags(&x.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x.join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    g_sides.push_back(x);
    return &g_sides.back();
}
// The vocoder's third stream and chain events, created on first use only: HIP maps streams onto a handful of hardware
// queues in creation order, and idle extra streams made four caller streams (bucket groups) collide (25 -> 39 ms).
bool side2_ready(SideStream* ss) {
    if (ss->side2) return true;
    if (hipStreamCreateWithFlags(&ss->side2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ss->join2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ss->done0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ss->done1, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        ss->side2 = nullptr;
        return false;
    }
    return true;
}
int branch_fork(SideStream* ss) {       // work queued on ss->side after this sees everything queued on ss->user so far
    HIPCHK(hipEventRecord(ss->fork, ss->user));
    HIPCHK(hipStreamWaitEvent(ss->side, ss->fork, 0));
    return 0;
}
int branch_join(SideStream* ss) {       // work queued on ss->user after this sees everything queued on ss->side so far
    HIPCHK(hipEventRecord(ss->join, ss->side));
    HIPCHK(hipStreamWaitEvent(ss->user, ss->join, 0));
    return 0;
}

constexpr long SPLIT_MAX_TILES = 176;    // 32-frame tiles up to which the two-launch residual block wins (tools/split_crossover.py: 0.88 vs 1.86 ms per evaluation up to 64 tiles, 1.46 vs 2.08 at 128, 2.00 vs 2.19 at 192, 2.65 vs 2.39 at 256)

int persist_blocks() {          // workgroups that are certainly co-resident: one 1024-thread workgroup per CU
    static int n = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev);
                        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
    return n;
}

struct EncLayer {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    PackedConv qk, qkv, wo, ffn1, ffn2;     // qkv: the whole in_proj_weight as one [3H][H] contraction (fused attention path)
    float* ffn1_f = nullptr;   // ffn1 as MFMA A fragments in iteration order (conv_xres.hip)
    float* ffn1_q = nullptr;   // ffn1 (k = 9, 256 input channels) as F(4,3) fragments (conv_xres.hip, WQ == 1 instances: to_wino43_xres_fragments), else null
    float* ffn1_p = nullptr;   // ... as F(2,3) fragments (WQ == 2 instances: to_wino23_xres_fragments), else null
    float* qkv_f = nullptr;    // the same for the in-projection and the out-projection (round 2: LayerNorm + projection in one launch)
    float* wo_f = nullptr;
    float* ffn2_f = nullptr;   // the FFN linear as A fragments in iteration order: conv_xres.hip's FFN fusion
    void* ffn1_f16[2] = {nullptr, nullptr};   // bf16 / fp16 fragment-order copies of the two FFN contractions (conv_mfma16.hip; the opt-in "text16")
    void* ffn2_f16[2] = {nullptr, nullptr};
    void* qkv_f16[2] = {nullptr, nullptr};    // ... and of the in- / out-projection of the self-attention
    void* wo_f16[2] = {nullptr, nullptr};
    float* wvT;  // [256 c][256 d]
};
struct Predictor {
    std::vector<PackedConv> convs;
    std::vector<float*> convs_f;     // 256 -> 256 convs as MFMA A fragments in iteration order (conv_xl_kernel), else null
    std::vector<void*> convs_f16[2]; // bf16 / fp16 fragment-order copies (conv_mfma16.hip; the opt-in "text16"), else null
    std::vector<float*> convs_q;     // k = 5 convs into 256 rows: F(4,3) transformed weights (conv_k5q.hip: to_wino43_iter_fragments), else null
    std::vector<float*> ln_g, ln_b;
    float *lin_w = nullptr, *lin_b = nullptr, *alpha = nullptr;
    int odim = 0;
};
struct ResLayer {
    PackedConv cond, conv3, outp;
    float *w3f = nullptr, *wof = nullptr;   // fragment-order copies for the fused kernel
    float* w3w = nullptr;                   // Winograd F(2,3) transformed conv weights as A fragments (persistent denoiser, 8-wave WINO instances)
    float* w3w43 = nullptr;                 // Winograd F(4,3) transformed conv weights (8-wave WINO == 2 instances: to_wino43_fragments)
    float* b3f = nullptr;                   // conv_layer bias in the fused kernel's row order
    void *w3f16[3] = {nullptr, nullptr, nullptr}, *wof16[3] = {nullptr, nullptr, nullptr};   // bf16 / fp16 / fp16x3 (hi | lo) fragment-order copies
};

}  // namespace

struct cmtts_model {
    cmtts_config cfg;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    int precision = 0;     // operand precision of the residual-block contractions: 0 fp32, 1 bf16, 2 fp16
    int text16 = 0;        // 16-bit models (precision 1 / 2): the FFN contractions of the FFT blocks with 16-bit operands too (opt-in: the text side feeds the integer stages — durations, pitch buckets, lengths — which then depend on the precision mode; cmtts_model_set_option)
    int winograd = 1;                       // fp32 persistent denoiser: Winograd k = 3 conv (cmtts_model_set_option "winograd"): 1 = F(4,3) (~8e-6 on the mel against the direct form),
                                            // 2 = F(2,3) (~4e-6), 0 = direct
    int ffn2_split = 1;    // FFT blocks: the FFN linear as 8 partial GEMMs over K segments + one reduction (another fp32 summation order than one launch: a property of the model handle, cmtts_model_set_option)
    cmtts_variance_controls vc = {1.f, 1.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Allocs al;
    float *embed = nullptr, *omega_h = nullptr, *omega_cwt = nullptr, *omega_res = nullptr;
    float *pe_h = nullptr, *pe_cwt = nullptr;   // sinusoid tables [PE_ROWS][C]
    std::vector<EncLayer> enc;
    float *encln_g = nullptr, *encln_b = nullptr;
    // FastspeechDecoder (model/modules.py:154-165): optional, present when the state dict holds "decoder.*"
    std::vector<EncLayer> dec;
    float *decln_g = nullptr, *decln_b = nullptr, *dec_alpha = nullptr;
    float *spk_wt = nullptr, *spk_b = nullptr, *spk_table = nullptr;
    Predictor dur, energy, cwt;
    PackedConv cwt_in;
    float* cwt_in_f = nullptr;     // the same as MFMA A fragments in iteration order (conv_xres.hip)
    float *energy_bins = nullptr, *energy_emb = nullptr, *pitch_emb = nullptr;
    float *st0_wt = nullptr, *st0_b = nullptr, *st2_wt = nullptr, *st2_b = nullptr, *st4_wt = nullptr, *st4_b = nullptr;
    PackedConv in_proj, skip_proj, out_proj;
    float* in_proj_f = nullptr;   // input projection as MFMA A fragments (inproj.hip)
    float *skip_f = nullptr, *outp_f = nullptr;   // skip / output projection in fragment order (persistent kernel's tail)
    PackedConv cond_all;   // the 20 conditioner_projections stacked: [256][NL*256] (+ stacked bias)
    float* cond_all_f = nullptr;   // the same in MFMA A-fragment order (cond_gemm.hip)
    // factored conditioner projections (cond_factored below): P2[r][i] = sum_k Wc[r][k] * pitch_embed[i][k] + bias[r], [NL*C][pitch_bins],
    // computed once at cmtts_finalize with cond_gemm_kernel itself; a zero bias vector for the phoneme-level factor
    float* cond_p2 = nullptr;
    float* cond_p2t = nullptr;             // [NL][pitch_bins][C]: cond_p2 with the channels contiguous (PersistArgs.p2t)
    float* cond_zero_bias = nullptr;
    void* cond_all_f16[3] = {nullptr, nullptr, nullptr};   // bf16 / fp16 / fp16x3 (hi | lo) fragment-order copies (cond_gemm16.hip)
    float *mlp0_wt = nullptr, *mlp2_wt = nullptr, *dproj_wt = nullptr, *sproj_wt = nullptr;
    std::vector<ResLayer> res;
    // Step-embedding cache (round 2): the DiffusionEmbedding -> MLP -> 20 stacked diffusion projections of a timestep depend
    // on nothing but the timestep, and the consistency sampler evaluates every batch at the same few sigmas: the row
    // [NL * C] of each rescaled timestep seen by cmtts_sample is kept on the device (first use computes and copies it; an
    // entry is used once the event recorded behind that copy has completed, whatever stream asks).
    struct StepRow { float t; float* row; hipEvent_t ready; };
    std::vector<StepRow> step_rows;
};

struct cmtts_vocoder {
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    Allocs al;
    PackedConv conv_pre;
    PackedConv ups[4];
    float* ups_f[4] = {nullptr, nullptr, nullptr, nullptr};   // two-tap stacked-phase weights as iteration-order fragments (convT_xl_kernel)
    void* ups_f16[4][3] = {};                                  // the same as bf16 / fp16 / fp16x3 (hi | lo) fragments (convT_xl16_kernel)
    int up_rate[4] = {8, 8, 2, 2};
    int up_kernel[4] = {16, 16, 4, 4};
    int rb_kernel[3] = {3, 7, 11};
    int rb_dil[3] = {1, 3, 5};
    PackedConv c1[12][3], c2[12][3];
    void *c1f[12][3][3] = {}, *c2f[12][3][3] = {};   // bf16 / fp16 / fp16x3 (hi | lo) fragment-order copies of the ResBlock convs
    float *c1f32[12][3] = {}, *c2f32[12][3] = {};    // fp32 fragments in iteration order (resblock_pair.hip: pair kernels at C <= 64, conv_xl above)
    float *c1q32[12][3] = {}, *c2q32[12][3] = {};    // F(4,3) fragments of the dilation-1 convs (conv_xlq_kernel; c1q32 only for the first pair of a ResBlock), else null
    float *c1w32[12][3] = {}, *c2w32[12][3] = {};    // Winograd-transformed fragments of the C >= 128 stages (conv_xlw_kernel), else null
    int winograd = 1;                                 // fp32 generator: ResBlock convs of the C >= 128 stages in their Winograd form (cmtts_vocoder_set_option "winograd")
    int precision = 0;                                // 0 fp32, 1 bf16, 2 fp16 operands in the ResBlock convs
    int ups16 = 1;                                    // 16-bit modes: upsampler operands in 16 bits as well (cmtts_vocoder_set_option "ups16"; 0 = fp32 upsamplers, different numerics)
    float *post_w = nullptr, *post_b = nullptr;
    int post_cin = 32, post_k = 7;
};

namespace {

int set_tensor(std::map<std::string, HostTensor>& host, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!name || !data || ndim < 0 || ndim > 4) return fail(CMTTS_E_INVALID, "set_tensor: bad argument");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0) return fail(CMTTS_E_INVALID, std::string("set_tensor: bad shape for ") + name);
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(data, data + n);
    host[name] = std::move(t);
    return 0;
}

struct Getter {
    const std::map<std::string, HostTensor>& host;
    std::string missing;
    const HostTensor* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = host.find(name);
        if (it == host.end()) {
            if (missing.empty()) missing = "missing tensor " + name;
            return nullptr;
        }
        if (it->second.shape != std::vector<int64_t>(shape)) {
            if (missing.empty()) missing = "wrong shape for tensor " + name;
            return nullptr;
        }
        return &it->second;
    }
};

int finalize_model(cmtts_model* m) {
    const cmtts_config& c = m->cfg;
    if (c.hidden != 256 || c.res_channels != 256 || c.pred_filter != 256)
        return fail(CMTTS_E_UNSUPPORTED, "kernels are specialised for hidden = residual_channels = filter_size = 256");
    if (c.hidden % c.enc_heads) return fail(CMTTS_E_INVALID, "hidden not divisible by heads (model/blocks.py:209)");
    const int H = c.hidden, C = c.res_channels;
    Getter g{m->host, ""};
    Allocs& al = m->al;
#define GET(var, name, ...)                                     \
    const HostTensor* var = g.get(name, {__VA_ARGS__});          \
    if (!var) return fail(CMTTS_E_INVALID, g.missing)
#define UP(dst, t) CHK(al.upload((t)->data, &(dst)))

    CHK(al.upload(omega_table(H), &m->omega_h));
    CHK(al.upload(omega_table(c.cwt_hidden), &m->omega_cwt));
    CHK(al.upload(omega_table(C), &m->omega_res));
    CHK(al.upload(pe_table(H, PE_ROWS), &m->pe_h));
    CHK(al.upload(pe_table(c.cwt_hidden, PE_ROWS), &m->pe_cwt));

    const std::string enc = "duration_pitch_energy_net.text_encoder.";
    GET(emb, enc + "embed_tokens.weight", c.n_symbols, H);
    UP(m->embed, emb);
    // one EncSALayer (model/blocks.py:560-618): shared by the text encoder and the optional FastspeechDecoder
    auto load_fft_layer = [&](const std::string& p, EncLayer& L) -> int {
        GET(l1g, p + "layer_norm1.weight", H); GET(l1b, p + "layer_norm1.bias", H);
        GET(l2g, p + "layer_norm2.weight", H); GET(l2b, p + "layer_norm2.bias", H);
        UP(L.ln1_g, l1g); UP(L.ln1_b, l1b); UP(L.ln2_g, l2g); UP(L.ln2_b, l2b);
        GET(inw, p + "self_attn.in_proj_weight", 3 * H, H);
        HostTensor qk; qk.shape = {2 * H, H, 1};
        qk.data.assign(inw->data.begin(), inw->data.begin() + (size_t)2 * H * H);
        CHK(pack_conv(al, qk, nullptr, nullptr, &L.qk));
        HostTensor qkv = *inw; qkv.shape = {3 * H, H, 1};
        {
            std::vector<float> hp;
            CHK(pack_conv(al, qkv, nullptr, nullptr, &L.qkv, &hp));
            if (H % 32 == 0 && L.qkv.ld == L.qkv.cout) CHK(al.upload(to_fragment_iter_order(hp, 1, H, 3 * H), &L.qkv_f));
            if (H % 32 == 0 && L.qkv.ld == L.qkv.cout)
                for (int mode = 1; mode <= 2; ++mode) {
                    const std::vector<unsigned short> f16 = to_fragment16(hp, 1, H, 3 * H, mode);
                    CHK(al.upload_bytes(f16.data(), f16.size() * 2, &L.qkv_f16[mode - 1]));
                }
        }
        CHK(al.upload(transpose2d(inw->data.data() + (size_t)2 * H * H, H, H), &L.wvT));
        GET(ow, p + "self_attn.out_proj.weight", H, H);
        HostTensor ow3 = *ow; ow3.shape = {H, H, 1};
        {
            std::vector<float> hp;
            CHK(pack_conv(al, ow3, nullptr, nullptr, &L.wo, &hp));
            if (H % 32 == 0 && L.wo.ld == L.wo.cout) CHK(al.upload(to_fragment_iter_order(hp, 1, H, H), &L.wo_f));
            if (H % 32 == 0 && L.wo.ld == L.wo.cout)
                for (int mode = 1; mode <= 2; ++mode) {
                    const std::vector<unsigned short> f16 = to_fragment16(hp, 1, H, H, mode);
                    CHK(al.upload_bytes(f16.data(), f16.size() * 2, &L.wo_f16[mode - 1]));
                }
        }
        GET(f1w, p + "ffn.ffn_1.weight", 4 * H, H, c.ffn_kernel); GET(f1b, p + "ffn.ffn_1.bias", 4 * H);
        {
            std::vector<float> hp;
            CHK(pack_conv(al, *f1w, f1b, nullptr, &L.ffn1, &hp));
            if (L.ffn1.cin % 32 == 0 && L.ffn1.cout % 32 == 0 && L.ffn1.ld == L.ffn1.cout)
                CHK(al.upload(to_fragment_iter_order(hp, L.ffn1.taps, L.ffn1.cin, L.ffn1.cout), &L.ffn1_f));
            if (L.ffn1.taps == 9 && L.ffn1.cin == 256 && L.ffn1.cout % 128 == 0 && L.ffn1.ld == L.ffn1.cout)
            {
                CHK(al.upload(to_wino43_xres_fragments(hp, L.ffn1.taps, L.ffn1.cin, L.ffn1.cout), &L.ffn1_q));
                CHK(al.upload(to_wino23_xres_fragments(hp, L.ffn1.taps, L.ffn1.cin, L.ffn1.cout), &L.ffn1_p));
            }
            if (L.ffn1.cin % 32 == 0 && L.ffn1.cout % 32 == 0 && L.ffn1.ld == L.ffn1.cout)
                for (int mode = 1; mode <= 2; ++mode) {
                    const std::vector<unsigned short> f16 = to_fragment16(hp, L.ffn1.taps, L.ffn1.cin, L.ffn1.cout, mode);
                    CHK(al.upload_bytes(f16.data(), f16.size() * 2, &L.ffn1_f16[mode - 1]));
                }
        }
        GET(f2w, p + "ffn.ffn_2.weight", H, 4 * H); GET(f2b, p + "ffn.ffn_2.bias", H);
        HostTensor f2 = *f2w; f2.shape = {H, 4 * H, 1};
        {
            std::vector<float> hp;
            CHK(pack_conv(al, f2, f2b, nullptr, &L.ffn2, &hp));
            if (H == 256 && L.ffn2.cin % 128 == 0 && L.ffn2.ld == L.ffn2.cout) CHK(al.upload(to_fragment_iter_order(hp, 1, L.ffn2.cin, H), &L.ffn2_f));
            if (L.ffn2.cin % 32 == 0 && H % 32 == 0 && L.ffn2.ld == L.ffn2.cout)
                for (int mode = 1; mode <= 2; ++mode) {
                    const std::vector<unsigned short> f16 = to_fragment16(hp, 1, L.ffn2.cin, H, mode);
                    CHK(al.upload_bytes(f16.data(), f16.size() * 2, &L.ffn2_f16[mode - 1]));
                }
        }
        return 0;
    };
    m->enc.resize(c.enc_layers);
    for (int i = 0; i < c.enc_layers; ++i) CHK(load_fft_layer(enc + "layers." + std::to_string(i) + ".op.", m->enc[i]));
    GET(eg, enc + "layer_norm.weight", H); GET(eb, enc + "layer_norm.bias", H);
    UP(m->encln_g, eg); UP(m->encln_b, eb);
    {   // optional FastspeechDecoder: as many layers as the state dict holds under "decoder.layers.N.op."
        int nd = 0;
        while (m->host.count("decoder.layers." + std::to_string(nd) + ".op.layer_norm1.weight")) ++nd;
        if (nd > 0) {
            m->dec.resize(nd);
            for (int i = 0; i < nd; ++i) CHK(load_fft_layer("decoder.layers." + std::to_string(i) + ".op.", m->dec[i]));
            GET(dg, "decoder.layer_norm.weight", H); GET(db, "decoder.layer_norm.bias", H);
            UP(m->decln_g, dg); UP(m->decln_b, db);
            GET(da, "decoder.pos_embed_alpha", 1);
            UP(m->dec_alpha, da);
        }
    }

    if (c.multi_speaker && c.n_speaker > 0) {   // speaker_embedder "none": nn.Embedding(n_speaker, hidden) (model/cmtts.py:26-38)
        GET(sw, "duration_pitch_energy_net.speaker_emb.weight", c.n_speaker, H);
        UP(m->spk_table, sw);
    } else if (c.multi_speaker) {
        GET(sw, "duration_pitch_energy_net.speaker_emb.weight", H, c.external_speaker_dim);
        GET(sb, "duration_pitch_energy_net.speaker_emb.bias", H);
        CHK(al.upload(transpose2d(sw->data.data(), H, c.external_speaker_dim), &m->spk_wt));
        UP(m->spk_b, sb);
    }

    const std::string va = "duration_pitch_energy_net.variance_adaptor.";
    auto load_pred = [&](Predictor& P, const std::string& p, int idim, int n_layers, int k, int odim, bool alpha) -> int {
        P.convs.resize(n_layers); P.ln_g.resize(n_layers); P.ln_b.resize(n_layers); P.odim = odim;
        for (int li = 0; li < n_layers; ++li) {
            const int cin = li == 0 ? idim : c.pred_filter;
            const std::string q = p + "conv." + std::to_string(li);
            GET(w, q + ".1.weight", c.pred_filter, cin, k); GET(b, q + ".1.bias", c.pred_filter);
            {
                std::vector<float> hp;
                CHK(pack_conv(al, *w, b, nullptr, &P.convs[li], &hp));
                P.convs_f.resize(n_layers, nullptr);
                if ((cin == 256 || (cin == 128 && k == 5)) && c.pred_filter == 256 && P.convs[li].ld == 256)
                    CHK(al.upload(to_fragment_iter_order(hp, k, cin, c.pred_filter), &P.convs_f[li]));
                P.convs_q.resize(n_layers, nullptr);
                if (k == 5 && (cin == 256 || cin == 128) && c.pred_filter == 256 && P.convs[li].ld == 256) {
                    const std::vector<float> wq = to_wino43_iter_fragments(hp, k, cin, c.pred_filter);
                    if (!wq.empty()) CHK(al.upload(wq, &P.convs_q[li]));
                }
                for (int mode = 1; mode <= 2; ++mode) {
                    P.convs_f16[mode - 1].resize(n_layers, nullptr);
                    if (cin % 32 == 0 && c.pred_filter % 32 == 0 && P.convs[li].ld == c.pred_filter) {
                        const std::vector<unsigned short> f16 = to_fragment16(hp, k, cin, c.pred_filter, mode);
                        CHK(al.upload_bytes(f16.data(), f16.size() * 2, &P.convs_f16[mode - 1][li]));
                    }
                }
            }
            GET(lg, q + ".3.weight", c.pred_filter); GET(lb, q + ".3.bias", c.pred_filter);
            UP(P.ln_g[li], lg); UP(P.ln_b[li], lb);
        }
        GET(lw, p + "linear.weight", odim, c.pred_filter); GET(lb2, p + "linear.bias", odim);
        UP(P.lin_w, lw); UP(P.lin_b, lb2);
        if (alpha) { GET(a, p + "pos_embed_alpha", 1); UP(P.alpha, a); }
        return 0;
    };
    CHK(load_pred(m->dur, va + "duration_predictor.", H, c.dur_layers, c.dur_kernel, 1, false));
    CHK(load_pred(m->energy, va + "energy_predictor.", H, c.pred_layers, c.pred_kernel, 1, true));
    const int cwt_out = c.use_uv ? 11 : 10;
    CHK(load_pred(m->cwt, va + "cwt_predictor.1.", c.cwt_hidden, c.pred_layers, c.pred_kernel, cwt_out, true));
    {
        GET(w, va + "cwt_predictor.0.weight", c.cwt_hidden, H); GET(b, va + "cwt_predictor.0.bias", c.cwt_hidden);
        HostTensor w3 = *w; w3.shape = {c.cwt_hidden, H, 1};
        {
            std::vector<float> hp;
            CHK(pack_conv(al, w3, b, nullptr, &m->cwt_in, &hp));
            if (H % 32 == 0 && c.cwt_hidden % 32 == 0 && m->cwt_in.ld == c.cwt_hidden)
                CHK(al.upload(to_fragment_iter_order(hp, 1, H, c.cwt_hidden), &m->cwt_in_f));
        }
        GET(bins, va + "energy_bins", c.energy_bins - 1); UP(m->energy_bins, bins);
        GET(ee, va + "energy_embedding.weight", c.energy_bins, H); UP(m->energy_emb, ee);
        GET(pe, va + "pitch_embed.weight", c.pitch_bins, H); UP(m->pitch_emb, pe);
        GET(s0w, va + "cwt_stats_layers.0.weight", c.cwt_hidden, H); GET(s0b, va + "cwt_stats_layers.0.bias", c.cwt_hidden);
        GET(s2w, va + "cwt_stats_layers.2.weight", c.cwt_hidden, c.cwt_hidden); GET(s2b, va + "cwt_stats_layers.2.bias", c.cwt_hidden);
        GET(s4w, va + "cwt_stats_layers.4.weight", 2, c.cwt_hidden); GET(s4b, va + "cwt_stats_layers.4.bias", 2);
        CHK(al.upload(transpose2d(s0w->data.data(), c.cwt_hidden, H), &m->st0_wt)); UP(m->st0_b, s0b);
        CHK(al.upload(transpose2d(s2w->data.data(), c.cwt_hidden, c.cwt_hidden), &m->st2_wt)); UP(m->st2_b, s2b);
        CHK(al.upload(transpose2d(s4w->data.data(), 2, c.cwt_hidden), &m->st4_wt)); UP(m->st4_b, s4b);
    }

    // ---- denoiser
    {
        GET(w, "net.input_projection.0.conv.weight", C, c.n_mels, 1); GET(b, "net.input_projection.0.conv.bias", C);
        {
            std::vector<float> hp;
            CHK(pack_conv(al, *w, b, nullptr, &m->in_proj, &hp));
            if (c.n_mels % 8 == 0 && C % 32 == 0 && m->in_proj.ld == C) CHK(al.upload(to_fragment_order(hp, 1, c.n_mels, C), &m->in_proj_f));
        }
        GET(m0, "net.mlp.0.linear.weight", 4 * C, C); GET(m2, "net.mlp.2.linear.weight", C, 4 * C);
        CHK(al.upload(transpose2d(m0->data.data(), 4 * C, C), &m->mlp0_wt));
        CHK(al.upload(transpose2d(m2->data.data(), C, 4 * C), &m->mlp2_wt));
    }
    const int NL = c.res_layers;
    m->res.resize(NL);
    std::vector<float> dproj((size_t)C * NL * C), sproj;
    if (c.multi_speaker) sproj.resize((size_t)H * NL * C);
    // gate permutation: packed 64-row group g = [rows g*32.. of the sigmoid half | rows C + g*32.. of the tanh half]
    std::vector<int> perm(2 * C);
    for (int gidx = 0; gidx < C / 32; ++gidx)
        for (int i = 0; i < 32; ++i) {
            perm[gidx * 64 + i] = gidx * 32 + i;
            perm[gidx * 64 + 32 + i] = C + gidx * 32 + i;
        }
    for (int l = 0; l < NL; ++l) {
        const std::string p = "net.residual_layers." + std::to_string(l) + ".";
        GET(w3, p + "conv_layer.conv.weight", 2 * C, C, 3); GET(b3, p + "conv_layer.conv.bias", 2 * C);
        std::vector<float> hp;
        CHK(pack_conv(al, *w3, b3, &perm, &m->res[l].conv3));
        {   // fused kernel: every 32-row tile = [16 sigmoid rows | 16 tanh rows] of the same 16 channels
            std::vector<int> perm16(2 * C);
            for (int gidx = 0; gidx < C / 16; ++gidx)
                for (int i = 0; i < 16; ++i) {
                    perm16[gidx * 32 + i] = gidx * 16 + i;
                    perm16[gidx * 32 + 16 + i] = C + gidx * 16 + i;
                }
            PackedConv tmp;
            Allocs scratch;                                   // device copy of the k-major form is not needed
            CHK(pack_conv(scratch, *w3, b3, &perm16, &tmp, &hp));
            CHK(al.upload(to_fragment_order(hp, 3, C, 2 * C), &m->res[l].w3f));
            if (C == 256) CHK(al.upload(to_wino_fragments(hp, C, 2 * C), &m->res[l].w3w));
            if (C == 256) CHK(al.upload(to_wino43_fragments(hp, C, 2 * C), &m->res[l].w3w43));
            for (int mode = 1; mode <= 2; ++mode) {
                const std::vector<unsigned short> f16 = to_fragment16(hp, 3, C, 2 * C, mode);
                CHK(al.upload_bytes(f16.data(), f16.size() * 2, &m->res[l].w3f16[mode - 1]));
            }
            {
                const std::vector<unsigned short> fs = to_fragment16_split(hp, 3, C, 2 * C);
                CHK(al.upload_bytes(fs.data(), fs.size() * 2, &m->res[l].w3f16[2]));
            }
            std::vector<float> bperm(2 * C);
            for (int r = 0; r < 2 * C; ++r) bperm[r] = b3->data[perm16[r]];
            CHK(al.upload(bperm, &m->res[l].b3f));
            scratch.release();
        }
        GET(wc, p + "conditioner_projection.conv.weight", C, H, 1); GET(bc, p + "conditioner_projection.conv.bias", C);
        CHK(pack_conv(al, *wc, bc, nullptr, &m->res[l].cond));
        GET(wo, p + "output_projection.conv.weight", 2 * C, C, 1); GET(bo, p + "output_projection.conv.bias", 2 * C);
        CHK(pack_conv(al, *wo, bo, nullptr, &m->res[l].outp, &hp));
        CHK(al.upload(to_fragment_order(hp, 1, C, 2 * C), &m->res[l].wof));
        for (int mode = 1; mode <= 2; ++mode) {
            const std::vector<unsigned short> f16 = to_fragment16(hp, 1, C, 2 * C, mode);
            CHK(al.upload_bytes(f16.data(), f16.size() * 2, &m->res[l].wof16[mode - 1]));
        }
        {
            const std::vector<unsigned short> fs = to_fragment16_split(hp, 1, C, 2 * C);
            CHK(al.upload_bytes(fs.data(), fs.size() * 2, &m->res[l].wof16[2]));
        }
        GET(wd, p + "diffusion_projection.linear.weight", C, C);
        for (int n = 0; n < C; ++n)
            for (int k = 0; k < C; ++k) dproj[(size_t)k * NL * C + l * C + n] = wd->data[(size_t)n * C + k];
        if (c.multi_speaker) {
            GET(ws, p + "speaker_projection.linear.weight", C, H);
            for (int n = 0; n < C; ++n)
                for (int k = 0; k < H; ++k) sproj[(size_t)k * NL * C + l * C + n] = ws->data[(size_t)n * H + k];
        }
    }
    {   // stacked conditioner projections (one GEMM for all layers; cond does not depend on the step)
        HostTensor W, Bv;
        W.shape = {(int64_t)NL * C, H, 1};
        W.data.resize((size_t)NL * C * H);
        Bv.shape = {(int64_t)NL * C};
        Bv.data.resize((size_t)NL * C);
        for (int l = 0; l < NL; ++l) {
            const std::string p = "net.residual_layers." + std::to_string(l) + ".";
            const HostTensor& wc = m->host.at(p + "conditioner_projection.conv.weight");
            const HostTensor& bc = m->host.at(p + "conditioner_projection.conv.bias");
            std::copy(wc.data.begin(), wc.data.end(), W.data.begin() + (size_t)l * C * H);
            std::copy(bc.data.begin(), bc.data.end(), Bv.data.begin() + (size_t)l * C);
        }
        std::vector<float> hp;
        CHK(pack_conv(al, W, &Bv, nullptr, &m->cond_all, &hp));
        if (H % 8 == 0 && (NL * C) % 32 == 0 && m->cond_all.ld == NL * C)
            CHK(al.upload(to_fragment_order(hp, 1, H, NL * C), &m->cond_all_f));
        if (H % 16 == 0 && (NL * C) % 32 == 0 && m->cond_all.ld == NL * C)
            for (int mode = 1; mode <= 2; ++mode) {
                const std::vector<unsigned short> f16 = to_fragment16(hp, 1, H, NL * C, mode);
                CHK(al.upload_bytes(f16.data(), f16.size() * 2, &m->cond_all_f16[mode - 1]));
            }
        if (H % 16 == 0 && (NL * C) % 32 == 0 && m->cond_all.ld == NL * C) {
            const std::vector<unsigned short> fs = to_fragment16_split(hp, 1, H, NL * C);
            CHK(al.upload_bytes(fs.data(), fs.size() * 2, &m->cond_all_f16[2]));
        }
    }
    CHK(al.upload(dproj, &m->dproj_wt));
    if (c.multi_speaker) CHK(al.upload(sproj, &m->sproj_wt));
    {
        GET(w, "net.skip_projection.conv.weight", C, C, 1); GET(b, "net.skip_projection.conv.bias", C);
        std::vector<float> hp;
        CHK(pack_conv(al, *w, b, nullptr, &m->skip_proj, &hp));
        if (C % 32 == 0 && m->skip_proj.ld == C) CHK(al.upload(to_fragment_order(hp, 1, C, C), &m->skip_f));
        GET(w2, "net.output_projection.conv.weight", c.n_mels, C, 1); GET(b2, "net.output_projection.conv.bias", c.n_mels);
        CHK(pack_conv(al, *w2, b2, nullptr, &m->out_proj, &hp));
        {   // rows padded to a multiple of 32 with zeros for the MFMA tiles of the fused tail
            const int ld = m->out_proj.ld, Mp = round_up(c.n_mels, 32);
            std::vector<float> padded((size_t)C * Mp, 0.f);
            for (int k = 0; k < C; ++k)
                for (int n = 0; n < c.n_mels; ++n) padded[(size_t)k * Mp + n] = hp[(size_t)k * ld + n];
            CHK(al.upload(to_fragment_order(padded, 1, C, Mp), &m->outp_f));
        }
    }
    if (m->cond_all_f && (NL * C) % 512 == 0) {
        // the pitch-table factor of the conditioner projections: the stacked GEMM on pitch_embed^T [H][pitch_bins] (one "utterance" of
        // pitch_bins "frames"), bias included
        GET(pe, va + "pitch_embed.weight", c.pitch_bins, H);
        float* peT = nullptr;
        void* p2 = nullptr;
        CHK(al.upload(transpose2d(pe->data.data(), c.pitch_bins, H), &peT));
        CHK(al.upload(std::vector<float>((size_t)NL * C, 0.f), &m->cond_zero_bias));
        HIPCHK(hipMalloc(&p2, (size_t)NL * C * c.pitch_bins * sizeof(float) + 256));
        al.ptrs.push_back(p2);
        CondGemmArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.X = peT; ga.Wf = m->cond_all_f; ga.bias = m->cond_all.bias; ga.Y = (float*)p2;
        ga.B = 1; ga.T = c.pitch_bins; ga.M = NL * C; ga.K = H; ga.force = 1; ga.row_split = NL * C / 512;
        if (cmtts_launch_cond_gemm(&ga, nullptr) == 0) {
            void* p2t = nullptr;
            HIPCHK(hipMalloc(&p2t, (size_t)NL * C * c.pitch_bins * sizeof(float) + 256));
            al.ptrs.push_back(p2t);
            k_transpose((const float*)p2, (float*)p2t, NL, C, c.pitch_bins, nullptr);      // [NL][C][bins] -> [NL][bins][C]
            HIPCHK(hipStreamSynchronize(nullptr));
            m->cond_p2 = (float*)p2;
            m->cond_p2t = (float*)p2t;
        }
    }
#undef GET
#undef UP
    m->host.clear();
    m->finalized = true;
    return 0;
}

// ---------------------------------------------------------------- workspaces
// The FFN linear (K = 4 H = 1024 -> H) is DEFINED as FFN2_SEG partial sums over 128-row K segments, added in ascending order,
// then + bias, + residual, mask — for every batch size and path: the segments are independent GEMMs (8x the workgroups of a
// launch whose single 1024-long accumulation chain per tile left most of the chip waiting: 60 us for 1.4 GFLOP) and a
// batch's values still do not depend on its size.
constexpr int FFN2_SEG = 8;
struct TextWs {
    float *x, *h, *qk, *vt, *st, *o, *f, *part, *c1, *c2, *spk, *out1, *h128, *logd, *dround, *epred, *escaled;
    int* cum;
    int64_t *eidx, *mlen;
    size_t bytes;
};
TextWs carve_text(const cmtts_config& c, int B, int L, void* base) {
    const int Lp = round_up(L, 4), H = c.hidden;
    Carver cv(base);
    TextWs w;
    const size_t n = (size_t)B * H * Lp;
    // persistent state (read by cmtts_frame_forward) first
    w.out1 = cv.take<float>(n);
    w.cum = cv.take<int>((size_t)B * L);
    w.spk = cv.take<float>((size_t)B * H);
    w.h128 = cv.take<float>((size_t)B * c.cwt_hidden * Lp);      // cwt_predictor[0] applied at the phoneme level (round 4): [B][cwt_hidden][Lp]
    w.x = cv.take<float>(n);
    w.h = cv.take<float>(n);
    w.qk = cv.take<float>(3 * n);      // fused attention: [B][3H][Lp] (Q | K | V); three-launch path: Q,K [B][2H][Lp] + V^T [B][Lp][H] behind it
    w.vt = w.qk + 2 * n;
    w.st = cv.take<float>((size_t)B * c.enc_heads * Lp * Lp);
    w.o = cv.take<float>(n);
    w.f = cv.take<float>(4 * n);
    w.part = cv.take<float>(FFN2_SEG * n);     // partial sums of the FFN linear, one [B][H][Lp] slab per K segment
    w.c1 = cv.take<float>(n);
    w.c2 = cv.take<float>(n);
    w.logd = cv.take<float>((size_t)B * L);
    w.dround = cv.take<float>((size_t)B * L);
    w.epred = cv.take<float>((size_t)B * L);
    w.eidx = cv.take<int64_t>((size_t)B * L);
    w.mlen = cv.take<int64_t>((size_t)B);
    w.escaled = cv.take<float>((size_t)B * L);      // energy prediction x control (behind everything else: the other offsets do not depend on it)
    w.bytes = cv.off + 256;
    return w;
}

struct FrameWs {
    float *xlr, *h128, *hp, *c1, *c2, *cwt, *r, *s1, *s2, *stats, *f0;
    int64_t* pidx;
    size_t bytes;
};
FrameWs carve_frame(const cmtts_config& c, int B, int T, void* base) {
    Carver cv(base);
    FrameWs w;
    const size_t n = (size_t)B * c.hidden * T;
    w.xlr = cv.take<float>(n);
    w.h128 = cv.take<float>((size_t)B * c.cwt_hidden * T);
    w.hp = cv.take<float>((size_t)B * c.cwt_hidden * T);
    w.c1 = cv.take<float>(n);
    w.c2 = cv.take<float>(n);
    w.cwt = cv.take<float>((size_t)B * T * 16);
    w.r = cv.take<float>((size_t)B * T);
    w.s1 = cv.take<float>((size_t)B * c.cwt_hidden);
    w.s2 = cv.take<float>((size_t)B * c.cwt_hidden);
    w.stats = cv.take<float>((size_t)B * 2);
    w.f0 = cv.take<float>((size_t)B * T);
    w.pidx = cv.take<int64_t>((size_t)B * T);
    w.bytes = cv.off + 256;
    return w;
}

struct DenWs {
    float *hin, *h, *u, *zb, *skip, *emb, *e1, *e2, *dproj, *sproj, *dp, *tbuf, *xcur, *cp;
    float* pst;                 // kernel-private state of the persistent denoiser's Winograd instances
    unsigned long long* halo;   // edge-column granules of the persistent denoiser kernel
    size_t bytes;
};
DenWs carve_den(const cmtts_config& c, int B, int T, void* base) {
    Carver cv(base);
    DenWs w;
    const int C = c.res_channels, NL = c.res_layers;
    const size_t n = (size_t)B * C * T;
    w.hin = cv.take<float>((size_t)B * c.n_mels * T);
    w.h = cv.take<float>(n);
    w.u = cv.take<float>(n);
    w.zb = cv.take<float>(n);
    w.skip = cv.take<float>(n);
    w.emb = cv.take<float>((size_t)B * C);
    w.e1 = cv.take<float>((size_t)B * 4 * C);
    w.e2 = cv.take<float>((size_t)B * C);
    w.dproj = cv.take<float>((size_t)B * NL * C);
    w.sproj = cv.take<float>((size_t)B * NL * C);
    w.dp = cv.take<float>((size_t)B * NL * C);
    w.tbuf = cv.take<float>((size_t)B);
    w.xcur = cv.take<float>((size_t)B * T * c.n_mels);
    w.cp = cv.take<float>((size_t)NL * n);       // conditioner projections of all layers [B][NL*C][T]
    w.halo = cv.take<unsigned long long>(cmtts_persist_halo_bytes(B, T) / sizeof(unsigned long long));
    w.pst = cv.take<float>(cmtts_persist_state_floats(B, T));
    w.bytes = cv.off + 256;
    return w;
}

// conv stack of Duration/Pitch/Energy predictors (model/modules.py:477-487): Conv1d + ReLU ->
// LayerNorm over channels (eps 1e-12) [-> mask].  Result ends in bufB.
// conv -> ReLU -> LayerNorm blocks of a predictor followed by its linear head (model/modules.py:470-506, 520-554): the last
// block's LayerNorm and the head are one launch (ln_linear_kernel) unless cmtts_internal_set("pred_head", 0)
// mode16: 0, or the 16-bit operand mode (1 = bf16, 2 = fp16) of a model with the opt-in "text16": the convs on conv_mfma16.hip (bias + ReLU in fp32)
struct EnergyHead {        // round 6: get_energy_embedding + the embedding add (model/modules.py:318-328,358-363) as the epilogue of the energy predictor's head (kernels.hip: ln_linear_kernel<1, true>)
    const float* xin; const float* e_target; float e_control; const float* bins; int nbins; const float* E; float* out1; int64_t* e_idx; float* e_scaled;
    bool done;
};
int g_energy_head = 1;          // internal switch "energy_head": 1 = in the head's launch (same bits), 0 = energy_embed_kernel behind the join
int g_pred_wino = 1;            // round 6: the frame-level pitch predictor's k = 5 convs as F(4,3) tap groups (conv_k5q.hip; NOT bitwise the direct form: fp32 Winograd rounding), at every launch size
int g_pred_xres = 1;            // round 4: phoneme-level 256 -> 256 predictor convs on conv_xres (32-column tiles), the previous block's LayerNorm as its prologue (same bits); 0 = generic kernel + LayerNorm launches
int predictor(const Predictor& P, const float* in, int ld_in, int B, int T, int ld, const int64_t* ln_lens,
              const int64_t* out_lens, float* bufA, float* bufB, float* out, int O, hipStream_t s, int mode16 = 0, bool frame_level = false,
              EnergyHead* eh = nullptr) {      // eh: the energy predictor — bucketize + embedding add inside the head's launch (eh->done reports it)
    const float* cur = in;
    int ldc = ld_in;
    auto other = [&](const float* p) { return p == bufA ? bufB : bufA; };
    int pend_ln = -1;            // >= 0: `cur` holds block pend_ln's conv + ReLU output, its LayerNorm not yet applied
    for (size_t li = 0; li < P.convs.size(); ++li) {
        const PackedConv& w = P.convs[li];
        int rx = -2;
        float* dst = other(cur);
        // frame-level k = 5 convs (the pitch predictor, round 6): F(4,3) tap groups over frame quads, the pending LayerNorm as the prologue — at
        // EVERY launch size (conv_k5q.hip splits a tile's rows over workgroups when there are few tiles; the bits do not depend on it).  Only the
        // frame-level predictor: the phoneme-level energy predictor (k = 5 too) feeds a bucketize, has 1/6 of the columns and stays in the direct form
        if (frame_level && !mode16 && g_pred_wino && !ln_lens && li < P.convs_q.size() && P.convs_q[li] && w.taps == 5 && w.cout == 256 &&
            (pend_ln < 0 || w.cin == 256)) {
            ConvXlArgs xa;
            memset(&xa, 0, sizeof(xa));
            xa.x = cur; xa.y = dst; xa.wf = P.convs_q[li]; xa.bias = w.bias; xa.bstride = (long)w.cout * ld;
            xa.B = B; xa.C = 256; xa.T = T; xa.ld = ld; xa.k = w.taps; xa.dil = 1; xa.slope = 1.0f; xa.relu = 1;
            if (w.cin != 256) { xa.cin = w.cin; xa.xbstride = (long)w.cin * ldc; }
            if (pend_ln >= 0) { xa.ln_g = P.ln_g[pend_ln]; xa.ln_b = P.ln_b[pend_ln]; xa.ln_eps = 1e-12f; }
            if (ldc == ld || w.cin != 256) {
                rx = cmtts_launch_conv_k5q(&xa, (void*)s);
                if (rx == -3) return fail(CMTTS_E_HIP, "conv_k5q launch failed");
                if (rx == 0) pend_ln = -2;
            }
        }
        // phoneme-level 256 -> 256 convs (round 4): X-resident, one 32-column n-tile per wave, the pending LayerNorm (eps 1e-12, length
        // mask) applied to the staged tile.  (Round 2 tried this with 96-column tiles: 64 workgroups of ~45 us — slower; deleted in round 3.)
        const bool small = (long)((T + 63) / 64) * B < 192;
        if (rx != 0 && !mode16 && g_pred_xres && small && P.convs_f[li] && w.cin == 256 && w.cout == 256 && ldc == ld) {
            ConvArgs a = conv_args(w, cur, T, ldc, (long)w.cin * ldc, dst, ld, (long)w.cout * ld, T);
            a.out[0].act = ACT_RELU;
            a.xres_nt = 1;
            if (pend_ln >= 0) { a.ln_g = P.ln_g[pend_ln]; a.ln_b = P.ln_b[pend_ln]; a.ln_eps = 1e-12f; a.ln_lens = ln_lens; a.ln_skip_tiles = 1; }      // (the next block's LayerNorm / ln_linear masks the same columns)
            rx = cmtts_launch_conv_xres(&a, P.convs_f[li], B, (void*)s);
            if (rx == -3) return fail(CMTTS_E_HIP, "conv_xres launch failed");
            if (rx == 0) pend_ln = -2;       // consumed (if any)
        }
        if (pend_ln >= 0) {                  // the LayerNorm as its own launch
            float* nd = other(cur);
            k_layernorm_ct(cur, nd, P.ln_g[pend_ln], P.ln_b[pend_ln], 1e-12f, ln_lens, B, T, ld, s);
            cur = nd;
            dst = other(cur);
        }
        pend_ln = -1;
        if (mode16 && li < P.convs_f16[mode16 - 1].size() && P.convs_f16[mode16 - 1][li]) {
            ConvArgs a = conv_args(w, cur, T, ldc, (long)w.cin * ldc, dst, ld, (long)w.cout * ld, T);
            a.out[0].act = ACT_RELU;
            a.text_epi = 1;
            rx = cmtts_launch_conv16(&a, P.convs_f16[mode16 - 1][li], mode16, B, (void*)s);
            if (rx == -3) return fail(CMTTS_E_HIP, "text16: predictor conv launch failed");
        }
        if (rx != 0 && g_pred_xl && P.convs_f[li] && ldc == ld && (long)((T + 63) / 64) * B >= 192) {
            // frame-level 256 -> 256 conv: whole x tile + halo resident in LDS, weights streamed as A fragments (the HiFi-GAN
            // kernel, resblock_pair.hip; same accumulation order and epilogue expressions as the generic kernel => same bits)
            ConvXlArgs xa;
            memset(&xa, 0, sizeof(xa));
            xa.x = cur; xa.y = dst; xa.wf = P.convs_f[li]; xa.bias = w.bias; xa.bstride = (long)w.cout * ld;
            xa.B = B; xa.C = 256; xa.T = T; xa.ld = ld; xa.k = w.taps; xa.dil = 1; xa.slope = 1.0f; xa.relu = 1;
            if (w.cin != 256) { xa.cin = w.cin; xa.xbstride = (long)w.cin * ldc; }
            rx = cmtts_launch_conv_xl(&xa, (void*)s);
            if (rx == -3) return fail(CMTTS_E_HIP, "conv_xl launch failed");
        }
        if (rx != 0) {
            ConvArgs a = conv_args(w, cur, T, ldc, (long)w.cin * ldc, dst, ld, (long)w.cout * ld, T);
            a.out[0].act = ACT_RELU;
            CHK(launch(a, EPI_PLAIN, B, s));
        }
        cur = dst;
        ldc = ld;
        if (g_pred_head && eh && g_energy_head && O == 1 && li + 1 == P.convs.size() && w.cout == 256) {
            k_ln_linear_energy(cur, P.ln_g[li], P.ln_b[li], 1e-12f, P.lin_w, P.lin_b, out, ln_lens, out_lens, B, T, ld, eh->xin, eh->e_target, eh->e_control,
                               eh->bins, eh->nbins, eh->E, eh->out1, eh->e_idx, eh->e_scaled, s);
            eh->done = true;
            return 0;
        }
        if (g_pred_head && li + 1 == P.convs.size() && w.cout == 256 &&
            k_ln_linear(cur, P.ln_g[li], P.ln_b[li], 1e-12f, P.lin_w, P.lin_b, out, ln_lens, out_lens, B, T, ld, O, s))
            return 0;
        pend_ln = (int)li;
    }
    if (pend_ln >= 0) {
        float* nd = other(cur);
        k_layernorm_ct(cur, nd, P.ln_g[pend_ln], P.ln_b[pend_ln], 1e-12f, ln_lens, B, T, ld, s);
        cur = nd;
    }
    k_chan_linear(cur, P.lin_w, P.lin_b, out, out_lens, B, P.convs.back().cout, T, ld, O, s);
    return 0;
}

// Denoiser.forward (model/modules.py:600-639) on x_src [B][T][80] scaled by in_scale -> F in w.hin [B][80][T]
// cp[b][l*C + m][t] = conditioner_projection_l(cond)[m][t] + bias: one stacked GEMM, reused by every step
int cond_projections(cmtts_model* m, const DenWs& w, const float* cond_ct, int B, int T, hipStream_t s) {
    const cmtts_config& c = m->cfg;
    if (g_cond_gemm16 && m->precision >= 1 && m->precision <= 3 && m->cond_all_f16[m->precision - 1]) {
        // bf16 / fp16 / fp16x3 models: the conditioner projections are residual-block contractions too — 16-bit operands (pairs), fp32 accumulate and output
        // (round 3; the oracle's operands16 modes quantise the same operands).  Shapes the kernel does not take run the fp32 kernels below.
        CondGemmArgs g;
        g.X = cond_ct; g.Wf = nullptr; g.bias = m->cond_all.bias; g.Y = w.cp;
        g.row_split = 0;
        g.B = B; g.T = T; g.M = c.res_layers * c.res_channels; g.K = c.hidden; g.force = 1;      // every shape: the numerics of a 16-bit model do not depend on the batch
        const int r = cmtts_launch_cond_gemm16(&g, m->cond_all_f16[m->precision - 1], m->precision, (void*)s);
        if (r == 0) return 0;
        if (r != -2) return fail(CMTTS_E_HIP, "cond_gemm16 launch failed");
    }
    if (m->cond_all_f) {   // X tile resident in LDS, one walk over all 20 x 256 rows (bitwise equal to the generic kernel)
        CondGemmArgs g;
        g.X = cond_ct; g.Wf = m->cond_all_f; g.bias = m->cond_all.bias; g.Y = w.cp;
        g.row_split = 0;
        g.B = B; g.T = T; g.M = c.res_layers * c.res_channels; g.K = c.hidden; g.force = g_cond_gemm == 2;
        const int r = g_cond_gemm ? cmtts_launch_cond_gemm(&g, (void*)s) : -2;
        if (r == 0) return 0;
        if (r != -2) return fail(CMTTS_E_HIP, "cond_gemm launch failed");
    }
    ConvArgs a = conv_args(m->cond_all, cond_ct, T, T, (long)c.hidden * T, w.cp, T, (long)c.res_layers * c.res_channels * T, T);
    return launch(a, EPI_PLAIN, B, s);
}

// The conditioner projections from their FACTORS (round 4).  The conditioning the path itself produces is
//   cond[:, t] = out1[:, mel2ph[t] - 1] (length regulator, 0 for padding) + pitch_embed[p_idx[t]]      (model/modules.py:373-395),
// so  Wc * cond[:, t] + b = (Wc * out1)[:, mel2ph[t] - 1] + (Wc * pitch_embed^T + b)[:, p_idx[t]]:  the stacked GEMM runs over the
// PHONEMES (L = T / 6 columns at the bench shape: 7 instead of 43 GFLOP) — cmtts_frame_forward_sub does it beside the frame-level
// predictors — the pitch-table factor is a constant of the model (cmtts_finalize), and the sampler only expands the two into
// cp[b][l*C + m][t] (cond_expand_kernel: one HBM-bound write of the tensor the dense GEMM wrote anyway).  Not bitwise the dense
// GEMM: W (a + b) and W a + W b round differently (relative 1e-7 on cp; tests/test_gpu_parity.py::test_cond_factored).  A caller that
// brings its own conditioning (CMDenoiserTTS.forward, tts_net.py:29-37) has no factors and takes the dense GEMM.
struct CondFactors {
    const float* p1 = nullptr;       // [B][NL*C][ldp]: Wc * out1, no bias
    const float* p1t = nullptr;      // [B][NL][ldp][C]: the same with the channels contiguous (sample_core / sample_ragged transpose it into the unused cp buffer)
    int ldp = 0, L = 0;
    const int64_t* mel2ph = nullptr; // [B][T]
    const int64_t* p_idx = nullptr;  // [B][T]
    bool usable(const cmtts_model* m) const;
};
int g_cond_factored = 1;        // internal switch "cond_factored": 0 = always the dense GEMM
int g_cond_inkernel = 1;        // internal switch "cond_inkernel": the fp32 persistent kernel gathers the factors itself (FACT instances: no cp tensor, the
                                // same bits as cond_expand_kernel + the plain instance); 0 = expand into cp first
inline bool C_IS_256(const cmtts_config& c) { return c.res_channels == 256; }
bool CondFactors::usable(const cmtts_model* m) const {
    return g_cond_factored && p1 && mel2ph && p_idx && m->precision == 0 && m->cond_p2 && g_fused_resblock;
}
int cond_factored(cmtts_model* m, const DenWs& w, const CondFactors& f, int B, int T, hipStream_t s) {
    const cmtts_config& c = m->cfg;
    const int r = cmtts_launch_cond_expand(f.p1, f.ldp, f.L, m->cond_p2, c.pitch_bins, f.mel2ph, f.p_idx, w.cp, B, c.res_layers * c.res_channels, T, (void*)s);
    return r == 0 ? 0 : fail(CMTTS_E_HIP, "cond_expand launch failed");
}
// P1 = Wc * out1 for utterances [b0, b0 + B) of a text workspace: [B][NL*C][Lp]
int cond_phoneme_factor(cmtts_model* m, const float* out1, int B, int Lp, float* p1, hipStream_t s) {
    const cmtts_config& c = m->cfg;
    if (!m->cond_p2) return fail(CMTTS_E_UNSUPPORTED, "factored conditioner projections are not available for this model");
    CondGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.X = out1; g.Wf = m->cond_all_f; g.bias = m->cond_zero_bias; g.Y = p1;
    g.B = B; g.T = Lp; g.M = c.res_layers * c.res_channels; g.K = c.hidden; g.force = 1;
    g.flat = 1;          // 64-column tiles over all utterances' phonemes: 32 x 88 columns = 44 tiles, not 32 x 2
    // workgroups = column tiles x row groups: as close to one per CU as the 512-row passes divide (a workgroup takes a pass in ~38 us)
    const long tiles = ((long)B * Lp + 63) / 64;
    const int npass = g.M / 512;
    int best = 1;
    double best_t = 1e30;
    for (int sp = 1; sp <= npass; ++sp) {
        const int ppg = (npass + sp - 1) / sp, groups = (npass + ppg - 1) / ppg;
        const long wg = tiles * groups, cap = persist_blocks();
        const double t = (double)((wg + cap - 1) / cap) * ppg;          // rounds x passes per workgroup
        if (t < best_t - 1e-9) { best_t = t; best = sp; }
    }
    g.row_split = best;
    const int r = cmtts_launch_cond_gemm(&g, (void*)s);
    return r == 0 ? 0 : fail(r == -2 ? CMTTS_E_UNSUPPORTED : CMTTS_E_HIP, "cond_gemm (phoneme factor) launch failed");
}

// DiffusionEmbedding -> mlp (Linear, Mish, Linear) -> the 20 stacked diffusion (+ speaker) projections
// (model/blocks.py:633-640,669-674; model/modules.py:579-583,626-627).  Depends only on the timesteps and
// the speaker vector, so the sampler re-uses it across evaluations at the same sigma (T = 2, 4: always 80).
// t_host: the (rescaled) timestep every row of `timesteps` holds, when the caller knows it (cmtts_sample); NaN otherwise
int step_embedding(cmtts_model* m, const DenWs& w, const float* timesteps, const float* spk, int B, hipStream_t s,
                   float t_host = NAN) {
    const cmtts_config& c = m->cfg;
    const int C = c.res_channels, NL = c.res_layers;
    if (c.multi_speaker && !spk) return fail(CMTTS_E_INVALID, "speaker_emb is required for a multi-speaker model");
    const bool cacheable = g_step_cache && t_host == t_host;
    if (cacheable) {
        for (const cmtts_model::StepRow& e : m->step_rows)
            if (e.t == t_host && hipEventQuery(e.ready) == hipSuccess) {
                // the same bits as the computation below: every row of dproj is that computation on the same timestep
                k_broadcast_row(e.row, w.dproj, B, NL * C, s);
                if (c.multi_speaker) {   // dp = dproj + speaker projection (dense_small's "sum, then + add")
                    k_dense_small(spk, c.hidden, 1, m->sproj_wt, nullptr, nullptr, w.sproj, B, c.hidden, NL * C, DENSE_NONE, s);
                    k_add_rows(w.dproj, w.sproj, w.dp, (long)B * NL * C, s);
                }
                return 0;
            }
    }
    k_diff_embed(timesteps, m->omega_res, w.emb, B, C, s);
    k_dense_small(w.emb, C, 1, m->mlp0_wt, nullptr, nullptr, w.e1, B, C, 4 * C, DENSE_MISH, s);
    k_dense_small(w.e1, 4 * C, 1, m->mlp2_wt, nullptr, nullptr, w.e2, B, 4 * C, C, DENSE_NONE, s);
    k_dense_small(w.e2, C, 1, m->dproj_wt, nullptr, nullptr, w.dproj, B, C, NL * C, DENSE_NONE, s);
    if (c.multi_speaker) {
        k_dense_small(spk, c.hidden, 1, m->sproj_wt, nullptr, nullptr, w.sproj, B, c.hidden, NL * C, DENSE_NONE, s);
        k_dense_small(w.e2, C, 1, m->dproj_wt, nullptr, w.sproj, w.dp, B, C, NL * C, DENSE_NONE, s);
    }
    if (cacheable && m->step_rows.size() < 16) {
        bool known = false;
        for (const cmtts_model::StepRow& e : m->step_rows) known = known || e.t == t_host;
        if (!known) {
            cmtts_model::StepRow e{t_host, nullptr, nullptr};
            void* p = nullptr;
            if (hipMalloc(&p, (size_t)NL * C * sizeof(float)) == hipSuccess && hipEventCreateWithFlags(&e.ready, hipEventDisableTiming) == hipSuccess) {
                e.row = (float*)p;
                m->al.ptrs.push_back(p);
                HIPCHK(hipMemcpyAsync(e.row, w.dproj, (size_t)NL * C * sizeof(float), hipMemcpyDeviceToDevice, s));
                HIPCHK(hipEventRecord(e.ready, s));
                m->step_rows.push_back(e);
            } else if (p) {
                (void)hipFree(p);
            }
        }
    }
    return 0;
}

// The sampler's post-scaling of the denoiser output (karras_diffusion.py:406,852): out = c_out*F + c_skip*xold (+ noise*nstd*0.85)
struct MelPost {
    const float* xold;
    const float* noise;
    float c_out, c_skip, nstd;
    float* out;
};

// The part of an evaluation in front of the residual layers: c_in scaling + transpose + input projection (+ clearing of the persistent
// kernel's halo granules) and the step embedding.  *halo_zeroed: the granules of this (B, T) batch have been cleared on `s`.
int denoiser_prologue(cmtts_model* m, const DenWs& w, const float* x_src, float in_scale, const float* timesteps, const float* spk, int B,
                      int T, hipStream_t s, bool embed, float t_host, bool* halo_zeroed) {
    const cmtts_config& c = m->cfg;
    const int C = c.res_channels, NL = c.res_layers, M = c.n_mels;
    const long cs = (long)C * T;
    *halo_zeroed = false;
    int rin = -2;
    if (g_inproj_fused && m->in_proj_f) {   // c_in scaling + transpose + input projection (+ halo clearing) in one launch: same bits
        InProjArgs ia;
        memset(&ia, 0, sizeof(ia));
        ia.x = x_src; ia.scale = in_scale; ia.wf = m->in_proj_f; ia.bias = m->in_proj.bias; ia.h = w.h;
        ia.B = B; ia.T = T; ia.M = M; ia.C = C;
        const bool persist_path = g_fused_resblock && g_persist && NL <= PERSIST_MAX_LAYERS;
        if (persist_path) { ia.zero = w.halo; ia.zero_f4 = (long)(cmtts_persist_halo_bytes(B, T) / 16); }
        rin = cmtts_launch_inproj(&ia, (void*)s);
        if (rin == -3) return fail(CMTTS_E_HIP, "inproj launch failed");
        *halo_zeroed = rin == 0 && persist_path && cmtts_persist_halo_bytes(B, T) % 16 == 0;
    }
    if (rin != 0) {
        k_mel_prep(x_src, nullptr, in_scale, w.hin, B, T, M, s);
        ConvArgs a = conv_args(m->in_proj, w.hin, T, T, (long)M * T, w.h, T, cs, T);
        a.out[0].act = ACT_RELU;   // relu(relu(.)) == relu(.), model/modules.py:575-577,624
        CHK(launch(a, EPI_PLAIN, B, s));
    }
    if (embed) CHK(step_embedding(m, w, timesteps, spk, B, s, t_host));
    return 0;
}

int denoiser_core(cmtts_model* m, const DenWs& w, const float* x_src, float in_scale, const float* timesteps,
                  const float* cond_ct, const float* spk, int B, int T, const MelPost& post, hipStream_t s, bool embed = true,
                  SideStream* pending = nullptr, float t_host = NAN,    // pending: a side branch (the conditioner GEMM) to join before the layers
                  const CondFactors* cfk = nullptr, bool* cp_ready = nullptr) {   // cfk: w.cp was NOT filled — the persistent kernel gathers the factors (FACT)
    CHK(check_device_flag(true));
    const cmtts_config& c = m->cfg;
    const int C = c.res_channels, NL = c.res_layers, M = c.n_mels;
    const long cs = (long)C * T;
    bool halo_zeroed = false;
    CHK(denoiser_prologue(m, w, x_src, in_scale, timesteps, spk, B, T, s, embed, t_host, &halo_zeroed));
    if (pending) CHK(branch_join(pending));
    const float* dp = m->cfg.multi_speaker ? w.dp : w.dproj;
    const bool unfused = !g_fused_resblock;   // three-launch form of the residual block (A/B and bitwise tests)
    float* hcur = w.h;
    float* halt = w.u;
    bool layers_done = false;
    if (!unfused && g_persist && NL <= PERSIST_MAX_LAYERS) {
        // Denoiser.forward's layer loop (model/modules.py:626-633) as ONE persistent launch per utterance chunk
        PersistArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.x0 = w.h; pa.cp = w.cp; pa.cp_bstride = (long)NL * C * T;
        pa.dp = dp; pa.d = w.dproj; pa.vec_stride = (long)NL * C;
        pa.skip = w.skip; pa.halo = w.halo; pa.tmo = g_tmo_host;
        pa.B = B; pa.T = T; pa.NL = NL;
        pa.halo_zeroed = halo_zeroed;
        const int prec = m->precision;
        if (cfk && !(cp_ready && *cp_ready)) {
            pa.fact = 1; pa.p1 = cfk->p1; pa.p2 = m->cond_p2; pa.mel2ph = (const long long*)cfk->mel2ph; pa.pidx = (const long long*)cfk->p_idx;
            pa.ldp = cfk->ldp; pa.Lph = cfk->L; pa.ld2 = c.pitch_bins;
            pa.p1t = cfk->p1t; pa.p2t = m->cond_p2t;
        }
        if (g_persist_tail && m->skip_f && m->outp_f) {   // skip head + post-scaling inside the launch
            pa.tail = 1;
            pa.Wsf = m->skip_f; pa.bs = m->skip_proj.bias; pa.Wpf = m->outp_f; pa.bp = m->out_proj.bias;
            pa.skip_div = (float)sqrt((double)NL); pa.n_mels = M;
            pa.xold = post.xold; pa.noise = post.noise; pa.c_out = post.c_out; pa.c_skip = post.c_skip; pa.nstd = post.nstd;
            pa.out = post.out;
        }
        pa.wino = g_persist_wino && m->winograd && !prec && m->res[0].w3w && w.pst;
        if (pa.wino && g_persist_wino == 3 && m->winograd == 1 && m->res[0].w3w43) pa.wino = 3;     // F(4,3) (model option "winograd" = 2 keeps F(2,3))
        pa.xst = w.pst;
        for (int l = 0; l < NL; ++l) {
            pa.W3f[l] = prec ? (const float*)m->res[l].w3f16[prec - 1] : (pa.wino == 3 ? m->res[l].w3w43 : pa.wino ? m->res[l].w3w : m->res[l].w3f);
            pa.Wof[l] = prec ? (const float*)m->res[l].wof16[prec - 1] : m->res[l].wof;
            pa.b3[l] = m->res[l].b3f; pa.bo[l] = m->res[l].outp.bias;
        }
        const bool prof = g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
        // fp16x3 exists as the persistent stack only: shapes that do not take it run the exact fp32 kernels
        const int need = cmtts_persist_plan(B, T, NL, persist_blocks(), g_persist == 2);
        if (need) CHK(persist_admit(s, need, persist_blocks()));
        if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used], s);
        const int rc = prec ? cmtts_launch_denoiser_persist_lp(&pa, prec, persist_blocks(), g_persist == 2, (void*)s)
                            : cmtts_launch_denoiser_persist(&pa, persist_blocks(), g_persist == 2, (void*)s);
        if (rc == -3) return fail(CMTTS_E_HIP, "persistent denoiser launch failed");
        if (rc == 0) {
            CHK(persist_launched(s, need));
            if (prof) { (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s); g_prof.used += 2; }
            layers_done = true;
            if (pa.tail) return 0;
        }
    }
    if (!layers_done && cfk && !(cp_ready && *cp_ready)) {
        // the persistent launch was expected to gather the factors and did not take this shape after all: expand them now
        CHK(cond_factored(m, w, *cfk, B, T, s));
        if (cp_ready) *cp_ready = true;
    }
    for (int l = 0; l < NL && !layers_done; ++l) {
        const ResLayer& R = m->res[l];
        const bool prof = g_prof.on && g_prof.used + 2 <= g_prof.ev.size() && (g_prof.seen++ % g_prof.stride) == 0;
        if (!unfused) {   // ResidualBlock.forward (model/blocks.py:667-686) as one kernel, x ping-pongs
            ResArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.x_in = hcur; ra.cp = w.cp + (long)l * C * T; ra.cp_bstride = (long)NL * C * T;
            ra.dp = dp + (long)l * C; ra.d = w.dproj + (long)l * C;
            ra.x_out = halt; ra.skip = w.skip;
            ra.W3f = R.w3f; ra.b3 = R.b3f; ra.Wof = R.wof; ra.bo = R.outp.bias;
            ra.vec_stride = (long)NL * C; ra.B = B; ra.T = T; ra.accum_skip = l > 0; ra.flag = g_tmo_host;
            if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used], s);
            int lrc;
            if (m->precision == 0 || m->precision == 3) {
                // few 32-frame tiles: one workgroup per tile leaves most of the chip idle for 83 us per layer; four
                // workgroups per tile in two launches finish sooner (measured crossover, tools/latency_bench.py)
                const long tiles32 = (long)((T + 31) / 32) * B;
                ra.z = w.zb;
                if (g_split_resblock == 2 || (g_split_resblock == 1 && tiles32 <= SPLIT_MAX_TILES)) lrc = cmtts_launch_resblock_split(&ra, (void*)s);
                else lrc = cmtts_launch_resblock(&ra, (void*)s);
            } else {
                ra.W3f = (const float*)R.w3f16[m->precision - 1];
                ra.Wof = (const float*)R.wof16[m->precision - 1];
                lrc = cmtts_launch_resblock_lp(&ra, m->precision, (void*)s);
            }
            if (lrc != 0) return fail(CMTTS_E_HIP, "fused residual block launch failed");
            if (prof) { (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s); g_prof.used += 2; }
            float* t = hcur; hcur = halt; halt = t;
            continue;
        }
        {   // u = (x + d [+ p]) + conditioner_projection(cond)      (model/blocks.py:669-677)
            ConvArgs a = conv_args(R.cond, cond_ct, T, T, (long)c.hidden * T, halt, T, cs, T);
            a.out[0].bvec = dp + (long)l * C; a.out[0].bvec_zs = (long)NL * C;
            a.out[0].res = hcur; a.out[0].r_zs0 = cs; a.out[0].ldr = T;
            CHK(launch(a, EPI_PLAIN, B, s));
        }
        {   // z = sigmoid(gate) * tanh(filter) of the k=3 conv        (:675-679)
            ConvArgs a = conv_args(R.conv3, halt, T, T, cs, w.zb, T, cs, T);
            if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used], s);
            CHK(launch(a, EPI_GATED, B, s));
            if (prof) { (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s); g_prof.used += 2; }
        }
        {   // o = output_projection(z); x' = (o[:C] + (x + d)) / sqrt(2); skip += o[C:]   (:681-686)
            ConvArgs a = conv_args(R.outp, w.zb, T, T, cs, hcur, T, cs, T);
            a.split = C;
            a.out[0].res = hcur; a.out[0].r_zs0 = cs; a.out[0].ldr = T;
            a.out[0].bvec = w.dproj + (long)l * C; a.out[0].bvec_zs = (long)NL * C;
            a.out[0].rmul = 0.70710678118654752440f;      // every form of the block scales by the fp32 reciprocal of sqrt(2) (gate.h: CMTTS_RSQRT2)
            a.out[1].Y = w.skip; a.out[1].row_off = C; a.out[1].accum = l > 0;
            CHK(launch(a, EPI_PLAIN, B, s));
        }
    }
    {   // sum(skips)/sqrt(NL) -> skip_projection -> relu -> output_projection   (model/modules.py:634-637)
        ConvArgs a = conv_args(m->skip_proj, w.skip, T, T, cs, halt, T, cs, T);
        a.pre_div = (float)sqrt((double)NL);
        a.out[0].act = ACT_RELU;
        CHK(launch(a, EPI_PLAIN, B, s));
        ConvArgs b = conv_args(m->out_proj, halt, T, T, cs, w.hin, T, (long)M * T, T);
        CHK(launch(b, EPI_PLAIN, B, s));
    }
    k_mel_post(w.hin, post.xold, post.noise, post.c_out, post.c_skip, post.nstd, post.out, B, T, M, g_tmo_host, s);
    return 0;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

const char* cmtts_last_error(void) { return g_err.c_str(); }
int cmtts_internal_fail(int code, const char* msg) { return fail(code, msg ? msg : "?"); }     // for the other translation units (rccl_gather.hip)
const char* cmtts_version(void) { return "cmtts_hip 0.3 (gfx950)"; }
int cmtts_abi_version(void) { return CMTTS_ABI_VERSION; }

int cmtts_create(const cmtts_config* cfg, cmtts_model** out) {
    if (!cfg || !out) return fail(CMTTS_E_INVALID, "cmtts_create: null argument");
    cmtts_model* m = new cmtts_model();
    m->cfg = *cfg;
    *out = m;
    return 0;
}

int cmtts_set_tensor(cmtts_model* m, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    if (!m || m->finalized) return fail(CMTTS_E_INVALID, "cmtts_set_tensor: model is null or already finalized");
    return set_tensor(m->host, name, host_data, shape, ndim);
}

int cmtts_finalize(cmtts_model* m) {
    if (!m || m->finalized) return fail(CMTTS_E_INVALID, "cmtts_finalize: model is null or already finalized");
    const int r = finalize_model(m);
    if (r != 0) m->al.release();
    if (r == 0 && !g_tmo_host) {   // one pinned word for the whole process
        if (hipHostMalloc((void**)&g_tmo_host, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) g_tmo_host = nullptr;
        else *g_tmo_host = 0;
    }
    return r;
}

void cmtts_destroy(cmtts_model* m) {
    if (!m) return;
    for (cmtts_model::StepRow& e : m->step_rows) (void)hipEventDestroy(e.ready);
    m->al.release();
    delete m;
}

size_t cmtts_text_workspace_bytes(const cmtts_model* m, int B, int L) { return carve_text(m->cfg, B, L, nullptr).bytes; }
size_t cmtts_frame_workspace_bytes(const cmtts_model* m, int B, int T) { return carve_frame(m->cfg, B, T, nullptr).bytes; }
size_t cmtts_denoiser_workspace_bytes(const cmtts_model* m, int B, int T) { return carve_den(m->cfg, B, T, nullptr).bytes; }

// FFTBlocks.forward's layer loop (model/modules.py:97-99): pre-LN self-attention + Conv1D FFN blocks over channel-major
// x = w.x [B][H][Lp], masked by `lens`.  Shared by the text encoder (L = phonemes) and the FastspeechDecoder (L = frames).
// pad_lens (ragged text batch, else NULL): columns l >= pad_lens[b] do not exist for utterance b.  The one place of an FFT block where that
// matters: LayerNorm2 turns a masked (zero) column into its bias vector, and the k = 9 FFN conv reads up to four such columns beyond
// src_len — inside the padded batch they hold that bias, beyond it the conv's zero padding (model/blocks.py:612-615, 539-546): the
// normalised tile is zeroed from pad_lens[b] on.  (LayerNorm1 feeds k = 1 projections; padded keys are masked, padded queries dropped.)
int fft_stack(cmtts_model* m, const std::vector<EncLayer>& layers, const TextWs& w, const int64_t* src_lens, int B, int L,
              hipStream_t s, const int64_t* pad_lens = nullptr) {
    const cmtts_config& c = m->cfg;
    const int H = c.hidden, Lp = round_up(L, 4), NH = c.enc_heads, dh = H / NH;
    const long hs = (long)H * Lp;
    // X-resident kernel (conv_xres.hip) for the K = 256 contractions: with 96-column tiles (three n-tiles per wave) when they pad no more
    // than 64-column ones and give every CU a workgroup; round 4: with 32-column tiles (one n-tile per wave, a third of the X tile staged
    // per workgroup) for launches that cannot fill the chip anyway — one request, a few utterances: the LayerNorm prologue, the FFN
    // fusion and a K loop without barriers instead of LayerNorm + generic kernel (+ FFN linear); every path has the same bits
    const int t96 = (L + 95) / 96, t64 = (L + 63) / 64;
    const bool cols96 = t96 * 96 <= t64 * 64;
    const bool xres_small = g_ffn_xres && g_xres_small && (long)t96 * B * 8 < 128;        // the FFN conv's 96-column grid leaves CUs idle
    const bool xres_cols = g_ffn_xres && (cols96 || xres_small);
    for (size_t i = 0; i < layers.size(); ++i) {
        const EncLayer& E = layers[i];
        const bool fused_attn = g_attn_fused && dh == 128;      // L <= 192: all keys in registers; longer: key-chunked online softmax (attention.hip)
        // LayerNorm1 as the prologue of the in-projection: one launch, no normalised copy in HBM
        // opt-in ("text16", cmtts_model_set_option; bf16 / fp16 models): the block's four K = 256 / 1024 contractions (in-projection, out-projection, FFN
        // conv, FFN linear) with 16-bit MFMA operands and fp32 accumulate on conv_mfma16.hip; LayerNorm, softmax, bias, scale, GELU, residuals, masks fp32
        const int pm = m->precision;
        const bool t16 = m->text16 && (pm == 1 || pm == 2) && E.ffn1_f16[pm - 1] && E.ffn2_f16[pm - 1] && E.qkv_f16[pm - 1] && E.wo_f16[pm - 1];
        const bool ln_qkv = !t16 && fused_attn && (g_text_xres & 1) && E.qkv_f && xres_cols && ((long)t96 * (3 * H / 128) * B >= 128 || xres_small);
        if (!ln_qkv) k_layernorm_ct(w.x, w.h, E.ln1_g, E.ln1_b, 1e-12f, nullptr, B, L, Lp, s);
        bool attn_done = false;
        if (fused_attn) {
            // q, k, v = h * W_in^T as ONE contraction, then softmax(q k^T / sqrt(dh) + mask) v in one launch per layer:
            // scores and probabilities never leave the CU (attention.hip)
            int rq = -2;
            if (ln_qkv) {
                ConvArgs a = conv_args(E.qkv, w.x, L, Lp, hs, w.qk, Lp, 3 * hs, L);
                a.ln_g = E.ln1_g; a.ln_b = E.ln1_b; a.ln_eps = 1e-12f;
                if (g_text_xres & 8) {   // debugging aid: the projection on the X-resident kernel behind a separate LayerNorm launch
                    k_layernorm_ct(w.x, w.h, E.ln1_g, E.ln1_b, 1e-12f, nullptr, B, L, Lp, s);
                    a.X = w.h; a.ln_g = a.ln_b = nullptr;
                }
                // more workgroups than CUs with 96-column tiles (B = 64, or two column tiles per utterance): the 32-column instance packs two
                // per CU and overlaps their phases — 45-55 us less per text side at 64 x 85 and 32 x 171 phonemes, neutral at 32 x 85 (tools/text_xres_ab2.py)
                a.xres_nt = g_qkv_nt ? g_qkv_nt : ((long)t96 * (3 * H / 128) * B > persist_blocks() ? 1 : 0);
                rq = cmtts_launch_conv_xres(&a, E.qkv_f, B, (void*)s);
                a.xres_nt = 0;
                if (rq == -3) return fail(CMTTS_E_HIP, "conv_xres launch failed");
                if (rq != 0) k_layernorm_ct(w.x, w.h, E.ln1_g, E.ln1_b, 1e-12f, nullptr, B, L, Lp, s);
            }
            if (rq != 0 && t16) {
                ConvArgs a = conv_args(E.qkv, w.h, L, Lp, hs, w.qk, Lp, 3 * hs, L);
                a.text_epi = 1;
                rq = cmtts_launch_conv16(&a, E.qkv_f16[pm - 1], pm, B, (void*)s);
                if (rq == -3) return fail(CMTTS_E_HIP, "text16: in-projection launch failed");
            }
            if (rq != 0) {
                ConvArgs a = conv_args(E.qkv, w.h, L, Lp, hs, w.qk, Lp, 3 * hs, L);
                CHK(launch(a, EPI_PLAIN, B, s));
            }
            AttnArgs at;
            memset(&at, 0, sizeof(at));
            at.qkv = w.qk; at.out = w.o; at.lens = src_lens; at.bstride = 3 * hs; at.obstride = hs;
            at.B = B; at.H = NH; at.dh = dh; at.L = L; at.ld = Lp; at.scale = (float)(1.0 / sqrt((double)dh));
            const int arc = cmtts_launch_attention(&at, (void*)s);
            if (arc == -3) return fail(CMTTS_E_HIP, "attention launch failed");
            attn_done = arc == 0;
        }
        // the V projection is needed only by the PV product: side stream, joined after the softmax
        SideStream* ss = attn_done ? nullptr : side_for(s);
        if (!attn_done) {
        hipStream_t sv = ss ? ss->side : s;
        if (ss) CHK(branch_fork(ss));
        {   // Q,K = h * W[0:2H]^T, channel-major [B][2H][Lp]
            ConvArgs a = conv_args(E.qk, w.h, L, Lp, hs, w.qk, Lp, 2 * hs, L);
            CHK(launch(a, EPI_PLAIN, B, s));
        }
        {   // V^T[b] = h[b]^T * Wv^T : [L][H]   (A operand = activation, X operand = weights)
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            a.A = w.h; a.a_ld = Lp; a.a_cols = Lp; a.M = L; a.K = H; a.taps = 1; a.dil = 1;
            a.X = E.wvT; a.ldx = H; a.Tin = H; a.N = H;
            a.zdiv = 1; a.a_zs0 = hs; a.x_zs0 = 0;
            a.pre_div = 1.f; a.pre_slope = 1.f; a.split = INT_MAX;
            ConvOut& o = a.out[0];
            o.Y = w.vt; o.y_zs0 = (long)Lp * H; o.ldy = H; o.Tout = H; o.ostride = 1; o.alpha = 1.f; o.div = 1.f;
            CHK(launch(a, EPI_PLAIN, B, sv));
        }
        {   // S^T[b,h][j][i] = sum_d K[d][j] Q[d][i] / sqrt(dh)
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            a.A = w.qk + hs; a.a_ld = Lp; a.a_cols = Lp; a.M = L; a.K = dh; a.taps = 1; a.dil = 1;
            a.X = w.qk; a.ldx = Lp; a.Tin = L; a.N = L;
            a.zdiv = NH; a.a_zs0 = 2 * hs; a.a_zs1 = (long)dh * Lp; a.x_zs0 = 2 * hs; a.x_zs1 = (long)dh * Lp;
            a.pre_div = 1.f; a.pre_slope = 1.f; a.split = INT_MAX;
            ConvOut& o = a.out[0];
            o.Y = w.st; o.y_zs0 = (long)NH * Lp * Lp; o.y_zs1 = (long)Lp * Lp; o.ldy = Lp; o.Tout = L; o.ostride = 1;
            o.alpha = (float)(1.0 / sqrt((double)dh)); o.div = 1.f;
            CHK(launch(a, EPI_PLAIN, B * NH, s));
        }
        k_softmax_cols(w.st, src_lens, B * NH, NH, L, Lp, (long)Lp * Lp, s);
        if (ss) CHK(branch_join(ss));
        {   // O[b,h][d][i] = sum_j V^T[j][d] P^T[j][i]
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            a.A = w.vt; a.a_ld = H; a.a_cols = dh; a.M = dh; a.K = L; a.taps = 1; a.dil = 1;
            a.X = w.st; a.ldx = Lp; a.Tin = L; a.N = L;
            a.zdiv = NH; a.a_zs0 = (long)Lp * H; a.a_zs1 = dh; a.x_zs0 = (long)NH * Lp * Lp; a.x_zs1 = (long)Lp * Lp;
            a.pre_div = 1.f; a.pre_slope = 1.f; a.split = INT_MAX;
            ConvOut& o = a.out[0];
            o.Y = w.o; o.y_zs0 = hs; o.y_zs1 = (long)dh * Lp; o.ldy = Lp; o.Tout = L; o.ostride = 1; o.alpha = 1.f; o.div = 1.f;
            CHK(launch(a, EPI_PLAIN, B * NH, s));
        }
        }   // three-launch attention
        {   // x = (x + out_proj(o)) * nonpad      (model/blocks.py:609-610)
            ConvArgs a = conv_args(E.wo, w.o, L, Lp, hs, w.x, Lp, hs, L);
            a.out[0].res = w.x; a.out[0].r_zs0 = hs; a.out[0].ldr = Lp; a.out[0].lens = src_lens;
            int rc = -2;
            if (t16) {
                a.text_epi = 1;
                rc = cmtts_launch_conv16(&a, E.wo_f16[pm - 1], pm, B, (void*)s);
                if (rc == -3) return fail(CMTTS_E_HIP, "text16: out-projection launch failed");
                a.text_epi = 0;
            } else if (E.wo_f && g_ffn_xres && (g_text_xres & 2)) {
                // round 4: M = 256 is two m-blocks — with 96-column tiles 64 workgroups at B = 32 (measured slower than the generic kernel in round 2);
                // with 32-column tiles 192 workgroups of one short chain each, the tile staged once, no barrier in the K loop: 25 -> 13 us per block
                a.xres_nt = 1;
                rc = cmtts_launch_conv_xres(&a, E.wo_f, B, (void*)s);
                a.xres_nt = 0;
            }
            if (rc == -3) return fail(CMTTS_E_HIP, "conv_xres launch failed");
            if (rc != 0) CHK(launch(a, EPI_PLAIN, B, s));
        }
        if (t16) {      // LayerNorm2, FFN conv (+ k^-0.5, GELU), FFN linear (+ residual, mask): three launches instead of two
            k_layernorm_ct(w.x, w.h, E.ln2_g, E.ln2_b, 1e-12f, pad_lens, B, L, Lp, s);
            ConvArgs a = conv_args(E.ffn1, w.h, L, Lp, hs, w.f, Lp, 4 * hs, L);
            a.out[0].alpha = (float)pow((double)c.ffn_kernel, -0.5);
            a.out[0].act = ACT_GELU_ERF;
            a.text_epi = 1;
            int rc = cmtts_launch_conv16(&a, E.ffn1_f16[pm - 1], pm, B, (void*)s);
            if (rc == 0) {
                ConvArgs b = conv_args(E.ffn2, w.f, L, Lp, 4 * hs, w.x, Lp, hs, L);
                b.out[0].res = w.x; b.out[0].r_zs0 = hs; b.out[0].ldr = Lp; b.out[0].lens = src_lens;
                b.text_epi = 1;
                rc = cmtts_launch_conv16(&b, E.ffn2_f16[pm - 1], pm, B, (void*)s);
                if (rc != 0) return fail(CMTTS_E_HIP, "text16: FFN linear launch failed");
                continue;
            }
            if (rc != -2) return fail(CMTTS_E_HIP, "text16: FFN conv launch failed");
        }
        const bool ffn2_seg = m->ffn2_split && E.ffn2.cin % FFN2_SEG == 0 && E.ffn2.taps == 1;
        bool ffn_fused = false;
        {   // gelu((conv_k9(LayerNorm2(x)) + b) * k^-0.5)      (model/blocks.py:539-546, 612-615)
            // X-resident kernel when it fills the chip; LayerNorm2 is then its prologue
            // round 5: with the F(4,3) form available the X-resident fused launch is taken at EVERY L and B (the Winograd and the direct form differ by fp32
            // rounding: one form for all shapes keeps the text side independent of the batch)
            const float* wqf = g_ffn_wino == 2 ? E.ffn1_q : E.ffn1_p;
            const bool wq = g_ffn_wino && g_ffn_xres && g_ffn_fused && (g_text_xres & 4) && wqf && E.ffn2_f && m->ffn2_split && E.ffn2.cin % FFN2_SEG == 0 &&
                            E.ffn2.taps == 1 && E.ffn1.cout == FFN2_SEG * 128 && H == 256;
            const bool xr = wq || (xres_cols && E.ffn1_f && ((long)t96 * ((E.ffn1.cout + 127) / 128) * B >= 128 || xres_small));
            const bool ln_ffn = xr && (g_text_xres & 4);
            if (!ln_ffn) k_layernorm_ct(w.x, w.h, E.ln2_g, E.ln2_b, 1e-12f, pad_lens, B, L, Lp, s);
            ConvArgs a = conv_args(E.ffn1, ln_ffn ? w.x : w.h, L, Lp, hs, w.f, Lp, 4 * hs, L);
            a.out[0].alpha = (float)pow((double)c.ffn_kernel, -0.5);
            a.out[0].act = ACT_GELU_ERF;
            if (ln_ffn) { a.ln_g = E.ln2_g; a.ln_b = E.ln2_b; a.ln_eps = 1e-12f; a.ln_lens = pad_lens; a.ln_skip_tiles = pad_lens != nullptr; }      // (reduce_partials and the k = 1 linear mask by select)
            int rc = -2;
            if (xr && g_ffn_fused && ffn2_seg && E.ffn2_f && E.ffn1.cout == FFN2_SEG * 128) {
                // ... and the FFN linear's partial products in the same launch: the activated rows never leave the CU
                a.w2frag = E.ffn2_f; a.part = w.part; a.part_zs0 = (long)FFN2_SEG * hs; a.part_zs1 = hs; a.part_ld = Lp; a.M2 = H;
                rc = wq ? cmtts_launch_conv_xresq(&a, wqf, B, (void*)s, g_ffn_wino == 2 ? 1 : 2) : -2;
                if (rc == -2) rc = cmtts_launch_conv_xres(&a, E.ffn1_f, B, (void*)s);
                if (rc == 0) ffn_fused = true;
                else { a.w2frag = nullptr; a.part = nullptr; }
            }
            if (xr && !ffn_fused) rc = cmtts_launch_conv_xres(&a, E.ffn1_f, B, (void*)s);
            if (rc == -3) return fail(CMTTS_E_HIP, "conv_xres launch failed");
            if (rc != 0) {
                if (ln_ffn) {
                    k_layernorm_ct(w.x, w.h, E.ln2_g, E.ln2_b, 1e-12f, pad_lens, B, L, Lp, s);
                    a.X = w.h; a.ln_g = a.ln_b = nullptr; a.ln_lens = nullptr;
                }
                CHK(launch(a, EPI_PLAIN, B, s));
            }
        }
        if (ffn_fused) {   // the partial products are in w.part already
            k_reduce_partials(w.part, FFN2_SEG, E.ffn2.bias, w.x, src_lens, w.x, B, H, L, Lp, s);
        } else if (ffn2_seg) {
            // x = (x + ffn_2(.)) * nonpad          (:551, :616-617) as FFN2_SEG independent partial GEMMs + one reduction
            const int kseg = E.ffn2.cin / FFN2_SEG;
            ConvArgs a = conv_args(E.ffn2, w.f, L, Lp, 4 * hs, w.part, Lp, (long)FFN2_SEG * hs, L);
            a.K = kseg;
            a.zdiv = FFN2_SEG; a.a_zs0 = 0; a.a_zs1 = (long)kseg * a.a_ld; a.x_zs1 = (long)kseg * Lp;
            a.out[0].y_zs1 = hs; a.out[0].bias = nullptr;
            // (64x64 tiles for the 85-phoneme case — a 128-column tile is one third padding — were tried: 59 vs 44 us)
            CHK(launch(a, EPI_PLAIN, B * FFN2_SEG, s));
            k_reduce_partials(w.part, FFN2_SEG, E.ffn2.bias, w.x, src_lens, w.x, B, H, L, Lp, s);
        } else {   // x = (x + ffn_2(.)) * nonpad          (:551, :616-617)
            ConvArgs a = conv_args(E.ffn2, w.f, L, Lp, 4 * hs, w.x, Lp, hs, L);
            a.out[0].res = w.x; a.out[0].r_zs0 = hs; a.out[0].ldr = Lp; a.out[0].lens = src_lens;
            CHK(launch(a, EPI_PLAIN, B, s));
        }
    }
    return 0;
}

int cmtts_text_forward(cmtts_model* m, const int64_t* texts, const int64_t* src_lens, const float* spker_embeds,
                       const int64_t* speakers, int B, int L, float d_control, float* log_d, float* d_rounded, int64_t* mel_len,
                       float* e_pred, int64_t* e_idx, float* enc_out_ct, float* speaker_emb,
                       void* text_ws, size_t text_ws_bytes, void* stream) {
    return cmtts_text_forward_ragged(m, texts, src_lens, nullptr, spker_embeds, speakers, B, L, d_control, log_d, d_rounded, mel_len, e_pred,
                                     e_idx, enc_out_ct, speaker_emb, text_ws, text_ws_bytes, stream);
}

// The phoneme-level half for a RAGGED batch: utterances of several padded groups (bucket groups of a shard, BASELINE.json configs[3])
// in one call, padded to the longest group's L.  pad_lens[b] = the padded phoneme count of utterance b's own group: columns
// l >= pad_lens[b] do not exist for it.  Where the padded length enters the reference's arithmetic — LayerNorm2 of every FFT block makes
// a masked column its bias vector, which the k = 9 FFN conv then reads (fft_stack); the speaker vector is added to every column of the
// padded batch (model/modules.py:349-352); the energy predictor runs unmasked over them (:520-554): the columns src_len <= l < L of a
// group feed the convolutions' halos and are returned — the kernels stop at pad_lens[b]; everything else on this path is column-local
// or masked by src_lens.  Every utterance therefore gets the bits of running its group alone
// (tests/test_gpu_parity.py::test_ragged_text_batch_bitwise), and ~60 latency-bound launches serve the whole shard instead of one
// group.  pad_lens == NULL: the uniform batch (= cmtts_text_forward).
int cmtts_text_forward_ragged(cmtts_model* m, const int64_t* texts, const int64_t* src_lens, const int64_t* pad_lens, const float* spker_embeds,
                              const int64_t* speakers, int B, int L, float d_control, float* log_d, float* d_rounded, int64_t* mel_len,
                              float* e_pred, int64_t* e_idx, float* enc_out_ct, float* speaker_emb,
                              void* text_ws, size_t text_ws_bytes, void* stream) {
    if (!m || !m->finalized) return fail(CMTTS_E_INVALID, "model not finalized");
    if (!texts || !src_lens || !text_ws || B <= 0 || L <= 0) return fail(CMTTS_E_INVALID, "cmtts_text_forward: bad argument");
    const cmtts_config& c = m->cfg;
    if (c.multi_speaker && c.n_speaker > 0 && !speakers) return fail(CMTTS_E_INVALID, "speakers (ids into the speaker_emb table) are required (model/cmtts.py:78)");
    if (c.multi_speaker && c.n_speaker <= 0 && !spker_embeds) return fail(CMTTS_E_INVALID, "Speaker embedding should not be None (model/cmtts.py:80)");
    TextWs w = carve_text(c, B, L, text_ws);
    if (text_ws_bytes < w.bytes) return fail(CMTTS_E_WORKSPACE, "text workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int H = c.hidden, Lp = round_up(L, 4);
    if (!log_d) log_d = w.logd;
    if (!d_rounded) d_rounded = w.dround;
    if (!mel_len) mel_len = w.mlen;
    if (!e_pred) e_pred = w.epred;
    if (!e_idx) e_idx = w.eidx;

    k_embed_tokens(texts, src_lens, m->embed, m->omega_h, m->pe_h, PE_ROWS, w.x, B, L, Lp, H, (float)sqrt((double)H), s);
    CHK(fft_stack(m, m->enc, w, src_lens, B, L, s, pad_lens));
    k_layernorm_ct(w.x, w.x, m->encln_g, m->encln_b, 1e-5f, src_lens, B, L, Lp, s);
    // the encoder output for the caller: a copy of x BEFORE the speaker vector is added; a single-speaker model never modifies x again, so its copy waits until
    // the duration predictor is through (round 6: in the shadow of the longer energy branch instead of in front of both predictors)
    if (enc_out_ct && c.multi_speaker)
        k_copy_rows(enc_out_ct, L, w.x, Lp, L, (long)B * H, s);
    if (c.multi_speaker) {
        if (c.n_speaker > 0) k_gather_rows(m->spk_table, speakers, w.spk, B, H, c.n_speaker, s);
        else k_dense_small(spker_embeds, c.external_speaker_dim, 1, m->spk_wt, m->spk_b, nullptr, w.spk, B,
                      c.external_speaker_dim, H, DENSE_NONE, s);
        k_add_rowvec(w.x, w.spk, B, H, L, Lp, s, pad_lens);
        if (speaker_emb) HIPCHK(hipMemcpyAsync(speaker_emb, w.spk, (size_t)B * H * 4, hipMemcpyDeviceToDevice, s));
    }
    // The duration and the energy predictor both read x and nothing of each other: the energy branch runs on the side
    // stream with its own scratch (the encoder's q/k buffer is free by now)
    SideStream* ss = side_for(s);
    hipStream_t se = ss ? ss->side : s;
    float* ec1 = ss ? w.qk : w.c1;
    float* ec2 = ss ? w.qk + (size_t)B * H * Lp : w.c2;
    if (ss) CHK(branch_fork(ss));
    // duration predictor (masked) -> log_d
    const int t16mode = (m->text16 && (m->precision == 1 || m->precision == 2)) ? m->precision : 0;
    CHK(predictor(m->dur, w.x, Lp, B, L, Lp, src_lens, src_lens, w.c1, w.c2, log_d, 1, s, t16mode));
    // durations -> cumulative frame counts, mel_len: they need log_d only and run here, in the shadow of the (longer) energy branch (round 6; they
    // sat behind the join, the energy embedding and the pitch predictor's input projection)
    if (m->vc.d_target) {   // teacher-forced durations (model/modules.py:365-367)
        HIPCHK(hipMemcpyAsync(d_rounded, m->vc.d_target, (size_t)B * L * 4, hipMemcpyDeviceToDevice, s));
        k_cumsum_durations(m->vc.d_target, w.cum, mel_len, B, L, s);
    } else {
        k_durations(log_d, d_control, d_rounded, w.cum, mel_len, B, L, s);
    }
    if (enc_out_ct && !c.multi_speaker)
        k_copy_rows(enc_out_ct, L, w.x, Lp, L, (long)B * H, s);
    // energy predictor (unmasked, positions from x[...,0] != 0) -> bucketize -> embedding add
    k_pos_embed_add(w.x, w.h, m->energy.alpha, m->omega_h, m->pe_h, PE_ROWS, B, H, L, Lp, se);
    EnergyHead eh{w.x, m->vc.e_target, m->vc.e_control, m->energy_bins, c.energy_bins - 1, m->energy_emb, w.out1, e_idx, w.escaled, false};
    CHK(predictor(m->energy, w.h, Lp, B, L, Lp, pad_lens, pad_lens, ec1, ec2, e_pred, 1, se, t16mode, false, H == 256 ? &eh : nullptr));
    if (ss) CHK(branch_join(ss));
    if (!eh.done)
        k_energy_embed(w.x, e_pred, w.escaled, m->vc.e_target, m->vc.e_control, m->energy_bins, c.energy_bins - 1, m->energy_emb,
                       w.out1, e_idx, B, H, L, Lp, s);
    {   // cwt_predictor[0]: Linear(H -> cwt_hidden) (model/modules.py:204-205).  The reference applies it to the length-regulated frames;
        // a k = 1 contraction commutes with the gather (frame t copies phoneme mel2ph[t] - 1, a padding frame is W 0 + b = b), so it runs
        // over the L phonemes here and cmtts_frame_forward gathers its output: the same bits (tests/test_gpu_parity.py goldens,
        // test_cwt_in_phoneme_level_bitwise) for a sixth of the work, off the frame-level chain
        ConvArgs a = conv_args(m->cwt_in, w.out1, L, Lp, (long)H * Lp, w.h128, Lp, (long)c.cwt_hidden * Lp, L);
        int rc = -2;
        if (g_ffn_xres && g_pred_xres && m->cwt_in_f) {      // 128 rows = one m-block: 32-column tiles, no barrier in the K loop (same bits)
            a.xres_nt = 1;
            rc = cmtts_launch_conv_xres(&a, m->cwt_in_f, B, (void*)s);
            if (rc == -3) return fail(CMTTS_E_HIP, "conv_xres launch failed");
            a.xres_nt = 0;
        }
        if (rc != 0) CHK(launch(a, EPI_PLAIN, B, s));
    }
    if (!m->vc.e_target && m->vc.e_control != 1.0f)     // the reference returns prediction * control (:326)
        HIPCHK(hipMemcpyAsync(e_pred, w.escaled, (size_t)B * L * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipGetLastError());
    return 0;
}

int cmtts_frame_forward(cmtts_model* m, const void* text_ws, int B, int L, int T, float* cond_ct, int64_t* mel2ph,
                        float* cwt_out, float* f0_denorm, int64_t* p_idx, float* f0_stats, void* frame_ws,
                        size_t frame_ws_bytes, void* stream) {
    return cmtts_frame_forward_sub(m, text_ws, B, L, 0, B, T, cond_ct, mel2ph, cwt_out, f0_denorm, p_idx, f0_stats, nullptr, frame_ws, frame_ws_bytes, stream);
}

// The frame-level half for the sub-batch [b0, b0 + B) of a text workspace that cmtts_text_forward(_ragged) filled for B_all utterances
// padded to L_all phonemes: a bucket group of a ragged shard takes its own padded frame count T (results are defined per padded
// bucket, model/modules.py:429-430).  Every buffer of the text workspace is batch-major, so the sub-batch is a pointer offset.
// cond_p1 (optional, [B][res_layers * res_channels][L_all rounded up to 4]): the phoneme-level factor of the conditioner projections for
// cmtts_sample_factored / cmtts_sample_group — computed on the branch stream beside the frame-level predictors.
int cmtts_frame_forward_sub(cmtts_model* m, const void* text_ws, int B_all, int L_all, int b0, int B, int T, float* cond_ct, int64_t* mel2ph,
                            float* cwt_out, float* f0_denorm, int64_t* p_idx, float* f0_stats, float* cond_p1, void* frame_ws,
                            size_t frame_ws_bytes, void* stream) {
    return cmtts_frame_forward_sub_t(m, text_ws, B_all, L_all, b0, B, T, cond_ct, mel2ph, cwt_out, f0_denorm, p_idx, f0_stats, cond_p1, nullptr, frame_ws,
                                     frame_ws_bytes, stream);
}

// ... and the factor's channel-contiguous copy cond_p1t [B][res_layers][Lp][res_channels] (round 6): written on the branch stream right behind the
// GEMM that produces cond_p1, under the frame-level convs, instead of at the sampler's entry in front of the first evaluation.
int cmtts_frame_forward_sub_t(cmtts_model* m, const void* text_ws, int B_all, int L_all, int b0, int B, int T, float* cond_ct, int64_t* mel2ph,
                              float* cwt_out, float* f0_denorm, int64_t* p_idx, float* f0_stats, float* cond_p1, float* cond_p1t, void* frame_ws,
                              size_t frame_ws_bytes, void* stream) {
    if (!m || !m->finalized) return fail(CMTTS_E_INVALID, "model not finalized");
    if (!text_ws || !frame_ws || !cond_ct || !mel2ph || B <= 0 || L_all <= 0 || T <= 0 || b0 < 0 || b0 + B > B_all)
        return fail(CMTTS_E_INVALID, "cmtts_frame_forward: bad argument");
    const cmtts_config& c = m->cfg;
    const int L = L_all;
    TextWs tw = carve_text(c, B_all, L_all, const_cast<void*>(text_ws));
    tw.out1 += (size_t)b0 * c.hidden * round_up(L_all, 4);
    tw.h128 += (size_t)b0 * c.cwt_hidden * round_up(L_all, 4);
    tw.cum += (size_t)b0 * L_all;
    FrameWs w = carve_frame(c, B, T, frame_ws);
    if (frame_ws_bytes < w.bytes) return fail(CMTTS_E_WORKSPACE, "frame workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int H = c.hidden, Lp = round_up(L, 4), O = c.use_uv ? 11 : 10, CH = c.cwt_hidden;
    if (!cwt_out) cwt_out = w.cwt;
    if (!f0_denorm) f0_denorm = w.f0;
    if (!p_idx) p_idx = w.pidx;
    if (!f0_stats) f0_stats = w.stats;

    // cwt_stats_layers on the first phoneme of output_1 (model/modules.py:212-215,279) read nothing of the frame-level
    // chain below: side stream, joined before pitch_index
    SideStream* ss = side_for(s);
    hipStream_t sst = ss ? ss->side : s;
    if (ss) CHK(branch_fork(ss));
    // the phoneme-level factor first: it fills the chip for ~40 us while the main stream runs its short, latency-bound launches
    // (mel2ph, length regulator, the 256 -> 128 projection, positions); behind the statistics MLP it ran beside the frame-level
    // k = 5 convs instead and both took twice as long (profiles/r04_text_side.md)
    // (a model without the pitch-table factor — odd res_layers at C = 256, hidden != 256, a failed finalize-time GEMM — leaves cond_p1 untouched:
    // CondFactors::usable() is false for it and the sampler takes the dense GEMM, as it did before the factors existed)
    if (cond_p1t && !cond_p1) return fail(CMTTS_E_INVALID, "cmtts_frame_forward_sub_t: cond_p1t needs cond_p1");
    if (cond_p1 && m->cond_p2) {
        CHK(cond_phoneme_factor(m, tw.out1, B, Lp, cond_p1, sst));
        if (cond_p1t) k_transpose(cond_p1, cond_p1t, B * c.res_layers, c.res_channels, Lp, sst);      // [B NL][C][Lp] -> [B NL][Lp][C]
    }
    if (!(g_stats_mlp && k_stats_mlp(tw.out1, (long)H * Lp, Lp, m->st0_wt, m->st0_b, m->st2_wt, m->st2_b, m->st4_wt, m->st4_b, f0_stats, B, H, CH, CH, 2, sst))) {
        k_dense_small(tw.out1, (long)H * Lp, Lp, m->st0_wt, m->st0_b, nullptr, w.s1, B, H, CH, DENSE_RELU, sst);
        k_dense_small(w.s1, CH, 1, m->st2_wt, m->st2_b, nullptr, w.s2, B, CH, CH, DENSE_RELU, sst);
        k_dense_small(w.s2, CH, 1, m->st4_wt, m->st4_b, nullptr, f0_stats, B, CH, 2, DENSE_NONE, sst);
    }
    k_mel2ph(tw.cum, mel2ph, B, L, T, s);
    // with the pitch predictor's input projection applied before the gather the length-regulated [B][H][T] tensor has ONE reader left, the pitch
    // embedding add at the end: that kernel gathers from out1 itself (k_lr_gather_add: the same values, one launch and 2 x 17 MB less)
    if (!g_cwt_in_phoneme) k_length_regulate(tw.out1, mel2ph, w.xlr, B, H, Lp, T, s);
    bool hp_done = false;
    if (g_cwt_in_phoneme) {   // cwt_predictor[0] was applied at the phoneme level (cmtts_text_forward): gather it; padding frames = its bias —
        // inside the position-embedding add that follows (one launch, no [B][128][T] intermediate)
        k_pos_embed_add_lr(tw.h128, Lp, mel2ph, m->cwt_in.bias, w.hp, m->cwt.alpha, m->omega_cwt, m->pe_cwt, PE_ROWS, B, CH, T, s);
        hp_done = true;
    } else {   // cwt_predictor[0]: Linear(H -> cwt_hidden) over the frames       (model/modules.py:204-205)
        ConvArgs a = conv_args(m->cwt_in, w.xlr, T, T, (long)H * T, w.h128, T, (long)CH * T, T);
        CHK(launch(a, EPI_PLAIN, B, s));
    }
    if (!hp_done) k_pos_embed_add(w.h128, w.hp, m->cwt.alpha, m->omega_cwt, m->pe_cwt, PE_ROWS, B, CH, T, T, s);
    CHK(predictor(m->cwt, w.hp, T, B, T, T, nullptr, nullptr, w.c1, w.c2, cwt_out, O, s, (m->text16 && (m->precision == 1 || m->precision == 2)) ? m->precision : 0, true));
    if (ss) CHK(branch_join(ss));
    if (m->vc.p_control != 1.0f) k_scale(cwt_out, cwt_out, (long)B * T * O, m->vc.p_control, s);   // :270
    if (m->vc.cwt_spec) {   // teacher-forced pitch: target spectrogram, statistics and uv (:379-390)
        k_pitch_index(m->vc.cwt_spec, 10, m->vc.f0_mean, m->vc.f0_std, 1, 1.0f, nullptr, 0, c.use_uv ? m->vc.uv : nullptr,
                      c.pitch_norm_eps, w.r, p_idx, f0_denorm, B, T, s);
    } else {
        k_pitch_index(cwt_out, O, f0_stats, f0_stats + 1, 2, c.cwt_std_scale, c.use_uv ? cwt_out + (O - 1) : nullptr, O, nullptr,
                      c.pitch_norm_eps, w.r, p_idx, f0_denorm, B, T, s);
    }
    if (g_cwt_in_phoneme) k_lr_gather_add(tw.out1, mel2ph, Lp, p_idx, m->pitch_emb, cond_ct, B, H, T, s);
    else k_gather_add(w.xlr, p_idx, m->pitch_emb, cond_ct, B, H, T, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int cmtts_length_regulate(const float* x_ct, const float* durations, int B, int C, int L, int T, float* out_ct,
                          int64_t* mel2ph, int64_t* mel_len, int32_t* scratch_cum, void* stream) {
    if (!x_ct || !durations || !out_ct || !mel2ph || !mel_len || !scratch_cum || B <= 0 || L <= 0 || T <= 0)
        return fail(CMTTS_E_INVALID, "cmtts_length_regulate: bad argument");
    hipStream_t s = (hipStream_t)stream;
    k_cumsum_durations(durations, scratch_cum, mel_len, B, L, s);
    k_mel2ph(scratch_cum, mel2ph, B, L, T, s);
    k_length_regulate(x_ct, mel2ph, out_ct, B, C, L, T, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int cmtts_denoiser_forward(cmtts_model* m, const float* x, const float* timesteps, const float* cond_ct,
                           const float* speaker_emb, int B, int T, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!m || !m->finalized) return fail(CMTTS_E_INVALID, "model not finalized");
    if (!x || !timesteps || !cond_ct || !out || !ws || B <= 0 || T <= 0)
        return fail(CMTTS_E_INVALID, "cmtts_denoiser_forward: bad argument");
    DenWs w = carve_den(m->cfg, B, T, ws);
    if (ws_bytes < w.bytes) return fail(CMTTS_E_WORKSPACE, "denoiser workspace too small");
    hipStream_t s = (hipStream_t)stream;
    // the conditioner GEMM (all layers' projections of cond) and the input projection + step-embedding MLP are
    // independent: the GEMM runs on the side stream and is joined before the first residual layer
    SideStream* ss = g_fused_resblock ? side_for(s) : nullptr;
    if (ss) CHK(branch_fork(ss));
    if (g_fused_resblock) CHK(cond_projections(m, w, cond_ct, B, T, ss ? ss->side : s));
    const MelPost post = {nullptr, nullptr, 1.0f, 0.0f, 0.0f, out};
    CHK(denoiser_core(m, w, x, 1.0f, timesteps, cond_ct, speaker_emb, B, T, post, s, true, ss));
    HIPCHK(hipGetLastError());
    return 0;
}

int cmtts_schedule(const cmtts_model* m, int n_steps, float* sigmas, float* renoise_std) {
    if (!m || !sigmas || !renoise_std || n_steps < 1) return fail(CMTTS_E_INVALID, "cmtts_schedule: bad argument");
    const cmtts_config& c = m->cfg;
    if (n_steps == 1) {   // sample_onestep: sigmas[0] = sigma_max (get_sigmas_karras, karras_diffusion.py:580-586)
        sigmas[0] = c.sigma_max;
        renoise_std[0] = -1.0f;   // no re-noising
        return 0;
    }
    // synthesize.py:122-147: sampler="multistep", steps=2, ts=(0,)*T+(1,)
    const double rho = c.rho, tmax = pow((double)c.sigma_max, 1.0 / rho), tmin = pow((double)c.sigma_min, 1.0 / rho);
    for (int i = 0; i < n_steps; ++i) {
        const double tsi = 0.0, tsn = (i + 1 == n_steps) ? 1.0 : 0.0;
        const double t = pow(tmax + tsi / 1.0 * (tmin - tmax), rho);
        double nt = pow(tmax + tsn / 1.0 * (tmin - tmax), rho);
        nt = nt < c.sigma_min ? (double)c.sigma_min : (nt > c.sigma_max ? (double)c.sigma_max : nt);
        sigmas[i] = (float)t;
        renoise_std[i] = (float)sqrt(nt * nt - (double)c.sigma_min * (double)c.sigma_min);   // x0.85 is applied in-kernel
    }
    return 0;
}

}  // extern "C" (reopened below)
namespace {
// The sampler on one padded (B, T) batch whose workspace is already carved.  noise_stride = elements between consecutive noise tensors
// (B * T * n_mels for a whole batch; a sub-batch [b0, b0 + B) of a larger batch keeps the larger batch's stride).
int sample_core(cmtts_model* m, const DenWs& w, const float* noise, long noise_stride, const float* cond_ct, const float* speaker_emb, int B,
                int T, int n_steps, const float* sigmas, const float* renoise_std, float* mel, hipStream_t s, const CondFactors* cf = nullptr) {
    const cmtts_config& c = m->cfg;
    const long nel = (long)B * T * c.n_mels;
    // once for all n_steps evaluations, on the side stream: joined before the first residual layer of the first evaluation
    SideStream* ss = g_fused_resblock ? side_for(s) : nullptr;
    if (ss) CHK(branch_fork(ss));
    const CondFactors* cfk = nullptr;       // non-null: no cp tensor at all, the persistent launches gather the factors (denoiser_core)
    CondFactors cf_local;
    bool cp_ready = false;
    if (g_fused_resblock) {
        if (cf && cf->usable(m)) {
            const bool inkernel = g_cond_inkernel && g_persist && g_persist_tail && m->skip_f && m->outp_f && c.res_layers <= PERSIST_MAX_LAYERS &&
                                  cmtts_persist_plan(B, T, c.res_layers, persist_blocks(), g_persist == 2) > 0;
            if (inkernel && cf->ldp <= T && m->cond_p2t && C_IS_256(c)) {
                // the persistent launches gather the factors: w.cp (NL * C * T floats per utterance) is free and takes p1 with the channels
                // contiguous, [B][NL][ldp][C] — one 57-MB transpose per sample call (bench shape, ~25 us) buys 16-byte gathers in every layer of
                // every evaluation
                // (round 6: on the side stream — nothing before the first persistent launch reads it; it ran in front of the input projection)
                cf_local = *cf;
                if (!cf_local.p1t) {
                    k_transpose(cf->p1, w.cp, B * c.res_layers, c.res_channels, cf->ldp, ss ? ss->side : s);
                    cf_local.p1t = w.cp;
                }
                cfk = &cf_local;
            } else CHK(cond_factored(m, w, *cf, B, T, ss ? ss->side : s));
        } else CHK(cond_projections(m, w, cond_ct, B, T, ss ? ss->side : s));
    }
    k_scale(noise, w.xcur, nel, c.sigma_max, s);        // x_T = randn * sigma_max (karras_diffusion.py:534)
    const float smin = c.sigma_min, sd2 = c.sigma_data * c.sigma_data;
    for (int i = 0; i < n_steps; ++i) {
        // get_scalings_for_boundary_condition in fp32 (karras_diffusion.py:87-102)
        const float sg = sigmas[i];
        const float dm = sg - smin;
        const float c_skip = sd2 / (dm * dm + sd2);
        const float rt = sqrtf(sg * sg + sd2);
        const float c_out = dm * c.sigma_data / rt;
        const float c_in = 1.0f / rt;
        const float t_resc = 250.0f * logf(sg + 1e-44f);
        const bool new_sigma = i == 0 || sigmas[i] != sigmas[i - 1];
        // the first evaluation's step embedding reads only the timestep (and the speaker vector): side stream, beside x_T's scaling and the
        // input projection, joined with the conditioner branch before the residual layers (round 6; same kernels, same bits)
        const bool embed_side = i == 0 && ss != nullptr;
        if (new_sigma) k_fill_float(w.tbuf, t_resc, B, embed_side ? ss->side : s);
        if (embed_side) CHK(step_embedding(m, w, w.tbuf, speaker_emb, B, ss->side, t_resc));
        const bool last = i + 1 == n_steps;
        const bool renoise = renoise_std[i] >= 0.0f;
        const MelPost post = {w.xcur, renoise ? noise + (long)(1 + i) * noise_stride : nullptr, c_out, c_skip,
                              renoise ? renoise_std[i] : 0.0f, last ? mel : w.xcur};
        CHK(denoiser_core(m, w, w.xcur, c_in, w.tbuf, cond_ct, speaker_emb, B, T, post, s, new_sigma && !embed_side, i == 0 ? ss : nullptr, t_resc, cfk, &cp_ready));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// The workspace of utterances [b0, B) of a batch carved for (B, T): every buffer is batch-major, so a sub-batch is a pointer offset
// (the halo granules are the persistent kernel's and are not used by the sub-batch path).
DenWs den_slice(const cmtts_config& c, const DenWs& w, int b0, int T) {
    const long C = c.res_channels, NL = c.res_layers, M = c.n_mels;
    DenWs v = w;
    v.hin += (long)b0 * M * T; v.h += b0 * C * T; v.u += b0 * C * T; v.zb += b0 * C * T; v.skip += b0 * C * T;
    v.emb += b0 * C; v.e1 += b0 * 4 * C; v.e2 += b0 * C; v.dproj += b0 * NL * C; v.sproj += b0 * NL * C; v.dp += b0 * NL * C;
    v.tbuf += b0; v.xcur += (long)b0 * T * M; v.cp += (long)b0 * NL * C * T;
    return v;
}
}  // namespace
extern "C" {

int cmtts_sample(cmtts_model* m, const float* noise, const float* cond_ct, const float* speaker_emb, int B, int T,
                 int n_steps, const float* sigmas, const float* renoise_std, float* mel, void* ws, size_t ws_bytes,
                 void* stream) {
    if (!m || !m->finalized) return fail(CMTTS_E_INVALID, "model not finalized");
    if (!noise || !cond_ct || !mel || !ws || !sigmas || !renoise_std || B <= 0 || T <= 0 || n_steps < 1)
        return fail(CMTTS_E_INVALID, "cmtts_sample: bad argument");
    const cmtts_config& c = m->cfg;
    DenWs w = carve_den(c, B, T, ws);
    if (ws_bytes < w.bytes) return fail(CMTTS_E_WORKSPACE, "denoiser workspace too small");
    return sample_core(m, w, noise, (long)B * T * c.n_mels, cond_ct, speaker_emb, B, T, n_steps, sigmas, renoise_std, mel, (hipStream_t)stream);
}

// cmtts_sample for conditioning that cmtts_frame_forward_sub produced together with its phoneme-level factor `cond_p1`: the stacked
// conditioner GEMM over the frames is replaced by the expansion of the factors (cond_factored above).  cond_ct is still required
// (16-bit models and the unfused path use it); a NULL cond_p1 is cmtts_sample.
int cmtts_sample_factored(cmtts_model* m, const float* noise, const float* cond_ct, const float* speaker_emb, int B, int T,
                          int n_steps, const float* sigmas, const float* renoise_std, float* mel, void* ws, size_t ws_bytes,
                          void* stream, const float* cond_p1, int p1_ld, int L, const int64_t* mel2ph, const int64_t* p_idx) {
    return cmtts_sample_factored_t(m, noise, cond_ct, speaker_emb, B, T, n_steps, sigmas, renoise_std, mel, ws, ws_bytes, stream, cond_p1, nullptr, p1_ld, L,
                                   mel2ph, p_idx);
}

int cmtts_sample_factored_t(cmtts_model* m, const float* noise, const float* cond_ct, const float* speaker_emb, int B, int T,
                            int n_steps, const float* sigmas, const float* renoise_std, float* mel, void* ws, size_t ws_bytes,
                            void* stream, const float* cond_p1, const float* cond_p1t, int p1_ld, int L, const int64_t* mel2ph, const int64_t* p_idx) {
    if (!m || !m->finalized) return fail(CMTTS_E_INVALID, "model not finalized");
    if (!noise || !cond_ct || !mel || !ws || !sigmas || !renoise_std || B <= 0 || T <= 0 || n_steps < 1)
        return fail(CMTTS_E_INVALID, "cmtts_sample_factored: bad argument");
    if (cond_p1 && (!mel2ph || !p_idx || L <= 0 || p1_ld < L)) return fail(CMTTS_E_INVALID, "cmtts_sample_factored: incomplete factors");
    const cmtts_config& c = m->cfg;
    DenWs w = carve_den(c, B, T, ws);
    if (ws_bytes < w.bytes) return fail(CMTTS_E_WORKSPACE, "denoiser workspace too small");
    CondFactors cf;
    cf.p1 = cond_p1; cf.ldp = p1_ld; cf.L = L; cf.mel2ph = mel2ph; cf.p_idx = p_idx;
    cf.p1t = cond_p1 ? cond_p1t : nullptr;      // the caller's channel-contiguous copy (cmtts_frame_forward_sub_t): no transpose at the sampler's entry
    return sample_core(m, w, noise, (long)B * T * c.n_mels, cond_ct, speaker_emb, B, T, n_steps, sigmas, renoise_std, mel, (hipStream_t)stream,
                       cond_p1 ? &cf : nullptr);
}

// karras_sample_tts for a RAGGED shard (BASELINE.json configs[3]: utterances dealt into static frame buckets): every group is a
// padded (B, T) batch with its own buffers — results are defined per padded bucket (model/modules.py:429-430 via model/cmtts.py:61-62)
// — but the residual layers of ALL groups run in ONE persistent launch per evaluation (denoiser_persist.hip, RAGGED instance), so that
// small buckets fill the chip together instead of one after the other.
int cmtts_sample_ragged(cmtts_model* m, const cmtts_sample_group* groups, int n_groups, int n_steps, const float* sigmas,
                        const float* renoise_std, int tail_frames, void* stream) {
    if (!m || !m->finalized) return fail(CMTTS_E_INVALID, "model not finalized");
    if (!groups || n_groups < 1 || n_steps < 1 || !sigmas || !renoise_std || tail_frames < 0)
        return fail(CMTTS_E_INVALID, "cmtts_sample_ragged: bad argument");
    const cmtts_config& c = m->cfg;
    const int C = c.res_channels, NL = c.res_layers, M = c.n_mels;
    hipStream_t s = (hipStream_t)stream;
    long padded_tiles = 0;
    for (int g = 0; g < n_groups; ++g) {
        const cmtts_sample_group& G = groups[g];
        if (!G.noise || !G.cond_ct || !G.mel || !G.ws || G.B <= 0 || G.T <= 0) return fail(CMTTS_E_INVALID, "cmtts_sample_ragged: bad group");
        if (c.multi_speaker && !G.speaker_emb) return fail(CMTTS_E_INVALID, "speaker_emb is required for a multi-speaker model");
        if (G.ws_bytes < carve_den(c, G.B, G.T, nullptr).bytes) return fail(CMTTS_E_WORKSPACE, "denoiser workspace too small");
        padded_tiles += (long)G.B * ((G.T + 63) / 64);
    }
    // the one-launch form needs the fp32 persistent kernel with its in-kernel tail; anything else (16-bit modes, switches off, more
    // groups / longer utterances than a descriptor holds, too little work to beat the per-layer kernels) runs group after group
    bool one_launch = m->precision == 0 && g_fused_resblock && g_persist && g_persist_tail && g_inproj_fused && m->skip_f && m->outp_f &&
                      m->in_proj_f && NL <= PERSIST_MAX_LAYERS && n_groups <= PERSIST_MAX_GROUPS && padded_tiles * 2 > persist_blocks();
    // (an utterance with more tiles than the launch can keep resident is known NOW: fall back before anything is queued — ADVICE r03)
    for (int g = 0; g < n_groups && one_launch; ++g)
        if ((groups[g].T + 63) / 64 > 127 || (groups[g].T + 63) / 64 > std::min(persist_blocks(), PERSIST_MAX_WG) || groups[g].B > 1023 || (long)C * groups[g].T >= (1L << 30) ||
            cmtts_persist_halo_bytes(groups[g].B, groups[g].T) % 16 != 0)
            one_launch = false;
    if (!one_launch) {
        for (int g = 0; g < n_groups; ++g) {
            const cmtts_sample_group& G = groups[g];
            const int rc = cmtts_sample_factored(m, G.noise, G.cond_ct, G.speaker_emb, G.B, G.T, n_steps, sigmas, renoise_std, G.mel, G.ws, G.ws_bytes,
                                                 stream, G.cond_p1, G.p1_ld, G.L, G.mel2ph, G.p_idx);
            if (rc != 0) return rc;
        }
        return 0;
    }
    CHK(check_device_flag(true));
    // ---- which utterances share the persistent launch.  Every workgroup of that launch must be resident, so a shard with more active
    // tiles than CUs needs a second ROUND of 20 layers (2.7 ms per evaluation whatever its size).  When leaving out a few SMALL
    // utterances (<= 64 active tiles together: the range where the per-layer / split kernels take < 1 ms per evaluation,
    // tools/split_crossover.py) makes the rest fit one round, those go through the ordinary sampler on the side stream instead: 2.7 +
    // ~0.9 ms per evaluation where two rounds cost 5.4.  They are taken from the END of the groups with the shortest padded length
    // (a sub-batch [b0, B) of a group is a pointer offset: every buffer is batch-major); keep[g] = utterances of group g that stay.
    const int cap_all = std::min(persist_blocks(), PERSIST_MAX_WG);
    auto utt_tiles = [&](const cmtts_sample_group& G, int b, int eval) {
        const int tiles = (G.T + 63) / 64;
        if (!G.active_frames) return (long)tiles;
        const long need = (long)G.active_frames[b] + tail_frames + (long)NL * (n_steps - eval);
        return std::min<long>(tiles, std::max<long>(1, (need + 63) / 64));
    };
    std::vector<int> keep(n_groups);
    {
        long total = 0;
        for (int g = 0; g < n_groups; ++g) {
            keep[g] = groups[g].B;
            for (int b = 0; b < groups[g].B; ++b) total += utt_tiles(groups[g], b, 0);
        }
        if (total > cap_all) {
            std::vector<int> order(n_groups), k2 = keep;
            for (int g = 0; g < n_groups; ++g) order[g] = g;
            std::sort(order.begin(), order.end(), [&](int x, int y) { return groups[x].T < groups[y].T; });
            long out = 0, rest = total;
            for (int g : order) {
                while (rest > cap_all && k2[g] > 0) {
                    const long t = utt_tiles(groups[g], k2[g] - 1, 0);
                    out += t; rest -= t; --k2[g];
                }
                if (rest <= cap_all) break;
            }
            if (rest <= cap_all && rest > 0 && out <= 64) keep = k2;
        }
    }
    // every group that takes part in the one launch brings usable conditioner factors: the FACT instance of the ragged kernel
    bool fact_all = g_cond_inkernel != 0;
    for (int g = 0; g < n_groups && fact_all; ++g) {
        if (keep[g] == 0) continue;
        CondFactors cf;
        cf.p1 = groups[g].cond_p1; cf.ldp = groups[g].p1_ld; cf.L = groups[g].L; cf.mel2ph = groups[g].mel2ph; cf.p_idx = groups[g].p_idx;
        if (!cf.usable(m) || groups[g].p1_ld > groups[g].T || !m->cond_p2t) fact_all = false;
    }
    std::vector<DenWs> ws(n_groups);
    SideStream* ss = side_for(s);
    // whatever path leaves this function after a fork, the caller's stream is ordered behind the side stream again
    struct JoinGuard {
        SideStream* ss; bool armed;
        ~JoinGuard() { if (armed && ss) (void)branch_join(ss); }
    } guard{ss, false};
    if (ss) { CHK(branch_fork(ss)); guard.armed = true; }
    bool any_aside = false;
    for (int g = 0; g < n_groups; ++g) {
        const cmtts_sample_group& G = groups[g];
        ws[g] = carve_den(c, G.B, G.T, G.ws);
        if (keep[g] < G.B) any_aside = true;
        if (keep[g] == 0) continue;
        CondFactors cf;
        cf.p1 = G.cond_p1; cf.ldp = G.p1_ld; cf.L = G.L; cf.mel2ph = G.mel2ph; cf.p_idx = G.p_idx;
        if (fact_all) {   // the ragged FACT instance gathers the factors itself: no cp tensor — the buffer takes p1 with the channels contiguous
            k_transpose(G.cond_p1, ws[g].cp, keep[g] * NL, C, G.p1_ld, s);
        }
        else if (cf.usable(m)) CHK(cond_factored(m, ws[g], cf, keep[g], G.T, ss ? ss->side : s));
        else CHK(cond_projections(m, ws[g], G.cond_ct, keep[g], G.T, ss ? ss->side : s));   // once for all evaluations, beside the first prologues
        k_scale(G.noise, ws[g].xcur, (long)keep[g] * G.T * M, c.sigma_max, s);         // x_T = randn * sigma_max (karras_diffusion.py:534)
        if (G.active_frames) HIPCHK(hipMemsetAsync(G.mel, 0, (size_t)keep[g] * G.T * M * sizeof(float), s));   // frames beyond the trimmed range: zeros
    }
    if (ss) { guard.armed = false; CHK(branch_join(ss)); }          // the conditioner projections; the side stream then carries the set-aside groups
    if (any_aside) {
        // the small groups: the ordinary sampler (per-layer / split kernels), queued on the side stream so that its launches fill the
        // gaps of the main stream (prologues, launch boundaries) and whatever CUs the persistent grid leaves free
        hipStream_t q = ss ? ss->side : s;
        if (ss) { CHK(branch_fork(ss)); guard.armed = true; }
        const int prev_persist = g_persist;
        g_persist = 0;
        int rc = 0;
        for (int g = 0; g < n_groups && rc == 0; ++g)
            if (keep[g] < groups[g].B) {
                const cmtts_sample_group& G = groups[g];
                const int b0 = keep[g], nb = G.B - b0;
                const long per = (long)G.T * M;
                CondFactors cf;
                cf.p1 = G.cond_p1 ? G.cond_p1 + (long)b0 * NL * C * G.p1_ld : nullptr; cf.ldp = G.p1_ld; cf.L = G.L;
                cf.mel2ph = G.mel2ph ? G.mel2ph + (long)b0 * G.T : nullptr; cf.p_idx = G.p_idx ? G.p_idx + (long)b0 * G.T : nullptr;
                rc = sample_core(m, den_slice(c, ws[g], b0, G.T), G.noise + b0 * per, (long)G.B * per, G.cond_ct + (long)b0 * c.hidden * G.T,
                                 G.speaker_emb ? G.speaker_emb + (long)b0 * c.hidden : nullptr, nb, G.T, n_steps, sigmas, renoise_std,
                                 G.mel + b0 * per, q, &cf);
            }
        g_persist = prev_persist;
        if (rc != 0) return rc;
    }
    const float smin = c.sigma_min, sd2 = c.sigma_data * c.sigma_data;
    const int cap = cap_all;
    PersistArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.vec_stride = (long)NL * C; pa.tmo = g_tmo_host; pa.NL = NL; pa.halo_zeroed = 1;
    pa.tail = 1;
    pa.Wsf = m->skip_f; pa.bs = m->skip_proj.bias; pa.Wpf = m->outp_f; pa.bp = m->out_proj.bias;
    pa.skip_div = (float)sqrt((double)NL); pa.n_mels = M;
    pa.wino = g_persist_wino && m->winograd && m->res[0].w3w;
    for (int g = 0; g < n_groups && pa.wino; ++g)
        if (keep[g] > 0 && !ws[g].pst) pa.wino = 0;
    if (pa.wino && g_persist_wino == 3 && m->winograd == 1 && m->res[0].w3w43) pa.wino = 3;
    for (int l = 0; l < NL; ++l) {
        pa.W3f[l] = pa.wino == 3 ? m->res[l].w3w43 : pa.wino ? m->res[l].w3w : m->res[l].w3f; pa.Wof[l] = m->res[l].wof; pa.b3[l] = m->res[l].b3f; pa.bo[l] = m->res[l].outp.bias;
    }
    pa.n_groups = n_groups;
    if (fact_all) { pa.fact = 1; pa.p2 = m->cond_p2; pa.p2t = m->cond_p2t; pa.ld2 = c.pitch_bins; }
    for (int i = 0; i < n_steps; ++i) {
        // get_scalings_for_boundary_condition in fp32 (karras_diffusion.py:87-102)
        const float sg = sigmas[i];
        const float dm = sg - smin;
        const float c_skip = sd2 / (dm * dm + sd2);
        const float rt = sqrtf(sg * sg + sd2);
        const float c_out = dm * c.sigma_data / rt;
        const float c_in = 1.0f / rt;
        const float t_resc = 250.0f * logf(sg + 1e-44f);
        const bool new_sigma = i == 0 || sigmas[i] != sigmas[i - 1];
        const bool last = i + 1 == n_steps;
        const bool renoise = renoise_std[i] >= 0.0f;
        // an output frame depends on NL frames of input to either side (NL k = 3 layers): evaluation i must be exact on every frame the
        // later evaluations and the caller's `tail_frames` reach, so it is computed NL * (n_steps - i) frames beyond that
        struct Utt { int g, b, act; };
        std::vector<Utt> utts;
        long total = 0;
        for (int g = 0; g < n_groups; ++g) {
            const cmtts_sample_group& G = groups[g];
            const int Bk = keep[g];
            if (Bk == 0) continue;
            const DenWs& w = ws[g];
            if (new_sigma) k_fill_float(w.tbuf, t_resc, Bk, s);
            bool hz = false;
            CHK(denoiser_prologue(m, w, w.xcur, c_in, w.tbuf, G.speaker_emb, Bk, G.T, s, new_sigma, t_resc, &hz));
            if (!hz) HIPCHK(hipMemsetAsync(w.halo, 0, cmtts_persist_halo_bytes(Bk, G.T), s));
            const int tiles = (G.T + 63) / 64;
            PersistGroup& pg = pa.grp[g];
            pg.x0 = w.h; pg.cp = w.cp; pg.cp_bstride = (long)NL * C * G.T;
            pg.dp = c.multi_speaker ? w.dp : w.dproj; pg.d = w.dproj; pg.skip = w.skip; pg.xst = w.pst; pg.halo = w.halo;
            pg.xold = w.xcur; pg.noise = renoise ? G.noise + (long)(1 + i) * G.B * G.T * M : nullptr; pg.out = last ? G.mel : w.xcur;
            pg.B = Bk; pg.T = G.T; pg.tiles = tiles;
            if (fact_all) { pg.p1 = G.cond_p1; pg.p1t = w.cp; pg.mel2ph = (const long long*)G.mel2ph; pg.pidx = (const long long*)G.p_idx; pg.ldp = G.p1_ld; pg.Lph = G.L; }
            for (int b = 0; b < Bk; ++b) {
                const int act = (int)utt_tiles(G, b, i);
                utts.push_back({g, b, act});
                total += act;
            }
        }
        pa.c_out = c_out; pa.c_skip = c_skip; pa.nstd = renoise ? renoise_std[i] : 0.0f;
        // rounds: every workgroup of a launch must be resident, so a shard with more active tiles than CUs runs as balanced rounds of
        // whole utterances (tiles of one utterance exchange their edge columns and must share a launch)
        const int nrounds = (int)((total + cap - 1) / cap);
        const long target = (total + nrounds - 1) / nrounds;
        size_t u = 0;
        while (u < utts.size()) {
            int n = 0;
            while (u < utts.size() && n + utts[u].act <= cap && (n == 0 || n + utts[u].act <= target + 8)) {
                for (int t = 0; t < utts[u].act; ++t)
                    pa.desc[n++] = (unsigned)utts[u].g | ((unsigned)utts[u].b << 3) | ((unsigned)t << 13) | ((unsigned)utts[u].act << 20);
                ++u;
            }
            if (n == 0) return fail(CMTTS_E_UNSUPPORTED, "cmtts_sample_ragged: an utterance has more 64-frame tiles than the GPU has CUs");
            pa.n_wg = n;
            const bool prof = g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
            CHK(persist_admit(s, n, persist_blocks()));
            if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used], s);
            const int rc = cmtts_launch_denoiser_persist_ragged(&pa, (void*)s);
            if (rc != 0) return fail(rc == -2 ? CMTTS_E_UNSUPPORTED : CMTTS_E_HIP, "ragged persistent denoiser launch failed");
            CHK(persist_launched(s, n));
            if (prof) { (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s); g_prof.used += 2; }
        }
    }
    if (any_aside && ss) { guard.armed = false; CHK(branch_join(ss)); }
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------ vocoder
int cmtts_vocoder_create(cmtts_vocoder** out) {
    if (!out) return fail(CMTTS_E_INVALID, "cmtts_vocoder_create: null argument");
    *out = new cmtts_vocoder();
    return 0;
}
int cmtts_vocoder_set_tensor(cmtts_vocoder* v, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    if (!v || v->finalized) return fail(CMTTS_E_INVALID, "cmtts_vocoder_set_tensor: null or finalized");
    return set_tensor(v->host, name, host_data, shape, ndim);
}
int cmtts_vocoder_finalize(cmtts_vocoder* v) {
    if (!v || v->finalized) return fail(CMTTS_E_INVALID, "cmtts_vocoder_finalize: null or finalized");
    Getter g{v->host, ""};
    Allocs& al = v->al;
#define GETV(var, name, ...)                                     \
    const HostTensor* var = g.get(name, {__VA_ARGS__});           \
    if (!var) { al.release(); return fail(CMTTS_E_INVALID, g.missing); }
    GETV(pw, "conv_pre.weight", 512, 80, 7); GETV(pb, "conv_pre.bias", 512);
    CHK(pack_conv(al, *pw, pb, nullptr, &v->conv_pre));
    int ch = 512;
    for (int i = 0; i < 4; ++i) {
        const int co = ch / 2;
        GETV(uw, "ups." + std::to_string(i) + ".weight", ch, co, v->up_kernel[i]);
        GETV(ub, "ups." + std::to_string(i) + ".bias", co);
        {
            std::vector<float> tt;
            CHK(pack_conv_transpose(al, *uw, *ub, v->up_rate[i], &v->ups[i], &tt));
            const int mrows = v->up_rate[i] * co;
            if (!tt.empty() && ch % 16 == 0 && mrows % 32 == 0) {
                CHK(al.upload(to_fragment_iter_order(tt, 2, ch, mrows), &v->ups_f[i]));
                for (int mode = 1; mode <= 2; ++mode) {
                    const std::vector<unsigned short> f16 = to_fragment16(tt, 2, ch, mrows, mode);
                    CHK(al.upload_bytes(f16.data(), f16.size() * 2, &v->ups_f16[i][mode - 1]));
                }
                const std::vector<unsigned short> fs = to_fragment16_split(tt, 2, ch, mrows);
                CHK(al.upload_bytes(fs.data(), fs.size() * 2, &v->ups_f16[i][2]));
            }
        }
        for (int j = 0; j < 3; ++j) {
            const int r = i * 3 + j;
            for (int mi = 0; mi < 3; ++mi) {
                const std::string p = "resblocks." + std::to_string(r);
                GETV(w1, p + ".convs1." + std::to_string(mi) + ".weight", co, co, v->rb_kernel[j]);
                GETV(b1, p + ".convs1." + std::to_string(mi) + ".bias", co);
                GETV(w2, p + ".convs2." + std::to_string(mi) + ".weight", co, co, v->rb_kernel[j]);
                GETV(b2, p + ".convs2." + std::to_string(mi) + ".bias", co);
                std::vector<float> hp;
                CHK(pack_conv(al, *w1, b1, nullptr, &v->c1[r][mi], &hp));
                CHK(al.upload(to_fragment_iter_order(hp, v->rb_kernel[j], co, co), &v->c1f32[r][mi]));
                if (co >= 128 || (co == 64 && v->rb_kernel[j] >= 7)) { const std::vector<float> wf = to_wino_iter_fragments(hp, v->rb_kernel[j], co, co); if (!wf.empty()) CHK(al.upload(wf, &v->c1w32[r][mi])); }
                if (co >= 64) { const std::vector<float> wf = to_wino43_iter_fragments(hp, v->rb_kernel[j], co, co); if (!wf.empty()) CHK(al.upload(wf, &v->c1q32[r][mi])); }     // (C = 64, k = 3: the fused F(4,3) pair, conv_xlq_pair.hip)
                for (int mode = 1; mode <= 2; ++mode) {
                    const std::vector<unsigned short> f16 = to_fragment16(hp, v->rb_kernel[j], co, co, mode);
                    CHK(al.upload_bytes(f16.data(), f16.size() * 2, &v->c1f[r][mi][mode - 1]));
                }
                {
                    const std::vector<unsigned short> fs = to_fragment16_split(hp, v->rb_kernel[j], co, co);
                    CHK(al.upload_bytes(fs.data(), fs.size() * 2, &v->c1f[r][mi][2]));
                }
                CHK(pack_conv(al, *w2, b2, nullptr, &v->c2[r][mi], &hp));
                CHK(al.upload(to_fragment_iter_order(hp, v->rb_kernel[j], co, co), &v->c2f32[r][mi]));
                if (co >= 128 || (co == 64 && v->rb_kernel[j] >= 7)) { const std::vector<float> wf = to_wino_iter_fragments(hp, v->rb_kernel[j], co, co); if (!wf.empty()) CHK(al.upload(wf, &v->c2w32[r][mi])); }
                if (co >= 64) { const std::vector<float> wf = to_wino43_iter_fragments(hp, v->rb_kernel[j], co, co); if (!wf.empty()) CHK(al.upload(wf, &v->c2q32[r][mi])); }
                for (int mode = 1; mode <= 2; ++mode) {
                    const std::vector<unsigned short> f16 = to_fragment16(hp, v->rb_kernel[j], co, co, mode);
                    CHK(al.upload_bytes(f16.data(), f16.size() * 2, &v->c2f[r][mi][mode - 1]));
                }
                {
                    const std::vector<unsigned short> fs = to_fragment16_split(hp, v->rb_kernel[j], co, co);
                    CHK(al.upload_bytes(fs.data(), fs.size() * 2, &v->c2f[r][mi][2]));
                }
            }
        }
        ch = co;
    }
    GETV(qw, "conv_post.weight", 1, ch, 7); GETV(qb, "conv_post.bias", 1);
    CHK(al.upload(qw->data, &v->post_w));
    CHK(al.upload(qb->data, &v->post_b));
    v->post_cin = ch;
#undef GETV
    v->host.clear();
    v->finalized = true;
    return 0;
}
void cmtts_vocoder_destroy(cmtts_vocoder* v) {
    if (!v) return;
    v->al.release();
    delete v;
}
// Row padding of the stage buffers (floats).  Power-of-two row strides were suspected of HBM channel
// camping; padding by 256 B or 4 KB + 128 B changed the vocoder time by < 2 %, so rows stay dense.
static int voc_row_pad() {      // extra floats per row of the stage buffers (experiment switch CMTTS_VOC_PAD: power-of-two row strides put every channel row of a column block on the same HBM channel)
    static const int p = [] { const char* e = getenv("CMTTS_VOC_PAD"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 4096 ? v & ~3 : 0; }();
    return p;
}
// The three ResBlocks of an MRF stage are independent until their sum: they run on three streams (own xt / residual
// buffers, + 4 stage buffers of workspace).  Small batches, whose convs cannot fill the chip (stage 2 has B*T/2
// workgroups), gain most — one 150-frame utterance 4.1 -> 2.7 ms — and 32 x 512 frames still 1 % (tails of one ResBlock's
// launches under the next one's).  Above this many mel frames per call the extra workspace (4 x 32 KB per frame) is not
// spent and the ResBlocks run in line.
constexpr long VOC_PAR_FRAMES = 65536;
size_t cmtts_vocoder_workspace_bytes(const cmtts_vocoder* v, int B, int T) {
    (void)v;
    // five stage buffers of B * max_i(C_i * T_i) floats: C_i*T_i = T * {512, 2048, 8192, 8192, 8192}
    // (rows are padded by VOC_ROW_PAD floats; at most 512 rows per utterance)
    // + four more (xt / running residual of the second and third ResBlock) when the batch is small enough for the three
    // ResBlocks of a stage to run side by side (VOC_PAR_FRAMES)
    const int nb = (long)B * T <= VOC_PAR_FRAMES ? 9 : 5;
    return (size_t)nb * (((size_t)B * T * 8192 + (size_t)B * 512 * voc_row_pad()) * sizeof(float) + 256) + 256;
}
int cmtts_vocoder_forward(cmtts_vocoder* v, const float* mel_ct, int B, int T, float* wav, void* ws, size_t ws_bytes,
                          void* stream) {
    if (!v || !v->finalized) return fail(CMTTS_E_INVALID, "vocoder not finalized");
    if (!mel_ct || !wav || !ws || B <= 0 || T <= 0) return fail(CMTTS_E_INVALID, "cmtts_vocoder_forward: bad argument");
    if (ws_bytes < cmtts_vocoder_workspace_bytes(v, B, T)) return fail(CMTTS_E_WORKSPACE, "vocoder workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver cv(ws);
    const int P = voc_row_pad();
    const size_t nbuf = (size_t)B * T * 8192 + (size_t)B * 512 * P;
    float* bufA = cv.take<float>(nbuf);   // stage input
    float* bufU = cv.take<float>(nbuf);   // upsampled
    float* bufT = cv.take<float>(nbuf);   // xt
    float* bufR = cv.take<float>(nbuf);   // running residual inside a ResBlock
    float* bufS = cv.take<float>(nbuf);   // MRF sum
    SideStream* ss = (long)B * T <= VOC_PAR_FRAMES ? side_for(s) : nullptr;
    if (ss && !side2_ready(ss)) ss = nullptr;
    float *bufTj[3] = {bufT, bufT, bufT}, *bufRj[3] = {bufR, bufR, bufR};
    hipStream_t sj[3] = {s, s, s};
    if (ss) {   // own xt / residual buffers and streams for the second and third ResBlock
        for (int j = 1; j < 3; ++j) { bufTj[j] = cv.take<float>(nbuf); bufRj[j] = cv.take<float>(nbuf); }
        sj[1] = ss->side; sj[2] = ss->side2;
    }
    {   // conv_pre (hifigan/models.py:150)
        ConvArgs a = conv_args(v->conv_pre, mel_ct, T, T, (long)80 * T, bufA, T + P, (long)512 * (T + P), T);
        CHK(launch(a, EPI_PLAIN, B, s));
    }
    int Ti = T, ch = 512;
    for (int i = 0; i < 4; ++i) {
        const int st = v->up_rate[i], K = v->up_kernel[i], pd = (K - st) / 2, co = ch / 2, To = Ti * st;
        {   // x = ups[i](leaky_relu(x, 0.1)) as `st` polyphase sub-convolutions (hifigan/models.py:152-153)
            const PackedConv& U = v->ups[i];
            int rt = -2;
            if (v->ups16 && v->precision >= 1 && v->precision <= 3 && v->ups_f16[i][v->precision - 1] && K == 2 * st)
                // 16-bit modes: the upsamplers' operands are 16-bit too (since round 2; the oracle's operands16 modes follow); fp16x3 (round 3):
                // (hi, lo) operand pairs like the ResBlock convs
                rt = cmtts_launch_convT16(bufA, bufU, v->ups_f16[i][v->precision - 1], U.bias, (long)ch * (Ti + P), (long)co * (To + P), B, ch,
                                          co, Ti, To, Ti + P, To + P, st, i > 0 ? 3.0f : 1.0f, 0.1f, v->precision, (void*)s);
            if (rt == -3) return fail(CMTTS_E_HIP, "convT16 launch failed");
            if (rt != 0 && g_voc_upsT && v->ups_f[i] && K == 2 * st)      // all phases in one X-resident launch (same bits)
                rt = cmtts_launch_convT(bufA, bufU, v->ups_f[i], U.bias, (long)ch * (Ti + P), (long)co * (To + P), B, ch, co, Ti, To,
                                        Ti + P, To + P, st, i > 0 ? 3.0f : 1.0f, 0.1f, (void*)s);
            if (rt == -3) return fail(CMTTS_E_HIP, "convT launch failed");
            if (rt != 0) {
            ConvArgs a = conv_args(U, bufA, Ti, Ti + P, (long)ch * (Ti + P), bufU, To + P, (long)co * (To + P), Ti + 1);
            a.dil = -1; a.pad = 0;
            a.zdiv = st; a.a_zs0 = 0; a.a_zs1 = U.phase_stride; a.x_zs0 = (long)ch * (Ti + P); a.x_zs1 = 0;
            a.pre_slope = 0.1f;
            a.pre_div = i > 0 ? 3.0f : 1.0f;     // x = xs / num_kernels of the previous stage (:160)
            ConvOut& o = a.out[0];
            o.Tout = To; o.ostride = st; o.ooff_base = -pd; o.ooff_mul = 1; o.y_zs0 = (long)co * (To + P); o.y_zs1 = 0;
            CHK(launch(a, EPI_PLAIN, B * st, s));
            }
        }
        const int ld = To + P;               // row stride: not a power of two (HBM channel spread)
        const long cs = (long)co * ld;
        if (ss) {   // the three chains see the upsampled input
            HIPCHK(hipEventRecord(ss->fork, s));
            HIPCHK(hipStreamWaitEvent(ss->side, ss->fork, 0));
            HIPCHK(hipStreamWaitEvent(ss->side2, ss->fork, 0));
        }
        for (int j = 0; j < 3; ++j) {          // MRF: three ResBlocks on the same input (:154-159)
            const int r = i * 3 + j, rk = v->rb_kernel[j];
            hipStream_t q = sj[j];
            float *bT = bufTj[j], *bR = bufRj[j];
            const float* xr = bufU;
            static const char* rb_only = getenv("CMTTS_RB16_ONLY");      // debugging aid: "C,k" restricts the fused ResBlock to one shape
            int rbC = 0, rbK = 0;
            const bool rb_sel = !rb_only || (sscanf(rb_only, "%d,%d", &rbC, &rbK) == 2 && rbC == co && rbK == rk);
            // measured per ResBlock (bf16, 32 x 512 frames): C = 32: 0.51 / 0.83 / 1.10 ms (k = 3 / 7 / 11) against 1.21 / 1.30 / 1.40 for three
            // pair launches; C = 64 (8 waves, one 118-KB workgroup per CU): 0.76 / 1.31 / 1.91 against 1.15 / 1.40 / 1.66 — at k = 11 the halo
            // recompute (+45 % MFMAs) costs more than the four tensor passes saved (g_voc_rb16 == 2: the fused form always)
            const bool rb_pays = co == 32 || rk <= 7 || g_voc_rb16 == 2;
            if (g_voc_rb16 && rb_pays && rb_sel && (v->precision == 1 || v->precision == 2) && co <= 64 && v->c1f[r][0][v->precision - 1]) {
                // narrow stages, 16-bit operands: the WHOLE ResBlock (three pairs) in one launch — x in, MRF sum out: 2 tensor
                // passes instead of 6 (resblock16_kernel; bitwise equal to three pair launches)
                const void *w1[3], *w2[3];
                const float *bb1[3], *bb2[3];
                for (int mi = 0; mi < 3; ++mi) {
                    w1[mi] = v->c1f[r][mi][v->precision - 1]; w2[mi] = v->c2f[r][mi][v->precision - 1];
                    bb1[mi] = v->c1[r][mi].bias; bb2[mi] = v->c2[r][mi].bias;
                }
                if (ss && j > 0) HIPCHK(hipStreamWaitEvent(q, j == 1 ? ss->done0 : ss->done1, 0));     // MRF sum in ResBlock order
                const int rrc = cmtts_launch_resblock16(xr, bufS, w1, w2, bb1, bb2, cs, B, co, To, ld, rk, j > 0, 0.1f, v->precision, (void*)q);
                if (rrc == -3) return fail(CMTTS_E_HIP, "resblock16 launch failed");
                if (rrc == 0) {
                    if (ss && j < 2) HIPCHK(hipEventRecord(j == 0 ? ss->done0 : ss->done1, q));
                    continue;
                }
            }
            // narrow stages: conv1 -> LeakyReLU -> conv2 -> + x of a pair in ONE launch, xt never leaves the CU
            // (resblock_pair.hip; the pair's output must not alias its input, so the chain ping-pongs bR / bT)
            // 16-bit operands: with the weight ring issued by hand (resblock_pair16.hip: the compiler had sunk every fragment load
            // next to its use) the pair kernel wins for every (C, k): 0.37-0.55 ms per pair against 0.60-0.64 for two launches
            // (profiles/r02_vocoder_bf16.md)
            const bool pair16_pays = true;
            // 16-bit, C = 128 (round 2): the pair as one 8-wave workgroup with both images in LDS (151 KB) — conv_xl16 otherwise
            // (measured, bf16: k = 3 / 7 / 11: 471 / 754 / 967 us per pair against 527 / 700 / 903 for the two launches: k = 3 only)
            const bool pair128 = g_voc_pair128 && co == 128 && rk == 3 && (v->precision == 1 || v->precision == 2);
            // 16-bit, C = 128 (round 3): the pair in ONE launch with a single in-place image (81 KB: two workgroups per CU, 2 x 4 tiles per wave)
            const bool pairw = g_voc_pairw && co == 128 && (v->precision == 1 || v->precision == 2);
            // fp16x3 (round 3): the pair X-resident with (hi, lo) images, C <= 128 (resblock_pair16x3.hip; same bits as the two conv16 launches)
            const bool pair3 = g_voc_pair3 && v->precision == 3 && co <= 128;
            bool pair_ok = g_voc_pair && (co <= 64 || pair128 || pairw || pair3) && (v->precision != 3 || pair3) &&
                                 (v->precision ? (pair16_pays && v->c1f[r][0][v->precision - 1] != nullptr) : v->c1f32[r][0] != nullptr);
            // fp32, C = 64, k >= 7, chip-filling launches (round 4): the pair as two Winograd launches (conv_xlw_kernel<64>: one wave per workgroup with both
            // m-tiles, eight workgroups per CU) instead of the fused pair kernel — 10 / 15 products per output pair instead of 14 / 22 outweigh xt's trip through HBM (k = 11: 2 x 1225 against 3217 us; k = 7: -0.3 ms per batch)
            const bool xw64 = g_voc_wino64 && g_voc_wino && v->winograd && !v->precision && co == 64 && rk >= g_voc_wino64_k && v->c1w32[r][0] && v->c2w32[r][0] &&
                              (g_voc_wino == 2 || (long)((To + 63) / 64) * B >= 1024);
            if (xw64) pair_ok = false;
            // round 6: k = 3 pairs at C = 64 / 128 with both convs in the F(4,3) form and xt kept on the CU (conv_xlq_pair.hip): at C = 128 the two conv_xlq
            // launches below without xt's trip through HBM and the residual's second read (five tensor passes -> two; the same products on quads one frame apart: fp32
            // Winograd rounding between the two); at C = 64 half the MFMAs of the direct pair kernel.  64.4 -> 62.9 ms per 32 x 512-frame batch
            bool qpair = g_voc_qpair && g_voc_wino && g_voc_wino43 && v->winograd && !v->precision && rk == 3 && (co == 64 || co == 128) &&
                         (g_voc_qpair == 2 || g_voc_wino == 2 || (long)((To + 63) / 64) * B >= 1024);
            for (int mi = 0; mi < 3 && qpair; ++mi) qpair = v->c1q32[r][mi] && v->c2q32[r][mi];
            for (int mi = 0; mi < 3 && qpair; ++mi) {
                const bool lastm = mi == 2;
                if (ss && lastm && j > 0) HIPCHK(hipStreamWaitEvent(q, j == 1 ? ss->done0 : ss->done1, 0));
                PairArgs pa;
                memset(&pa, 0, sizeof(pa));
                pa.x = xr; pa.y = lastm ? bufS : (mi == 0 ? bR : bT);
                pa.b1 = v->c1[r][mi].bias; pa.b2 = v->c2[r][mi].bias;
                pa.w1f = v->c1q32[r][mi]; pa.w2f = v->c2q32[r][mi];
                pa.bstride = cs; pa.B = B; pa.C = co; pa.T = To; pa.ld = ld; pa.k = rk; pa.dil = v->rb_dil[mi];
                pa.accum = lastm && j > 0; pa.slope = 0.1f;
                if (cmtts_launch_conv_xlq_pair(&pa, (void*)q) != 0) return fail(CMTTS_E_HIP, "conv_xlq_pair launch failed");
                if (ss && lastm && j < 2) HIPCHK(hipEventRecord(j == 0 ? ss->done0 : ss->done1, q));
                xr = pa.y;
            }
            if (qpair) continue;
            for (int mi = 0; mi < 3 && pair_ok; ++mi) {
                const bool lastm = mi == 2;
                if (ss && lastm && j > 0) HIPCHK(hipStreamWaitEvent(q, j == 1 ? ss->done0 : ss->done1, 0));
                PairArgs pa;
                memset(&pa, 0, sizeof(pa));
                pa.x = xr; pa.y = lastm ? bufS : (mi == 0 ? bR : bT);
                pa.b1 = v->c1[r][mi].bias; pa.b2 = v->c2[r][mi].bias;
                if (v->precision) { pa.w1f = v->c1f[r][mi][v->precision - 1]; pa.w2f = v->c2f[r][mi][v->precision - 1]; }
                else { pa.w1f = v->c1f32[r][mi]; pa.w2f = v->c2f32[r][mi]; }
                pa.bstride = cs; pa.B = B; pa.C = co; pa.T = To; pa.ld = ld; pa.k = rk; pa.dil = v->rb_dil[mi];
                pa.accum = lastm && j > 0; pa.slope = 0.1f;
                int prc;
                if (!v->precision) prc = cmtts_launch_resblock_pair(&pa, (void*)q);
                else if (pair3) prc = cmtts_launch_resblock_pair16x3(&pa, (void*)q);
                else if (pairw) prc = cmtts_launch_resblock_pairw16(&pa, v->precision, (void*)q);
                else prc = cmtts_launch_resblock_pair16(&pa, v->precision, (void*)q);
                if (prc == -2 && mi == 0) {      // this (C, k, dilation) is not covered by the pair kernels: the per-conv path below
                    pair_ok = false;            // (nothing has been launched for this ResBlock yet)
                    break;
                }
                if (prc != 0) return fail(CMTTS_E_HIP, "resblock_pair launch failed");
                if (ss && lastm && j < 2) HIPCHK(hipEventRecord(j == 0 ? ss->done0 : ss->done1, q));
                xr = pa.y;
            }
            for (int mi = 0; mi < 3 && !pair_ok; ++mi) {   // ResBlock.forward (:96-103)
                const int dil = v->rb_dil[mi];
                const bool lastm = mi == 2;
                if (g_voc_xl && !v->precision && (co >= 128 || xw64) && v->c1f32[r][mi]) {   // wide stages: X-resident single convs
                    ConvXlArgs xa;
                    memset(&xa, 0, sizeof(xa));
                    // round 4: the Winograd form of both convs (conv_xlw_kernel: 4 / 10 / 15 products per output pair instead of 6 / 14 / 22)
                    const bool xw = g_voc_wino && v->winograd && v->c1w32[r][mi] && v->c2w32[r][mi];
                    xa.x = xr; xa.y = bT; xa.wf = xw ? v->c1w32[r][mi] : v->c1f32[r][mi]; xa.bias = v->c1[r][mi].bias;
                    xa.bstride = cs; xa.B = B; xa.C = co; xa.T = To; xa.ld = ld; xa.k = rk; xa.dil = dil; xa.slope = 0.1f; xa.wino_force = g_voc_wino == 2;
                    // round 5: dilation-1 convs (every conv2, conv1 of the first pair) in the F(4,3) form (conv_xlq_kernel: 6 / 16 / 24 products per quad of outputs where
                    // the F(2,3) tap groups take 8 / 20 / 30); -2 = launch too small or shape not covered: the F(2,3) form, then the direct one
                    int rc1 = -2;
                    // (dilation 3 in that form everywhere, dilation 5 only at C = 256 or k = 3: the five-class tiles of C = 128 / 64 (one workgroup fewer per CU, 15 of
                    //  16 quad lanes, strided stores) are slower than the F(2,3) pair tiles at k = 7 / 11 — 3.63 vs 2.44 ms at C = 128, k = 11; voc_wino43 = 3 forces them for tests)
                    if (xw && g_voc_wino43 && (dil == 1 || (g_voc_wino43 == 1 && (co == 256 || dil == 3 || rk == 3)) || g_voc_wino43 == 3) && v->c1q32[r][mi]) { xa.wf = v->c1q32[r][mi]; rc1 = cmtts_launch_conv_xlq(&xa, (void*)q); if (rc1 == -2) xa.wf = v->c1w32[r][mi]; }
                    if (rc1 == -2 && xw) rc1 = cmtts_launch_conv_xlw(&xa, (void*)q);
                    const bool xw1 = rc1 == 0;
                    if (rc1 == -2) { xa.wf = v->c1f32[r][mi]; rc1 = cmtts_launch_conv_xl(&xa, (void*)q); }
                    if (rc1 == -3) return fail(CMTTS_E_HIP, "conv_xl launch failed");
                    if (rc1 == 0) {
                        if (ss && lastm && j > 0) HIPCHK(hipStreamWaitEvent(q, j == 1 ? ss->done0 : ss->done1, 0));
                        // the residual operand is read at the positions this launch writes when y == res (in place: safe,
                        // every output element reads only its own residual); the INPUT must not alias the output
                        xa.x = bT; xa.y = lastm ? bufS : bR; xa.wf = xw1 ? v->c2w32[r][mi] : v->c2f32[r][mi]; xa.bias = v->c2[r][mi].bias;
                        xa.res = xr; xa.dil = 1; xa.accum = lastm && j > 0;
                        int rc2 = -2;
                        if (xw1 && g_voc_wino43 && v->c2q32[r][mi]) { xa.wf = v->c2q32[r][mi]; rc2 = cmtts_launch_conv_xlq(&xa, (void*)q); if (rc2 == -2) xa.wf = v->c2w32[r][mi]; }
                        if (rc2 == -2) rc2 = xw1 ? cmtts_launch_conv_xlw(&xa, (void*)q) : cmtts_launch_conv_xl(&xa, (void*)q);
                        if (rc2 != 0) return fail(CMTTS_E_HIP, "conv_xl launch failed");
                        if (ss && lastm && j < 2) HIPCHK(hipEventRecord(j == 0 ? ss->done0 : ss->done1, q));
                        xr = bR;
                        continue;
                    }
                }
                if (g_voc_xl16 && (v->precision == 1 || v->precision == 2) && co >= 128 && v->c1f[r][mi][v->precision - 1]) {
                    // wide stages, 16-bit operands: X-resident single convs (conv_xl16_kernel); xt crosses HBM in 16 bits
                    ConvXlArgs xa;
                    memset(&xa, 0, sizeof(xa));
                    xa.x = xr; xa.y = bT; xa.wf = (const float*)v->c1f[r][mi][v->precision - 1]; xa.bias = v->c1[r][mi].bias;
                    xa.bstride = cs; xa.B = B; xa.C = co; xa.T = To; xa.ld = ld; xa.k = rk; xa.dil = dil; xa.slope = 0.1f;
                    const int rc1 = cmtts_launch_conv_xl16(&xa, v->precision, 1, (void*)q);
                    if (rc1 == -3) return fail(CMTTS_E_HIP, "conv_xl16 launch failed");
                    if (rc1 == 0) {
                        if (ss && lastm && j > 0) HIPCHK(hipStreamWaitEvent(q, j == 1 ? ss->done0 : ss->done1, 0));
                        xa.x = bT; xa.y = lastm ? bufS : bR; xa.wf = (const float*)v->c2f[r][mi][v->precision - 1];
                        xa.bias = v->c2[r][mi].bias; xa.res = xr; xa.dil = 1; xa.accum = lastm && j > 0;
                        if (cmtts_launch_conv_xl16(&xa, v->precision, 2, (void*)q) != 0) return fail(CMTTS_E_HIP, "conv_xl16 launch failed");
                        if (ss && lastm && j < 2) HIPCHK(hipEventRecord(j == 0 ? ss->done0 : ss->done1, q));
                        xr = bR;
                        continue;
                    }
                }
                if (g_voc_pair3 && v->precision == 3 && co == 256 && v->c1f[r][mi][2]) {
                    // fp16x3, C = 256: X-resident single convs with (hi, lo) images (conv_xl16x3_kernel); xt crosses HBM in fp32
                    ConvXlArgs xa;
                    memset(&xa, 0, sizeof(xa));
                    xa.x = xr; xa.y = bT; xa.wf = (const float*)v->c1f[r][mi][2]; xa.bias = v->c1[r][mi].bias;
                    xa.bstride = cs; xa.B = B; xa.C = co; xa.T = To; xa.ld = ld; xa.k = rk; xa.dil = dil; xa.slope = 0.1f;
                    const int rc1 = cmtts_launch_conv_xl16x3(&xa, (void*)q);
                    if (rc1 == -3) return fail(CMTTS_E_HIP, "conv_xl16x3 launch failed");
                    if (rc1 == 0) {
                        if (ss && lastm && j > 0) HIPCHK(hipStreamWaitEvent(q, j == 1 ? ss->done0 : ss->done1, 0));
                        xa.x = bT; xa.y = lastm ? bufS : bR; xa.wf = (const float*)v->c2f[r][mi][2];
                        xa.bias = v->c2[r][mi].bias; xa.res = xr; xa.dil = 1; xa.accum = lastm && j > 0;
                        if (cmtts_launch_conv_xl16x3(&xa, (void*)q) != 0) return fail(CMTTS_E_HIP, "conv_xl16x3 launch failed");
                        if (ss && lastm && j < 2) HIPCHK(hipEventRecord(j == 0 ? ss->done0 : ss->done1, q));
                        xr = bR;
                        continue;
                    }
                }
                ConvArgs a = conv_args(v->c1[r][mi], xr, To, ld, cs, bT, ld, cs, To);
                a.dil = dil; a.pad = (rk * dil - dil) / 2; a.pre_slope = 0.1f;
                if (v->precision == 3) {                  // fp16x3: fp32 xt in HBM, operands split into hi + lo fp16 while staged
                    if (cmtts_launch_conv16(&a, v->c1f[r][mi][2], 3, B, (void*)q) != 0)
                        return fail(CMTTS_E_HIP, "conv16 launch failed");
                } else if (v->precision) {
                    a.y16 = 1; a.y16_slope = 0.1f;        // xt crosses HBM as convert(leaky_relu(xt)) in 16 bits
                    if (cmtts_launch_conv16(&a, v->c1f[r][mi][v->precision - 1], v->precision, B, (void*)q) != 0)
                        return fail(CMTTS_E_HIP, "conv16 launch failed");
                } else {
                    CHK(launch(a, EPI_PLAIN, B, q));
                }
                // the MRF sum accumulates in ResBlock order (bit-identical to the in-line order): the last conv of
                // chain j waits for the last conv of chain j-1
                if (ss && lastm && j > 0) HIPCHK(hipStreamWaitEvent(q, j == 1 ? ss->done0 : ss->done1, 0));
                ConvArgs b = conv_args(v->c2[r][mi], bT, To, ld, cs, lastm ? bufS : bR, ld, cs, To);
                b.pre_slope = 0.1f;
                b.out[0].res = xr; b.out[0].r_zs0 = cs; b.out[0].ldr = ld;
                b.out[0].accum = lastm && j > 0;
                if (v->precision == 3) {
                    if (cmtts_launch_conv16(&b, v->c2f[r][mi][2], 3, B, (void*)q) != 0)
                        return fail(CMTTS_E_HIP, "conv16 launch failed");
                } else if (v->precision) {
                    b.x16 = 1;
                    if (cmtts_launch_conv16(&b, v->c2f[r][mi][v->precision - 1], v->precision, B, (void*)q) != 0)
                        return fail(CMTTS_E_HIP, "conv16 launch failed");
                } else {
                    CHK(launch(b, EPI_PLAIN, B, q));
                }
                if (ss && lastm && j < 2) HIPCHK(hipEventRecord(j == 0 ? ss->done0 : ss->done1, q));
                xr = bR;
            }
        }
        if (ss) {   // the next stage (and conv_post) read the sum: chain 2's last conv is the last writer; chain 1 is
                    // ordered before it, but its stream must also be idle before its buffers are reused
            HIPCHK(hipEventRecord(ss->join, ss->side));
            HIPCHK(hipEventRecord(ss->join2, ss->side2));
            HIPCHK(hipStreamWaitEvent(s, ss->join, 0));
            HIPCHK(hipStreamWaitEvent(s, ss->join2, 0));
        }
        float* t = bufA; bufA = bufS; bufS = t;
        Ti = To; ch = co;
    }
    // x = leaky_relu(xs / 3) [slope 0.01] -> conv_post -> tanh (:161-163)
    k_conv_post(bufA, v->post_w, v->post_b, 3.0f, 0.01f, wav, B, ch, Ti, Ti + P, v->post_k, s);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- options.  Three tiers:
//   cmtts_set_option            process-wide SCHEDULING knobs of the public ABI: they never change a result bit
//   cmtts_model_set_option /    per-handle NUMERICS choices (another fp32 summation order / another operand precision):
//   cmtts_vocoder_set_option    properties of a model, not of the process
//   cmtts_internal_set          A/B switches between a fused kernel and the path it replaces (bitwise equal, used by tests/ and
//                               tools/ for cross-checks and measurements); declared in csrc/internal_hooks.h, NOT part of the ABI
struct Knob { const char* name; int* var; int lo, hi; };
static int knob_set(const Knob* tab, size_t n, const char* name, int value, bool* found) {
    for (size_t i = 0; i < n; ++i)
        if (!strcmp(name, tab[i].name)) {
            *found = true;
            const int prev = *tab[i].var;
            if (value >= tab[i].lo && value <= tab[i].hi) *tab[i].var = value;
            return prev;
        }
    *found = false;
    return 0;
}

int cmtts_set_option(const char* name, int value) {
    if (!name) return fail(CMTTS_E_INVALID, "cmtts_set_option: null name");
    if (!strcmp(name, "cooperative_launch")) {   // persistent denoiser through hipLaunchCooperativeKernel: 0 never, 1 always, 2 = automatic
        // (default): with a communicator / process group in the process ("process_group") the first launch of every grid shape is
        // cooperative (the runtime validates co-residency), later ones plain (a cooperative launch drains every queue: +24 % per step)
        const int prev = cmtts_persist_set_cooperative(value == 2 ? -1 : (value == 0 || value == 1) ? value : -2);
        return prev < 0 ? 2 : prev;
    }
    if (!strcmp(name, "process_group")) {        // the host tells the library that a process group exists (torch.distributed initialised)
        return cmtts_persist_note_process_group(value);
    }
    static const Knob tab[] = {
        {"branch_streams", &g_branch_streams, 0, 1},     // independent branches of a call on library-owned side streams
        {"resblock_split", &g_split_resblock, 0, 2},     // fp32 residual block as two launches over 4x the CUs: 0 never, 1 small batches, 2 always
        {"step_cache", &g_step_cache, 0, 1},             // cmtts_sample: keep the timestep-only step-embedding rows on the device
    };
    bool found;
    const int prev = knob_set(tab, sizeof(tab) / sizeof(tab[0]), name, value, &found);
    if (found) return prev;
    return fail(CMTTS_E_INVALID, "cmtts_set_option: unknown option");
}

int cmtts_model_set_option(cmtts_model* m, const char* name, int value) {
    if (!m || !name) return fail(CMTTS_E_INVALID, "cmtts_model_set_option: null argument");
    const Knob tab[] = {
        {"ffn2_split", &m->ffn2_split, 0, 1},            // FFN linear of the FFT blocks as 8 K-segment partial GEMMs + one reduction (1) or one launch (0)
        {"text16", &m->text16, 0, 1},                    // bf16 / fp16 models: 16-bit operands in the FFT blocks' FFN contractions as well (default 0)
        {"winograd", &m->winograd, 0, 2},                // fp32 persistent denoiser stack: the gated k = 3 conv as Winograd F(4,3) (default 1), F(2,3) (2) or in the direct form (0: bitwise the per-layer kernels)
    };
    bool found;
    const int prev = knob_set(tab, sizeof(tab) / sizeof(tab[0]), name, value, &found);
    if (found) return prev;
    return fail(CMTTS_E_INVALID, "cmtts_model_set_option: unknown option");
}

int cmtts_vocoder_set_option(cmtts_vocoder* v, const char* name, int value) {
    if (!v || !name) return fail(CMTTS_E_INVALID, "cmtts_vocoder_set_option: null argument");
    const Knob tab[] = {
        {"ups16", &v->ups16, 0, 1},                      // 16-bit modes: 16-bit operands in the upsamplers too (1) or fp32 upsamplers (0)
        {"winograd", &v->winograd, 0, 1},                // fp32 generator: the ResBlock convs of the C >= 128 stages in their Winograd form (default 1; 0 = the direct form)
    };
    bool found;
    const int prev = knob_set(tab, sizeof(tab) / sizeof(tab[0]), name, value, &found);
    if (found) return prev;
    return fail(CMTTS_E_INVALID, "cmtts_vocoder_set_option: unknown option");
}

// csrc/internal_hooks.h — fused kernel vs the path it replaces; every pair is bitwise equal (tests/test_gpu_parity.py)
int cmtts_internal_set(const char* name, int value) {
    if (!name) return fail(CMTTS_E_INVALID, "cmtts_internal_set: null name");
    static const Knob tab[] = {
        {"qkv_nt", &g_qkv_nt, 0, 3},
        {"cwt_in_phoneme", &g_cwt_in_phoneme, 0, 1},   // Linear(256 -> 128) of the pitch predictor before (1) or after (0) the length regulator
        {"pred_xres", &g_pred_xres, 0, 1},         // phoneme-level predictor convs on conv_xres with the LayerNorm prologue
        {"xres_small", &g_xres_small, 0, 1},       // FFT blocks of small batches on conv_xres with 32-column tiles
        {"cond_inkernel", &g_cond_inkernel, 0, 1}, // fp32 persistent denoiser gathers the conditioner factors itself (same bits as expanding them into cp first)
        {"cond_factored", &g_cond_factored, 0, 1}, // fp32 models: conditioner projections expanded from their phoneme-level / pitch-table factors when the caller hands them over (NOT bitwise the dense GEMM: W a + W b against W (a + b))
        {"cond_gemm16", &g_cond_gemm16, 0, 1},     // 16-bit models: conditioner GEMM with 16-bit operands (NOT bitwise: another operand precision)
        {"cond_gemm", &g_cond_gemm, 0, 2},         // stacked conditioner GEMM on cond_gemm.hip: 0 never, 1 when it pays, 2 whenever supported
        {"persist_tail", &g_persist_tail, 0, 1},   // skip head + post-scaling inside the persistent launch
        {"persist_wino", &g_persist_wino, 0, 3},   // fp32 persistent denoiser's k = 3 conv: 0 direct, 1 (and 2) Winograd F(2,3), 3 F(4,3) (NOT bitwise the direct form)
        {"inproj_fused", &g_inproj_fused, 0, 1},   // denoiser input as one launch
        {"ffn_xres", &g_ffn_xres, 0, 1},           // k = 9 FFN conv on conv_xres.hip
        {"ffn_wino", &g_ffn_wino, 0, 2},           // FFN conv as Winograd tap groups in the fused launch (fp32; NOT bitwise the direct form): 1 = F(2,3) pairs (default), 2 = F(4,3) quads
        {"ffn_fused", &g_ffn_fused, 0, 1},         // FFN linear's partial products inside the FFN conv's launch
        {"text_xres", &g_text_xres, 0, 15},        // bit mask: 1 LN1 + in-projection, 2 out-projection, 4 LN2 + FFN conv on conv_xres.hip
        {"attn_fused", &g_attn_fused, 0, 1},       // fused attention kernel vs three launches
        {"pred_xl", &g_pred_xl, 0, 1},             // frame-level predictor convs on conv_xl
        {"pred_head", &g_pred_head, 0, 1},         // LayerNorm + linear head in one launch
        {"voc_pair", &g_voc_pair, 0, 2},           // HiFi-GAN ResBlock pairs (C <= 64) as one launch
        {"voc_pair3", &g_voc_pair3, 0, 1},         // fp16x3 pairs: one X-resident launch (resblock_pair16x3.hip)
        {"voc_pairw", &g_voc_pairw, 0, 1},         // 16-bit C = 128 pairs: one launch, one in-place LDS image (resblock_pairw16.hip)
        {"voc_pair128", &g_voc_pair128, 0, 1},     // 16-bit C = 128, k = 3 pair kernel (two images, one workgroup per CU; only when voc_pairw = 0)
        {"voc_rb16", &g_voc_rb16, 0, 2},           // 16-bit whole-ResBlock kernel: 0 never, 1 where it pays, 2 always
        {"voc_xl", &g_voc_xl, 0, 1},               // fp32 wide-stage convs on conv_xl
        {"voc_wino64_k", &g_voc_wino64_k, 3, 99},
        {"voc_wino43", &g_voc_wino43, 0, 3},       // fp32 dilation-1 convs of the Winograd path as F(4,3) (conv_xlq_kernel; NOT bitwise F(2,3) or direct)
        {"voc_qpair", &g_voc_qpair, 0, 2},         // fp32 k = 3 pairs at C = 64 / 128 of the Winograd path as ONE F(4,3) launch (conv_xlq_pair.hip; NOT bitwise the forms it replaces)
        {"voc_wino64", &g_voc_wino64, 0, 1},       // fp32 C = 64 stage, k >= voc_wino64_k: two conv_xlw launches per pair (with voc_wino) instead of the pair kernel
        {"voc_wino", &g_voc_wino, 0, 2},           // fp32 wide-stage convs in their Winograd form (NOT bitwise: the A/B twin of the vocoder option "winograd")
        {"voc_xl16", &g_voc_xl16, 0, 1},           // 16-bit wide-stage convs on conv_xl16
        {"voc_upsT", &g_voc_upsT, 0, 1},           // upsamplers on convT_xl
        {"post_v4", &g_post_v4, 0, 1},             // conv_post with 16-byte loads
        {"pred_wino", &g_pred_wino, 0, 1},         // pitch predictor's k = 5 convs as F(4,3) tap groups (NOT bitwise the direct form)
        {"energy_head", &g_energy_head, 0, 1},     // energy bucketize + embedding add inside the energy predictor's head launch (same bits)
        {"stats_mlp", &g_stats_mlp, 0, 1},         // cwt_stats_layers as one launch (same bits)
        {"text_xt16", &g_conv_xt16, 0, 1},         // text16 convs with K = 256 on the X-resident 16-bit kernel (conv_xt16.hip) instead of the chunked one
    };
    if (!strcmp(name, "voc_xl_split")) return cmtts_xl_set_split(value);
    if (!strcmp(name, "attn_qb")) return cmtts_attention_set_qb(value);       // attention.hip: queries split over workgroups (round 6; same bits)
    if (!strcmp(name, "xres_nt")) return cmtts_xres_set_nt(value);            // conv_xres tile width for launches that do not choose one (measurements)      // conv_xl: m-tiles over several workgroups for launches of a few column tiles
    bool found;
    const int prev = knob_set(tab, sizeof(tab) / sizeof(tab[0]), name, value, &found);
    if (found) return prev;
    return fail(CMTTS_E_INVALID, "cmtts_internal_set: unknown switch");
}

int cmtts_internal_cond_projections(cmtts_model* m, const float* cond_ct, int B, int T, float* cp, void* stream) {
    if (!m || !cond_ct || !cp || B <= 0 || T <= 0) return fail(CMTTS_E_INVALID, "cmtts_internal_cond_projections: bad argument");
    DenWs w;
    memset(&w, 0, sizeof(w));
    w.cp = cp;
    return cond_projections(m, w, cond_ct, B, T, (hipStream_t)stream);
}

// Test hook: the conditioner projections expanded from their factors (cond_factored) into cp [B][NL*C][T]
int cmtts_internal_cond_factored(cmtts_model* m, const float* p1, int p1_ld, int L, const int64_t* mel2ph, const int64_t* p_idx, int B, int T,
                                 float* cp, void* stream) {
    if (!m || !m->finalized || !p1 || !mel2ph || !p_idx || !cp) return fail(CMTTS_E_INVALID, "cmtts_internal_cond_factored: bad argument");
    if (!m->cond_p2) return fail(CMTTS_E_UNSUPPORTED, "no pitch-table factor for this model");
    DenWs w;
    memset(&w, 0, sizeof(w));
    w.cp = cp;
    CondFactors cf;
    cf.p1 = p1; cf.ldp = p1_ld; cf.L = L; cf.mel2ph = mel2ph; cf.p_idx = p_idx;
    return cond_factored(m, w, cf, B, T, (hipStream_t)stream);
}

int cmtts_poll_error(void) { return check_device_flag(); }

int cmtts_set_persistent_denoiser(int mode) {
    const int prev = g_persist;
    if (mode >= 0 && mode <= 2) g_persist = mode;
    return prev;
}

int cmtts_set_fused_resblock(int on) {
    const int prev = g_fused_resblock ? 1 : 0;
    g_fused_resblock = on != 0;
    return prev;
}

int cmtts_set_precision(cmtts_model* m, int mode) {
    if (!m || mode < 0 || mode > 3) return fail(CMTTS_E_INVALID, "cmtts_set_precision: mode 0 (fp32), 1 (bf16), 2 (fp16) or 3 (fp16x3)");
    m->precision = mode;
    return 0;
}

int cmtts_set_variance_controls(cmtts_model* m, const cmtts_variance_controls* vc) {
    if (!m) return fail(CMTTS_E_INVALID, "cmtts_set_variance_controls: null model");
    const cmtts_variance_controls off = {1.f, 1.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!vc) { m->vc = off; return 0; }
    if (!(vc->p_control > 0.f) || !(vc->e_control == vc->e_control))
        return fail(CMTTS_E_INVALID, "cmtts_set_variance_controls: p_control must be > 0 (it also scales the uv logit)");
    if (vc->cwt_spec && (!vc->f0_mean || !vc->f0_std || (m->cfg.use_uv && !vc->uv)))
        return fail(CMTTS_E_INVALID, "cmtts_set_variance_controls: a pitch target needs cwt_spec, f0_mean, f0_std (and uv)");
    m->vc = *vc;
    return 0;
}

int cmtts_vocoder_set_precision(cmtts_vocoder* v, int mode) {
    if (!v || mode < 0 || mode > 3) return fail(CMTTS_E_INVALID, "cmtts_vocoder_set_precision: mode 0 (fp32), 1 (bf16), 2 (fp16) or 3 (fp16x3)");
    v->precision = mode;
    return 0;
}

int cmtts_set_resblock_tile(int frames) {
    if (frames != 0 && frames != 32 && frames != 64) return fail(CMTTS_E_INVALID, "cmtts_set_resblock_tile: 0, 32 or 64");
    cmtts_resblock_set_tile(frames);
    return 0;
}

int cmtts_set_debug_stamps(void* dev_buf) {
    cmtts_resblock_set_debug((long long*)dev_buf);
    cmtts_persist_set_debug((long long*)dev_buf);
    cmtts_pair_set_debug((long long*)dev_buf);
    cmtts_xres_set_debug((long long*)dev_buf);
    {   // generic conv kernel: CMTTS_CONV_DBG_MK="M,K" selects the launches to stamp (tools/conv_phases.py)
        int M = 0, K = 0;
        const char* mk = getenv("CMTTS_CONV_DBG_MK");
        if (mk && sscanf(mk, "%d,%d", &M, &K) == 2) cmtts_conv_set_debug(dev_buf ? (long long*)dev_buf : nullptr, M, K);
        else cmtts_conv_set_debug(nullptr, 0, 0);
    }
    return 0;
}

int cmtts_profile_begin(int max_launches, int stride) {
    if (max_launches <= 0 || stride <= 0) return fail(CMTTS_E_INVALID, "cmtts_profile_begin: bad argument");
    g_prof.stride = stride;
    g_prof.seen = 0;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.assign((size_t)max_launches * 2, nullptr);
    for (auto& e : g_prof.ev) HIPCHK(hipEventCreate(&e));
    g_prof.used = 0;
    g_prof.on = true;
    return 0;
}
int cmtts_profile_end(double* total_ms, int* n_launches) {
    if (!total_ms || !n_launches) return fail(CMTTS_E_INVALID, "cmtts_profile_end: bad argument");
    g_prof.on = false;
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        HIPCHK(hipEventSynchronize(g_prof.ev[i + 1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *n_launches = (int)(g_prof.used / 2);
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.used = 0;
    return 0;
}

int cmtts_wav_to_int16(const float* wav, int16_t* pcm, int64_t n, float max_wav_value, void* stream) {
    if (!wav || !pcm || n < 0) return fail(CMTTS_E_INVALID, "cmtts_wav_to_int16: bad argument");
    if (n) k_wav_to_int16(wav, pcm, (long)n, max_wav_value, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

size_t cmtts_decoder_workspace_bytes(const cmtts_model* m, int B, int T) { return carve_text(m->cfg, B, T, nullptr).bytes; }

// FastspeechDecoder.forward (model/modules.py:154-165 -> FFTBlocks.forward :80-105 with use_pos_embed=True):
// x + alpha * PE[positions(x[..., 0] != 0)], masked, 4 FFT blocks, final LayerNorm (eps 1e-5), masked.
int cmtts_decoder_forward(cmtts_model* m, const float* x_ct, const int64_t* lens, int B, int T, float* out_ct, void* ws,
                          size_t ws_bytes, void* stream) {
    if (!m || !m->finalized) return fail(CMTTS_E_INVALID, "model not finalized");
    if (m->dec.empty()) return fail(CMTTS_E_INVALID, "cmtts_decoder_forward: the state dict held no decoder.* tensors");
    if (!x_ct || !lens || !out_ct || !ws || B <= 0 || T <= 0) return fail(CMTTS_E_INVALID, "cmtts_decoder_forward: bad argument");
    if (T + 1 >= PE_ROWS) return fail(CMTTS_E_UNSUPPORTED, "cmtts_decoder_forward: T exceeds the position table");
    const cmtts_config& c = m->cfg;
    TextWs w = carve_text(c, B, T, ws);
    if (ws_bytes < w.bytes) return fail(CMTTS_E_WORKSPACE, "decoder workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int H = c.hidden, Tp = round_up(T, 4);
    k_copy_rows(w.f, Tp, x_ct, T, T, (long)B * H, s);
    k_pos_embed_add(w.f, w.x, m->dec_alpha, m->omega_h, m->pe_h, PE_ROWS, B, H, T, Tp, s, lens);
    CHK(fft_stack(m, m->dec, w, lens, B, T, s));
    k_layernorm_ct(w.x, w.x, m->decln_g, m->decln_b, 1e-5f, lens, B, T, Tp, s);
    k_copy_rows(out_ct, T, w.x, Tp, T, (long)B * H, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int cmtts_length_mask(const int64_t* lens, uint8_t* mask, int B, int W, void* stream) {
    if (!lens || !mask || B <= 0 || W < 0) return fail(CMTTS_E_INVALID, "cmtts_length_mask: bad argument");
    if (W == 0) return 0;
    k_length_mask(lens, mask, B, W, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int cmtts_transpose(const float* in, float* out, int B, int R, int C, void* stream) {
    if (!in || !out || B <= 0 || R <= 0 || C <= 0) return fail(CMTTS_E_INVALID, "cmtts_transpose: bad argument");
    k_transpose(in, out, B, R, C, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int cmtts_pack_conv_weight(const float* host_w, int Cout, int Cin, int K, float** dev_packed, int* ld) {
    if (!host_w || !dev_packed || !ld || Cout <= 0 || Cin <= 0 || K <= 0) return fail(CMTTS_E_INVALID, "cmtts_pack_conv_weight: bad argument");
    HostTensor W;
    W.shape = {Cout, Cin, K};
    W.data.assign(host_w, host_w + (size_t)Cout * Cin * K);
    Allocs al;
    PackedConv p;
    CHK(pack_conv(al, W, nullptr, nullptr, &p));
    *dev_packed = p.w;
    *ld = p.ld;
    return 0;
}
void cmtts_free_device(void* p) {
    if (p) (void)hipFree(p);
}
int cmtts_conv1d(const float* x, const float* packed_w, int ld, const float* bias, int B, int Cin, int Cout, int T, int K,
                 int dilation, int padding, int act, float* y, void* stream) {
    if (!x || !packed_w || !y) return fail(CMTTS_E_INVALID, "cmtts_conv1d: bad argument");
    PackedConv w;
    w.w = const_cast<float*>(packed_w); w.bias = const_cast<float*>(bias);
    w.cout = Cout; w.cin = Cin; w.taps = K; w.ld = ld; w.tap_stride = (long)Cin * ld;
    const int Tout = T + 2 * padding - dilation * (K - 1);
    if (Tout <= 0) return fail(CMTTS_E_INVALID, "cmtts_conv1d: empty output");
    ConvArgs a = conv_args(w, x, T, T, (long)Cin * T, y, Tout, (long)Cout * Tout, Tout);
    a.dil = dilation; a.pad = padding;
    a.out[0].act = act;
    CHK(launch(a, EPI_PLAIN, B, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
