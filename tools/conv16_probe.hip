// Shape sweep of the 16-bit Conv1D kernel (conv_mfma16.hip) through cmtts_launch_conv16, to separate the
// effects of M, K, N and taps.  Build: hipcc --offload-arch=gfx950 -O2 -I cm-tts_amd/csrc tools/conv16_probe.hip
//   -L cm-tts_amd -lcmtts_hip -Wl,-rpath,'$ORIGIN/../../cm-tts_amd' -o tools/bin/conv16_probe
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "conv_args.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = 32;
    struct Shape { int C, T, taps, dil, res; };
    std::vector<Shape> shapes;
    for (int res = 0; res < 2; ++res)
        for (int taps : {3, 11}) {
            shapes.push_back({256, 4096, taps, 1, res});
            shapes.push_back({256, 8192, taps, 1, res});
            shapes.push_back({128, 8192, taps, 1, res});
            shapes.push_back({128, 16384, taps, 1, res});
            shapes.push_back({256, 16384, taps, 1, res});
            shapes.push_back({64, 65536, taps, 1, res});
        }
    size_t maxel = (size_t)B * 256 * 16384;
    float *X, *Y, *R, *bias;
    void* W;
    CK(hipMalloc(&X, maxel * 4)); CK(hipMalloc(&Y, maxel * 4)); CK(hipMalloc(&R, maxel * 4));
    CK(hipMalloc(&bias, 1024 * 4)); CK(hipMalloc(&W, (size_t)11 * 256 * 256 * 2 + 4096));
    CK(hipMemset(R, 0, maxel * 4)); CK(hipMemset(bias, 0, 4096));
    const bool rnd = argc > 1 && atoi(argv[1]) != 0;
    {   // random activations in [-1, 1) and bf16 weights of magnitude ~2^-6 (or all-zero X / constant W with arg 0)
        std::vector<float> hx(maxel);
        unsigned st = 12345u;
        for (size_t i = 0; i < maxel; ++i) { st = st * 1664525u + 1013904223u; hx[i] = rnd ? (float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f : 0.f; }
        CK(hipMemcpy(X, hx.data(), maxel * 4, hipMemcpyHostToDevice));
        std::vector<unsigned short> hw((size_t)11 * 256 * 256);
        for (size_t i = 0; i < hw.size(); ++i) { st = st * 1664525u + 1013904223u; hw[i] = rnd ? (unsigned short)(0x3c00u | ((st >> 9) & 0x80ffu)) : 0x3c3cu; }
        CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.X = X; a.M = s.C; a.K = s.C; a.N = s.T; a.taps = s.taps; a.dil = s.dil; a.pad = (s.taps - 1) / 2 * s.dil;
        a.Tin = s.T; a.ldx = s.T; a.zdiv = 1; a.x_zs0 = (long)s.C * s.T; a.pre_div = 1.f; a.pre_slope = 0.1f; a.split = INT_MAX;
        ConvOut& o = a.out[0];
        o.Y = Y; o.y_zs0 = (long)s.C * s.T; o.ldy = s.T; o.Tout = s.T; o.ostride = 1; o.bias = bias; o.alpha = 1.f; o.div = 1.f;
        if (s.res) { o.res = R; o.r_zs0 = (long)s.C * s.T; o.ldr = s.T; }
        a.out[1] = a.out[0];
        for (int i = 0; i < 3; ++i) if (cmtts_launch_conv16(&a, W, 1, B, nullptr) != 0) { printf("launch failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        const int reps = 10;
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) cmtts_launch_conv16(&a, W, 1, B, nullptr);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const double flops = 2.0 * s.C * s.C * s.taps * (double)s.T * B;
        const double bytes = (double)(2 + s.res) * s.C * s.T * B * 4;
        printf("C=%3d T=%6d taps=%2d res=%d: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s\n", s.C, s.T, s.taps, s.res, us, flops / us * 1e-6, bytes / us * 1e-6);
    }
    return 0;
}
