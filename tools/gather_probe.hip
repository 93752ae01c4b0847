// How fast can a wave pull a tile out of a channel-major fp32 tensor [B][C][ld]?  The HiFi-GAN epilogues / stagings read and write in
// the MFMA accumulator layout: one instruction = two 128-B row segments (lanes 0-31 one row, lanes 32-63 another), 16 instructions per
// 32 x 32 tile.  Compared with (b) the same bytes as 512-B row runs (64 lanes x 8 B: four adjacent tiles of one row) and (c) 1-KB row
// runs (64 lanes x 16 B).  Prints B/clk per CU (2.4 GHz nominal) per pattern, waves per CU and loads in flight.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gather_probe tools/gather_probe.hip && tools/bin/gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 128;

// pattern 0: accumulator layout, DEPTH x 16 dword loads in flight (DEPTH tiles)
template <int DEPTH, bool WRITE>
__global__ __launch_bounds__(256) void acc_kernel(const float* __restrict__ x, float* __restrict__ y, int ld, long bstride, int tiles_per_wave, int nwaves_total) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    float s = 0.f;
    for (int it = 0; it < tiles_per_wave; it += DEPTH) {
        float v[DEPTH][16];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long tile = (long)wave * tiles_per_wave + it + d;      // tile id -> (batch, m-tile, column block)
            const long colblk = tile % (ld / 32), rest = tile / (ld / 32);
            const int mt = rest % (C / 32);
            const long b = rest / (C / 32);
            const float* p = x + b * bstride + (long)(mt * 32) * ld + colblk * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[d][r] = p[(long)((r & 3) + 8 * (r >> 2) + 4 * kh) * ld];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long tile = (long)wave * tiles_per_wave + it + d;
            const long colblk = tile % (ld / 32), rest = tile / (ld / 32);
            const int mt = rest % (C / 32);
            const long b = rest / (C / 32);
            float* q = y + b * bstride + (long)(mt * 32) * ld + colblk * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (WRITE) q[(long)((r & 3) + 8 * (r >> 2) + 4 * kh) * ld] = v[d][r] + 1.f;
                else s += v[d][r];
            }
        }
    }
    if (!WRITE && s == 12345.678f) y[0] = s;
}

// pattern 1/2: row runs: VEC floats per lane (2 -> 512 B per instruction, 4 -> 1 KB), the same tile set re-ordered: a wave takes
// 32 rows x (64 * VEC) columns = VEC * 2 tiles' worth per "super tile", 32 instructions each
template <int VEC, int DEPTH, bool WRITE>
__global__ __launch_bounds__(256) void row_kernel(const float* __restrict__ x, float* __restrict__ y, int ld, long bstride, int st_per_wave) {
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    constexpr int WCOLS = 64 * VEC;
    float s = 0.f;
    for (int it = 0; it < st_per_wave; ++it) {
        const long st = (long)wave * st_per_wave + it;
        const long colblk = st % (ld / WCOLS), rest = st / (ld / WCOLS);
        const int mt = rest % (C / 32);
        const long b = rest / (C / 32);
        const float* p = x + b * bstride + (long)(mt * 32) * ld + colblk * WCOLS + lane * VEC;
        float* q = y + b * bstride + (long)(mt * 32) * ld + colblk * WCOLS + lane * VEC;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += DEPTH) {
            vt v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const vt*>(p + (long)(r0 + d) * ld);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (WRITE) { vt o = v[d]; for (int e = 0; e < VEC; ++e) o[e] += 1.f; *reinterpret_cast<vt*>(q + (long)(r0 + d) * ld) = o; }
                else for (int e = 0; e < VEC; ++e) s += v[d][e];
            }
        }
    }
    if (!WRITE && s == 12345.678f) y[0] = s;
}

int main() {
    const int B = 32, ld = 32768;
    const long bstride = (long)C * ld;
    const size_t n = (size_t)B * bstride;
    float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
    hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long tiles = (long)B * (C / 32) * (ld / 32);
    const double bytes = (double)n * 4;
    auto report = [&](const char* name, int wpc, float ms, bool wr) {
        const double moved = bytes * (wr ? 2 : 1);
        printf("%-34s waves/CU %2d: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU\n", name, wpc, ms, moved / (ms * 1e-3) / 1e12, moved / (ms * 1e-3) / 256 / 2.4e9);
    };
    for (int wpc : {4, 8, 16, 32}) {
        const int nwaves = 256 * wpc, blocks = nwaves / 4;
        const int tpw = (int)(tiles / nwaves);
        float ms;
#define RUN(K, NAME, WR) hipLaunchKernelGGL(K); hipEventRecord(e0); hipLaunchKernelGGL(K); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report(NAME, wpc, ms, WR);
#define A1 (acc_kernel<1, false>), dim3(blocks), dim3(256), 0, 0, x, y, ld, bstride, tpw, nwaves
#define A4 (acc_kernel<4, false>), dim3(blocks), dim3(256), 0, 0, x, y, ld, bstride, tpw, nwaves
#define A4W (acc_kernel<4, true>), dim3(blocks), dim3(256), 0, 0, x, y, ld, bstride, tpw, nwaves
#define R2 (row_kernel<2, 16, false>), dim3(blocks), dim3(256), 0, 0, x, y, ld, bstride, tpw / 4
#define R4 (row_kernel<4, 16, false>), dim3(blocks), dim3(256), 0, 0, x, y, ld, bstride, tpw / 8
#define R4W (row_kernel<4, 16, true>), dim3(blocks), dim3(256), 0, 0, x, y, ld, bstride, tpw / 8
        RUN(A1, "acc layout, 16 loads in flight", false)
        RUN(A4, "acc layout, 64 loads in flight", false)
        RUN(A4W, "acc layout, 64 in flight, r+w", true)
        RUN(R2, "512-B row runs, 16 in flight", false)
        RUN(R4, "1-KB row runs, 16 in flight", false)
        RUN(R4W, "1-KB row runs, 16 in flight, r+w", true)
    }
    return 0;
}
