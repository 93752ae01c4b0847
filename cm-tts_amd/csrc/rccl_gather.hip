// The path's one collective behind the C ABI (include/cmtts_hip.h: cmtts_comm_* / cmtts_allgather_mels): pack the padded
// mel block and mel_len into one buffer, ONE ncclAllGather over RCCL/xGMI, unpack in rank order.  New work:
// the reference's inference is single-process (synthesize.py:32,43).  librccl.so is opened lazily with dlopen — the library
// has no link-time dependency on it, so single-GPU users (and the CPU-side symbol checks) never touch RCCL.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/cmtts_hip.h"

namespace {


typedef int (*fn_get_unique_id)(void*);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct NcclUniqueId { char internal[128]; };        // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
typedef int (*fn_comm_init_rank_v)(void**, int, NcclUniqueId, int);

struct Rccl {
    void* h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank_v comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    bool tried = false;
} g_rccl;

bool rccl_open() {
    if (g_rccl.h) return true;
    if (g_rccl.tried) return false;
    g_rccl.tried = true;
    // the copy torch already mapped (RTLD_NOLOAD) first, so that a Python host shares one RCCL with torch.distributed
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h)
        for (const char* n : names) {
            g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.h) break;
        }
    if (!g_rccl.h) return false;
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank_v)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.all_gather = (fn_all_gather)dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.error_string = (fn_error_string)dlsym(g_rccl.h, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_gather) {
        g_rccl.h = nullptr;
        return false;
    }
    return true;
}

// [Bl][T*M] mel + [Bl] mel_len -> [Bl][T*M + 1] (mel_len < 2^24 is exact in fp32)
__global__ void pack_kernel(const float* __restrict__ mel, const int64_t* __restrict__ mel_len, float* __restrict__ out, long row, int Bl) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < row) out[(long)b * (row + 1) + i] = mel[(long)b * row + i];
    if (i == 0) out[(long)b * (row + 1) + row] = (float)mel_len[b];
}
__global__ void unpack_kernel(const float* __restrict__ in, float* __restrict__ mel, int64_t* __restrict__ mel_len, long row) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < row) mel[(long)b * row + i] = in[(long)b * (row + 1) + i];
    if (i == 0) mel_len[b] = (int64_t)rintf(in[(long)b * (row + 1) + row]);
}

// int16 PCM rows (vocoder_infer's output, utils/model.py:187-205): [Bl][N] samples + [Bl] sample counts -> [Bl][rowp] int16 with
// the int64 count in the last four slots (rowp = N rounded up to a multiple of 4, + 4: the count sits 8-byte aligned)
__global__ void pack_pcm_kernel(const int16_t* __restrict__ pcm, const int64_t* __restrict__ len, int16_t* __restrict__ out, long N, long rowp) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < rowp - 4) out[(long)b * rowp + i] = i < N ? pcm[(long)b * N + i] : (int16_t)0;
    if (i == 0) *reinterpret_cast<int64_t*>(out + (long)b * rowp + rowp - 4) = len[b];
}
__global__ void unpack_pcm_kernel(const int16_t* __restrict__ in, int16_t* __restrict__ pcm, int64_t* __restrict__ len, long N, long rowp) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < N) pcm[(long)b * N + i] = in[(long)b * rowp + i];
    if (i == 0) len[b] = *reinterpret_cast<const int64_t*>(in + (long)b * rowp + rowp - 4);
}

}  // namespace


// cmtts_api.hip owns cmtts_last_error(); these entry points report through it
extern "C" int cmtts_internal_fail(int code, const char* msg);
extern "C" int cmtts_persist_note_process_group(int on);      // denoiser_persist.hip: persistent launches become cooperative

extern "C" {

int cmtts_comm_unique_id(void* id128_host) {
    if (!id128_host) return cmtts_internal_fail(CMTTS_E_INVALID, "cmtts_comm_unique_id: null argument");
    if (!rccl_open()) return cmtts_internal_fail(CMTTS_E_UNSUPPORTED, "librccl.so could not be opened (dlopen)");
    const int r = g_rccl.get_unique_id(id128_host);
    if (r != 0) return cmtts_internal_fail(CMTTS_E_HIP, g_rccl.error_string ? g_rccl.error_string(r) : "ncclGetUniqueId failed");
    return 0;
}

int cmtts_comm_init_rank(void** comm, int world, int rank, const void* id128_host) {
    if (!comm || !id128_host || world < 1 || rank < 0 || rank >= world) return cmtts_internal_fail(CMTTS_E_INVALID, "cmtts_comm_init_rank: bad argument");
    if (!rccl_open()) return cmtts_internal_fail(CMTTS_E_UNSUPPORTED, "librccl.so could not be opened (dlopen)");
    NcclUniqueId id;
    __builtin_memcpy(&id, id128_host, sizeof(id));
    const int r = g_rccl.comm_init_rank(comm, world, id, rank);
    if (r != 0) return cmtts_internal_fail(CMTTS_E_HIP, g_rccl.error_string ? g_rccl.error_string(r) : "ncclCommInitRank failed");
    cmtts_persist_note_process_group(1);     // RCCL kernels now share the GPU with the persistent grid: residency is checked by the runtime
    return 0;
}

int cmtts_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (!rccl_open()) return cmtts_internal_fail(CMTTS_E_UNSUPPORTED, "librccl.so could not be opened (dlopen)");
    const int r = g_rccl.comm_destroy(comm);
    if (r != 0) return cmtts_internal_fail(CMTTS_E_HIP, "ncclCommDestroy failed");
    return 0;
}

size_t cmtts_allgather_workspace_bytes(int world, int Bl, int T, int M) {
    const size_t row = (size_t)T * M + 1;
    return ((size_t)Bl * row + (size_t)world * Bl * row) * sizeof(float) + 512;
}

int cmtts_allgather_mels(void* comm, int world, const float* mel, const int64_t* mel_len, int Bl, int T, int M,
                         float* out_mel, int64_t* out_len, void* ws, size_t ws_bytes, void* stream) {
    if (!mel || !mel_len || !out_mel || !out_len || !ws || world < 1 || Bl <= 0 || T <= 0 || M <= 0)
        return cmtts_internal_fail(CMTTS_E_INVALID, "cmtts_allgather_mels: bad argument");
    if (ws_bytes < cmtts_allgather_workspace_bytes(world, Bl, T, M)) return cmtts_internal_fail(CMTTS_E_WORKSPACE, "all-gather workspace too small");
    if (world > 1 && !comm) return cmtts_internal_fail(CMTTS_E_INVALID, "cmtts_allgather_mels: a communicator is required for world > 1");
    hipStream_t s = (hipStream_t)stream;
    const long row = (long)T * M;
    float* send = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* recv = send + (size_t)Bl * (row + 1);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((row + 255) / 256), Bl), dim3(256), 0, s, mel, mel_len, send, row, Bl);
    const float* gathered = send;
    if (comm) {       // also with world == 1: the same RCCL call sequence on a 1-rank communicator (single-GPU tests)
        if (!rccl_open()) return cmtts_internal_fail(CMTTS_E_UNSUPPORTED, "librccl.so could not be opened (dlopen)");
        const int r = g_rccl.all_gather(send, recv, (size_t)Bl * (row + 1), /*ncclFloat*/ 7, comm, s);
        if (r != 0) return cmtts_internal_fail(CMTTS_E_HIP, g_rccl.error_string ? g_rccl.error_string(r) : "ncclAllGather failed");
        gathered = recv;
    }
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((row + 255) / 256), world * Bl), dim3(256), 0, s, gathered, out_mel, out_len, row);
    if (hipGetLastError() != hipSuccess) return cmtts_internal_fail(CMTTS_E_HIP, "all-gather pack/unpack launch failed");
    return 0;
}

static long pcm_row(int64_t N) { return ((N + 3) / 4) * 4 + 4; }

size_t cmtts_allgather_pcm_workspace_bytes(int world, int Bl, int64_t N) {
    return ((size_t)Bl + (size_t)world * Bl) * (size_t)pcm_row(N) * sizeof(int16_t) + 512;
}

int cmtts_allgather_pcm(void* comm, int world, const int16_t* pcm, const int64_t* wav_len, int Bl, int64_t N,
                        int16_t* out_pcm, int64_t* out_len, void* ws, size_t ws_bytes, void* stream) {
    if (!pcm || !wav_len || !out_pcm || !out_len || !ws || world < 1 || Bl <= 0 || N <= 0)
        return cmtts_internal_fail(CMTTS_E_INVALID, "cmtts_allgather_pcm: bad argument");
    if (ws_bytes < cmtts_allgather_pcm_workspace_bytes(world, Bl, N)) return cmtts_internal_fail(CMTTS_E_WORKSPACE, "pcm all-gather workspace too small");
    if (world > 1 && !comm) return cmtts_internal_fail(CMTTS_E_INVALID, "cmtts_allgather_pcm: a communicator is required for world > 1");
    hipStream_t s = (hipStream_t)stream;
    const long rowp = pcm_row(N);
    int16_t* send = (int16_t*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    int16_t* recv = send + (size_t)Bl * rowp;
    hipLaunchKernelGGL(pack_pcm_kernel, dim3((unsigned)((rowp + 255) / 256), Bl), dim3(256), 0, s, pcm, wav_len, send, N, rowp);
    const int16_t* gathered = send;
    if (comm) {       // bytes on the wire (RCCL has no int16 type): ncclInt8, count in bytes
        if (!rccl_open()) return cmtts_internal_fail(CMTTS_E_UNSUPPORTED, "librccl.so could not be opened (dlopen)");
        const int r = g_rccl.all_gather(send, recv, (size_t)Bl * rowp * sizeof(int16_t), /*ncclInt8*/ 0, comm, s);
        if (r != 0) return cmtts_internal_fail(CMTTS_E_HIP, g_rccl.error_string ? g_rccl.error_string(r) : "ncclAllGather failed");
        gathered = recv;
    }
    hipLaunchKernelGGL(unpack_pcm_kernel, dim3((unsigned)((N + 255) / 256), world * Bl), dim3(256), 0, s, gathered, out_pcm, out_len, N, rowp);
    if (hipGetLastError() != hipSuccess) return cmtts_internal_fail(CMTTS_E_HIP, "pcm all-gather pack/unpack launch failed");
    return 0;
}

}  // extern "C"
