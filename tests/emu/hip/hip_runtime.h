// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A host emulation of the small slice of the HIP programming model the kernels use, so that
// the *unchanged* product sources (gatk-bwamem-jni_amd/csrc/*.hip, pipeline.cpp) can be
// compiled with g++ and exercised against the CPU oracle in this GPU-less container
// (pytest -m "not gpu").  Every GPU thread is a ucontext fiber; wavefront collectives
// (__shfl*, __ballot) and __syncthreads() are rendezvous points between fibers.
// It is never built into, loaded by, or used as a fallback of the product library: the
// shipped libbwamem_hip.so is compiled by hipcc for gfx950 only (see __graft_entry__.build()).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <stdio.h>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_smem();

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint4 { uint32_t x, y, z, w; };
struct int4 { int32_t x, y, z, w; };

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
void* dyn_smem();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void sync_wave();
void sync_row();      // the 16 lanes of the calling thread's DPP row
void sync_block();
uint64_t* wave_slots();      // 64 slots of the calling thread's wave
uint64_t wave_alive_mask();
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })

static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }

template <typename T> static inline T emu_exchange(T v, int src)
{
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t* s = emu::wave_slots();
    int lane = threadIdx.x & 63;
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    s[lane] = raw;
    emu::sync_wave();
    T r = v;
    if (src >= 0 && src < 64 && (emu::wave_alive_mask() >> src & 1)) { uint64_t x = s[src]; memcpy(&r, &x, sizeof(T)); }
    emu::sync_wave();
    return r;
}
template <typename T> static inline T __shfl(T v, int src) { return emu_exchange(v, src & 63); }
template <typename T> static inline T __shfl_up(T v, unsigned d) { int l = threadIdx.x & 63; return emu_exchange(v, l - (int)d); }
template <typename T> static inline T __shfl_down(T v, unsigned d) { int l = threadIdx.x & 63; return emu_exchange(v, l + (int)d > 63 ? -1 : l + (int)d); }
template <typename T> static inline T __shfl_xor(T v, int m) { int l = threadIdx.x & 63; return emu_exchange(v, l ^ m); }
static inline unsigned long long __ballot(int pred)
{
    uint64_t* s = emu::wave_slots();
    int lane = threadIdx.x & 63;
    s[lane] = pred ? 1 : 0;
    emu::sync_wave();
    unsigned long long m = 0, alive = emu::wave_alive_mask();
    for (int i = 0; i < 64; ++i) if ((alive >> i & 1) && s[i]) m |= 1ull << i;
    emu::sync_wave();
    return m;
}
static inline void __syncthreads() { emu::sync_block(); }
static inline long long clock64() { return 0; }

// DPP subset used by wave_ops.h (gfx9 controls quad_perm, row_shl:n, row_shr:n, wave_shl/shr:1, row_mirror, row_half_mirror,
// row_bcast:15, row_bcast:31).  Controls that stay inside a 16-lane row rendezvous only the lanes of that row, so that the
// four rows of a wave may run different control flow (as the exec mask lets them on the hardware).
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    uint64_t* s = emu::wave_slots();
    int lane = threadIdx.x & 63, row = lane >> 4, in_row = lane & 15;
    const bool row_local = ctrl < 0x130 || ctrl == 0x140 || ctrl == 0x141;
    s[lane] = (uint32_t)src;
    if (row_local) emu::sync_row(); else emu::sync_wave();
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xff) from = (lane & ~3) | (ctrl >> (2 * (lane & 3)) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { int n = ctrl - 0x100; from = in_row + n <= 15 ? lane + n : -1; }
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { int n = ctrl - 0x110; from = in_row >= n ? lane - n : -1; }
    else if (ctrl == 0x138) from = lane >= 1 ? lane - 1 : -1;
    else if (ctrl == 0x130) from = lane < 63 ? lane + 1 : -1;
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - in_row);
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
    else if (ctrl == 0x142) from = row >= 1 ? (row << 4) - 1 : -1;
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;
    else { fprintf(stderr, "emu: unsupported DPP control 0x%x\n", ctrl); abort(); }
    bool enabled = (row_mask >> row & 1) && (bank_mask >> (in_row >> 2) & 1);
    int r = old;
    if (enabled) {
        if (from >= 0 && (emu::wave_alive_mask() >> from & 1)) r = (int)(uint32_t)s[from];
        else if (bound_ctrl) r = 0;
    }
    if (row_local) emu::sync_row(); else emu::sync_wave();
    return r;
}
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::sync_row()
#define __builtin_amdgcn_update_dpp emu_update_dpp
static inline int emu_readlane(int v, int lane) { return __shfl(v, lane); }
#define __builtin_amdgcn_readlane emu_readlane

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }

// ---- runtime API subset
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
typedef void* hipStream_t;
typedef void* hipEvent_t;
static inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63, hipDeviceAttributeMaxSharedMemoryPerBlock = 74 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMaxSharedMemoryPerBlock ? 160 << 10 : 1; return hipSuccess; }   // one "CU": small grids, so the kernels' work queues get exercised
static inline hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 256) == 0 ? (memset(*p, 0xCD, n), hipSuccess) : hipErrorUnknown; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = *total_b = (size_t)1 << 36; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return posix_memalign(p, 256, n ? n : 256) == 0 ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
