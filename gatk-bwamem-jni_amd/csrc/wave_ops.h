// wave_ops.h -- 64-lane wavefront collectives used by the DP kernels (CDNA wave64).
#pragma once
#include <hip/hip_runtime.h>

#define WAVE 64
#define NEG_INF_I32 (-0x40000000)

// inclusive prefix maximum across the wave (Hillis-Steele over lane shuffles)
static __device__ inline int wave_prefix_max(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        int u = __shfl_up(v, o);
        if (lane >= o) v = v > u ? v : u;
    }
    return v;
}

static __device__ inline int wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int u = __shfl_xor(v, o); v = v > u ? v : u; }
    return v;
}

static __device__ inline int wave_bcast(int v, int src) { return __shfl(v, src); }

static __device__ inline unsigned long long wave_ballot(int pred) { return __ballot(pred); }

static __device__ inline bool wave_any(int pred) { return __ballot(pred) != 0ull; }
