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

// ---- DPP (data-parallel primitives) forms: cross-lane moves folded into VALU instructions, no LDS round trip.
// gfx9 controls: row_shr:n = 0x110+n (within a 16-lane row), wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHR1 0x138
#define DPP_WAVE_SHL1 0x130
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// inclusive prefix maximum over the 64 lanes.  Lanes without a source keep their own value: the DPP "old" operand is
// INT_MIN, the identity of v_max_i32, which lets the compiler fold each move into the max (one v_max_i32_dpp per step).
static __device__ inline int dpp_prefix_max(int v, int /*neg*/)
{
    const int id = (int)0x80000000;
    int t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(1), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(2), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(4), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(8), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_BCAST15, 0xa, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_BCAST31, 0xc, 0xf, false); v = v > t ? v : t;
    return v;
}
// lane l <- lane l-1; lane 0 <- fill
static __device__ inline int dpp_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xf, 0xf, false); }
// lane l <- lane l+1; lane 63 <- fill
static __device__ inline int dpp_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHL1, 0xf, 0xf, false); }
// value of a (wave-uniform) lane as a scalar
static __device__ inline int wave_readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
