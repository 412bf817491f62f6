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

// ---- 16-lane groups ("rows" of the DPP network): collectives that stay inside a row, usable where the four rows of a wave
// run different control flow (k_extend's group form).  Row-local DPP controls: row_shr:n, row_shl:n = 0x100+n,
// quad_perm = 0x00..0xff, row_mirror = 0x140, row_half_mirror = 0x141.
#define GROUP 16
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_QUAD_PERM(a, b, c, d) ((a) | (b) << 2 | (c) << 4 | (d) << 6)
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_HALF_MIRROR 0x141

static __device__ inline int row_prefix_max(int v)            // inclusive, over the lanes of the row
{
    const int id = (int)0x80000000;
    int t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(1), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(2), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(4), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(id, v, DPP_ROW_SHR(8), 0xf, 0xf, false); v = v > t ? v : t;
    return v;
}
static __device__ inline int row_prefix_sum(int v)            // inclusive
{
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, false);
    return v;
}
// lane l <- lane l-1 of the row (first lane <- fill); lane l <- lane l+1 (last lane <- fill)
static __device__ inline int row_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_ROW_SHR(1), 0xf, 0xf, false); }
static __device__ inline int row_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_ROW_SHL(1), 0xf, 0xf, false); }
// butterfly over the 16 lanes: every lane ends up with the row's maximum / sum
static __device__ inline int row_all_max(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, DPP_QUAD_PERM(1, 0, 3, 2), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(v, v, DPP_QUAD_PERM(2, 3, 0, 1), 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_HALF_MIRROR, 0xf, 0xf, false); v = v > t ? v : t;
    t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_MIRROR, 0xf, 0xf, false); v = v > t ? v : t;
    return v;
}
static __device__ inline int row_all_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(v, v, DPP_QUAD_PERM(1, 0, 3, 2), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, DPP_QUAD_PERM(2, 3, 0, 1), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, DPP_ROW_MIRROR, 0xf, 0xf, false);
    return v;
}
// order the row's memory accesses around a hand-over between its lanes (a wave executes in lockstep: no barrier instruction needed)
static __device__ inline void row_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
