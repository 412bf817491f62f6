// Micro-benchmark: what random-gather rate does an MI355X sustain for DEPENDENT 32-byte (or 64-byte) block reads from a
// table far larger than the caches?  This is the practical ceiling of FM-index seeding (each interval extension needs
// the previous one's result).  Usage: gather_bench <table_GiB> <waves_per_cu> <ilp> <bytes 16|32|64>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

template <int ILP, int VEC>
__global__ void k_gather(const uint4* tab, uint64_t n_blocks, int steps, uint64_t* out)
{
    uint64_t x[ILP];
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < ILP; ++k) x[k] = (tid * 0x9E3779B97F4A7C15ull + k * 0xD1B54A32D192ED03ull) % n_blocks;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const uint4* p = tab + x[k] * VEC;
            uint64_t acc = 0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { uint4 q = p[v]; acc += (uint64_t)q.x + q.y + q.z + q.w; }
            x[k] = ((x[k] + acc) * 0x9E3779B97F4A7C15ull >> 7) % n_blocks;      // next address depends on the data
        }
    }
    uint64_t r = 0;
    for (int k = 0; k < ILP; ++k) r ^= x[k];
    out[tid] = r;
}

template <int ILP, int VEC>
static void run(const uint4* tab, uint64_t n_blocks, int waves_per_cu, uint64_t* out)
{
    const int steps = 2000, n_cu = 256;
    dim3 grid(n_cu * waves_per_cu), block(64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<ILP, VEC>), grid, block, 0, 0, tab, n_blocks, 200, out);   // warm-up
    hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<ILP, VEC>), grid, block, 0, 0, tab, n_blocks, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)grid.x * 64 * ILP * steps;
    printf("waves/CU=%2d ilp=%d bytes=%2d: %.2f ms, %.2f G gathers/s, %.1f GB/s useful, %.2f us per dependent step\n",
           waves_per_cu, ILP, VEC * 16, ms, n / ms / 1e6, n * VEC * 16 / ms / 1e6, ms * 1e3 / steps);
}

int main(int argc, char** argv)
{
    double gib = argc > 1 ? atof(argv[1]) : 3.0;
    uint64_t bytes = (uint64_t)(gib * (1ull << 30));
    uint4* tab; uint64_t* out;
    hipMalloc((void**)&tab, bytes); hipMemset(tab, 1, bytes);
    hipMalloc((void**)&out, (uint64_t)256 * 32 * 64 * 8);
    for (int vec : {2, 4, 1}) {
        uint64_t n_blocks = bytes / (16ull * vec);
        for (int w : {1, 4, 8, 12, 16, 24, 32}) {
            if (vec == 2) { run<1, 2>(tab, n_blocks, w, out); if (w >= 8) { run<2, 2>(tab, n_blocks, w, out); run<4, 2>(tab, n_blocks, w, out); } }
            if (vec == 4) run<1, 4>(tab, n_blocks, w, out);
            if (vec == 1) run<1, 1>(tab, n_blocks, w, out);
        }
    }
    return 0;
}
