// GPU unit harness for the device introsort (debugging / regression): runs ks_introsort on small
// AlnReg arrays inside a kernel and prints the resulting order.  Usage: sort_unit <variant>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dev_common.h"
#include "post_common.h"

template <typename T, typename LT>
static __device__ void introsort_v2(size_t n, T* a, LT lt)
{   // same decisions as ks_introsort, pivot compared through a pointer, bounded scans, no early 'continue'
    struct Frame { T* left; T* right; int depth; };
    Frame stack[66];
    int sp = 0;
    if (n < 1) return;
    if (n == 2) { if (lt(a[1], a[0])) { T tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return; }
    int d;
    for (d = 2; 1ul << d < n; ++d);
    T *s = a, *t = a + (n - 1);
    d <<= 1;
    bool done = false;
    while (!done) {
        if (s < t) {
            --d;
            if (d == 0) { ks_combsort((size_t)(t - s + 1), s, lt); t = s; }
            else {
                T *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
                if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
                else k = lt(*j, *i) ? i : j;
                if (k != t) { T tmp = *k; *k = *t; *t = tmp; }
                const T* rp = t;
                bool more = true;
                while (more) {
                    ++i; while (i < t && lt(*i, *rp)) ++i;
                    --j; while (i <= j && lt(*rp, *j)) --j;
                    if (j <= i) more = false;
                    else { T tmp = *i; *i = *j; *j = tmp; }
                }
                { T tmp = *i; *i = *t; *t = tmp; }
                if (i - s > t - i) {
                    if (i - s > 16) { stack[sp].left = s; stack[sp].right = i - 1; stack[sp].depth = d; ++sp; }
                    s = t - i > 16 ? i + 1 : t;
                } else {
                    if (t - i > 16) { stack[sp].left = i + 1; stack[sp].right = t; stack[sp].depth = d; ++sp; }
                    t = i - s > 16 ? i - 1 : s;
                }
            }
        } else if (sp == 0) {
            ks_insertsort(a, a + n, lt);
            done = true;
        } else { --sp; s = stack[sp].left; t = stack[sp].right; d = stack[sp].depth; }
    }
}

__global__ void k_sort(AlnReg* a, int n, int variant)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (variant == 0) ks_introsort((size_t)n, a, RegSLt());
    else if (variant == 1) introsort_v2((size_t)n, a, RegSLt());
    else if (variant == 2) ks_insertsort(a, a + n, RegSLt());
}

int main(int argc, char** argv)
{
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    AlnReg h[3]; memset(h, 0, sizeof h);
    h[0].score = 86;  h[0].rb = 2101721; h[0].re = 2101871; h[0].qe = 150; h[0].rid = 4;
    h[1].score = 133; h[1].rb = 3813027; h[1].re = 3813177; h[1].qe = 150; h[1].rid = 4;
    h[2].score = 133; h[2].rb = 4495819; h[2].re = 4495969; h[2].qe = 150; h[2].rid = 3;
    AlnReg* d;
    hipMalloc((void**)&d, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sort, dim3(1), dim3(64), 0, 0, d, 3, variant);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("variant %d: %s order: %d/%lld %d/%lld %d/%lld\n", variant, hipGetErrorString(e), h[0].score, (long long)h[0].rb, h[1].score, (long long)h[1].rb, h[2].score, (long long)h[2].rb);
    return 0;
}
