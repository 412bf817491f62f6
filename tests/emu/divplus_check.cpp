// tests/emu/divplus_check.cpp -- exhaustive small-range check that the integer form of upstream's gap-bound arithmetic
// (dev_common.h div_plus) equals the double expression it replaces.  Built and run by tests/test_host_math.py.
#include <hip/hip_runtime.h>
#include "dev_common.h"
#include <stdio.h>
int main()
{
    long bad = 0, n = 0;
    for (int k = 1; k <= 2; ++k)
        for (int e = 1; e <= 64; ++e)
            for (int x = -20000; x <= 20000; ++x) {
                int want = (int)((double)x / e + (double)k);
                if (div_plus(x, e, k) != want) { if (bad < 5) printf("x=%d e=%d k=%d want %d got %d\n", x, e, k, want, div_plus(x, e, k)); ++bad; }
                ++n;
            }
    for (int e : { 100, 1000, 65536, 1 << 20 })
        for (long x = -3000000; x <= 3000000; x += 7) {
            int want = (int)((double)(int)x / e + 1.);
            if (div_plus((int)x, e, 1) != want) ++bad;
            ++n;
        }
    printf("checked %ld cases, %ld mismatches\n", n, bad);
    return bad != 0;
}
