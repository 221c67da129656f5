/* Compares hmp3_amd/csrc/hx_libm32.h (host build) with this machine's libm for EVERY positive normal float:
 *   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/check_libm32 tools/check_libm32.c -lm && /tmp/check_libm32
 * Prints the number of arguments whose logf / log10f differ (expected 0 on glibc 2.35). */
#ifndef STEP
#define STEP 1      /* -DSTEP=997: every 997th float (the CPU test suite's quick run) */
#endif
#include <math.h>
#include <stdio.h>
#define HX_HD static inline
#include "../hmp3_amd/csrc/hx_libm32.h"
int main(void)
{
    long bad_log = 0, bad_log10 = 0, n = 0;
    unsigned first_log = 0, first_log10 = 0;
#pragma omp parallel for reduction(+: bad_log, bad_log10, n) schedule(static)
    for (long u = 0x00800000L; u < 0x7f800000L; u += STEP) {
        const float x = hx_u2f((uint32_t) u);
        const float a = logf(x), b = hx_logf(x), c = log10f(x), d = hx_log10f(x);
        n++;
        if (hx_f2u(a) != hx_f2u(b)) { bad_log++; if (!first_log) first_log = (unsigned) u; }
        if (hx_f2u(c) != hx_f2u(d)) { bad_log10++; if (!first_log10) first_log10 = (unsigned) u; }
    }
    printf("%ld floats: logf differs for %ld (first 0x%08x), log10f for %ld (first 0x%08x)\n", n, bad_log, first_log, bad_log10, first_log10);
    return (bad_log || bad_log10) ? 1 : 0;
}
