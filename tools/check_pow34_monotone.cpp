// Exhaustive check that the x^(3/4) fit (hx_dev.h hx_pow34: piecewise-linear mantissa fit x exponent table) is monotone
// non-decreasing over every non-negative float: the band maximum of x^(3/4) is then the x^(3/4) of the band's largest
// magnitude (hx_alloc3.inc, helper_prep).  g++ -O2 -ffp-contract=off -Ihmp3_amd/csrc tools/check_pow34_monotone.cpp hmp3_amd/csrc/hx_host.cpp -o /tmp/m && /tmp/m
#include <stdio.h>
#include <string.h>
#include <stdint.h>
#include <vector>
#include "hx_types.h"
#include "hx_host.h"
static inline float b2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main() {
    std::vector<HxGlobalTabs> gv(1); HxGlobalTabs &g = gv[0];
    hx_global_tabs(&g);
    long bad = 0; float prev = 0.0f; uint32_t firstbad = 0;
    for (uint32_t u = 0; u < 0x7f800000u; u++) {
        float x = b2f(u);
        float m = b2f((u & 0x7FFFFFu) | (127u << 23));
        unsigned seg = (u >> 19) & 15, e = (u >> 23) & 255;
        float y = (m * g.pow34_b[seg] + g.pow34_a[seg]) * g.pow34_exp[e];
        if (y < prev) { if (!bad) firstbad = u; bad++; }
        prev = y;
    }
    printf("non-monotone steps: %ld first at 0x%08x\n", bad, firstbad);
    return 0;
}
