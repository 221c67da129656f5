/* hxo_short.c - ORACLE (test infrastructure): short-block (3 x 192) allocator.
 * Restates CBitAlloShort (bitallos.cpp:128-1503, bitallosc.cpp:296-428).
 * ROUND-1 STATUS: not yet restated - a short granule aborts loudly so that no parity claim
 * can silently rest on it (tests force long blocks with short_block_threshold = 99999). */
#include <stdio.h>
#include <stdlib.h>
#include "hxo_int.h"

void hxo_short_init(hxo_encoder *e) { e->s.s_call_count = 0; }

int hxo_ms_metric_short(hxo_encoder *e, const float x[2][576])
{
    (void) e; (void) x;
    fprintf(stderr, "hxo: short-block M/S metric not restated yet\n");
    abort();
}

int hxo_bitallo_short(hxo_encoder *e, float xr[2][576], hxo_sigmask sm[2][36],
                      int min_bits, int target_bits, int max_bits, int bit_pool,
                      hxo_scalefact sf_out[2], hxo_gr gr[2], int ms_flag, int MNR)
{
    (void) e; (void) xr; (void) sm; (void) min_bits; (void) target_bits; (void) max_bits;
    (void) bit_pool; (void) sf_out; (void) gr; (void) ms_flag; (void) MNR;
    fprintf(stderr, "hxo: short-block allocator not restated yet\n");
    abort();
}
