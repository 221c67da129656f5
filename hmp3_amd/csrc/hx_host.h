// hx_host.h - host-side parameter resolution (see hx_host.cpp)
#pragma once
#include "hx_types.h"
void hx_host_default_control(HxControl *ec);
void hx_global_tabs(HxGlobalTabs *g);
int hx_resolve(const HxControl *ec, HxParams *p);
// after hx_resolve returned 0 on this thread: "" if the reference rejects the configuration too, else the limit of this
// library's kernel layout that it ran into (hx_host.cpp band_runs)
const char *hx_resolve_error(void);
void hx_stream_reset(const HxParams *p, int cls, HxStream *s);
// 1 if these tables have the structure the low-footprint allocator kernel (k_alloc_slim) derives them from: the gain tables
// as ldexp of their 4 / 16 mantissas, the x^(3/4) exponent table likewise, the mB tables within 16 bits (hx_alloc.hip, HX_SLIM)
int hx_slim_tables_ok(const HxParams *p, const HxGlobalTabs *g);
// the host's C library version, and how many of n sample arguments its logf / log10f treat differently from the restatement of
// glibc 2.35's that the first-generation allocator's kernels use (hx_libm32.h): 0 = the oracle on this box and the kernels agree
const char *hx_host_libc_version(void);
int hx_libm32_spot_check(int n);
