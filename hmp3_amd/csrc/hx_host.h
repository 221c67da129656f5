// hx_host.h - host-side parameter resolution (see hx_host.cpp)
#pragma once
#include "hx_types.h"
void hx_host_default_control(HxControl *ec);
void hx_global_tabs(HxGlobalTabs *g);
int hx_resolve(const HxControl *ec, HxParams *p);
void hx_stream_reset(const HxParams *p, int cls, HxStream *s);
