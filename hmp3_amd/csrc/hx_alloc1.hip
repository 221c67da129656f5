// hx_alloc1.hip - the allocator kernel for streams of the reference's first-generation allocator (joint stereo with
// an intensity part, dual channel) at the MPEG-1 rates: the stream walk of hx_alloc.hip with hx_alloc1.inc in
// place of the long / short block allocators.  Its own translation unit like hx_alloc_lsf.hip.
#define HX_A1 1
#include "hx_alloc.hip"
