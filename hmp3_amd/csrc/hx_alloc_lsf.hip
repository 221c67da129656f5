// hx_alloc_lsf.hip - the allocator / packer kernel for MPEG-2 LSF batches (k_alloc_lsf): the sources of
// hx_alloc.hip compiled a second time with HX_LSF = 1 (see the note at the end of hx_alloc3.inc).
#define HX_LSF 1
#include "hx_alloc.hip"
