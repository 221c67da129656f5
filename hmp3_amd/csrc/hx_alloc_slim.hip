// hx_alloc_slim.hip - the MPEG-1 stream walk a third time: low-footprint LDS layout (HX_SLIM, see hx_alloc.hip) and a
// register budget of three waves per SIMD, i.e. six streams per CU instead of four.  For batches with more streams than
// the chip holds at once (hx_cabi.hip picks it by batch size); kernel k_alloc_slim.
#define HX_SLIM 1
#define HX_WAVES 3
#define HX_SEEK_FORCEINLINE 1     // the gain search inlined at its three call sites (as a call it saved and restored 54 registers per lane each time)
#include "hx_alloc.hip"
