// hx_alloc1_lsf.hip - the same for the MPEG-2 LSF rates (one granule per frame, no pre-emphasis, three-group
// intensity scalefactors).
#define HX_A1 1
#define HX_LSF 1
#include "hx_alloc.hip"
