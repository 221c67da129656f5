// hx_dev.h - device-side helpers shared by the HIP kernels (gfx950 only).
// All floating-point expressions are evaluated in the reference's order with contraction
// disabled (-ffp-contract=off), so that the kernels are bit-exact against the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include "hx_types.h"

#define HX_WAVE 64

__device__ __forceinline__ float hx_bits2f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned hx_f2bits(float f) { return __float_as_uint(f); }

// millibel log: top 8 mantissa bits -> table, 301 mB per octave (reference l3math.c:228-242)
__device__ __forceinline__ int hx_mblog(const int *tab, float x)
{
    unsigned u = hx_f2bits(x);
    return tab[(u >> 15) & 255] + 301 * (int) (u >> 23);
}

// inverse (reference l3math.c:342-356)
__device__ __forceinline__ float hx_mbexp(const float *lo, const float *hi, int x)
{
    float t = lo[(unsigned) x & 0xff] * hi[((unsigned) x & 0xff00) >> 8];
    if (x > 32000) return 1.0E32f;
    if (x < -32000) return 1.0E-32f;
    return t;
}

// (int)(x + copysign(0.5, x))  (reference l3math.c:361)
__device__ __forceinline__ int hx_round(float x) { return (int) (x + copysignf(0.5f, x)); }

// Wave-wide reductions and scans on the DPP data path (row shifts inside the 16-lane rows, then
// row_bcast:15 / row_bcast:31 across rows): VALU speed, where __shfl_* would be a ds_bpermute
// round trip through the LDS crossbar per step.  All 64 lanes must be active.
#define HX_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)
#define HX_SCAN_ROWS(v, ident, OP)                                                    \
    v = OP(v, HX_DPP(ident, v, 0x111, 0xf));    /* row_shr:1 */                        \
    v = OP(v, HX_DPP(ident, v, 0x112, 0xf));    /* row_shr:2 */                        \
    v = OP(v, HX_DPP(ident, v, 0x114, 0xf));    /* row_shr:4 */                        \
    v = OP(v, HX_DPP(ident, v, 0x118, 0xf));    /* row_shr:8 */                        \
    v = OP(v, HX_DPP(ident, v, 0x142, 0xa));    /* row_bcast:15 into rows 1 and 3 */
#define HX_SCAN_WAVE(v, ident, OP)                                                    \
    HX_SCAN_ROWS(v, ident, OP)                                                        \
    v = OP(v, HX_DPP(ident, v, 0x143, 0xc));    /* row_bcast:31 into rows 2 and 3 */
#define HX_OP_ADD(a, b) ((a) + (b))
#define HX_OP_MAX(a, b) max((a), (b))
#define HX_OP_OR(a, b) ((a) | (b))

__device__ __forceinline__ int hx_wave_sum(int v)
{
    HX_SCAN_WAVE(v, 0, HX_OP_ADD)
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int hx_wave_max(int v)
{
    HX_SCAN_WAVE(v, (int) 0x80000000, HX_OP_MAX)
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int hx_wave_or(int v)
{
    HX_SCAN_WAVE(v, 0, HX_OP_OR)
    return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int hx_wave_scan(int v)
{
    HX_SCAN_WAVE(v, 0, HX_OP_ADD)
    return v;
}
// reductions over each 32-lane half of the wave (result in every lane of the half)
__device__ __forceinline__ int hx_half_sum(int v)
{
    HX_SCAN_ROWS(v, 0, HX_OP_ADD)
    const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ int hx_half_max(int v)
{
    HX_SCAN_ROWS(v, (int) 0x80000000, HX_OP_MAX)
    const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ int hx_half_or(int v)
{
    HX_SCAN_ROWS(v, 0, HX_OP_OR)
    const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
