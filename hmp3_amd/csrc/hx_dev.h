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

__device__ __forceinline__ int hx_wave_sum(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ int hx_wave_max(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int hx_wave_or(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m, 64);
    return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int hx_wave_scan(int v)
{
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    return v;
}
