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

// the same with the table as 16-bit values without its constant term (the allocator's low-footprint LDS layout)
__device__ __forceinline__ int hx_mblog16(const unsigned short *tab, float x)
{
    unsigned u = hx_f2bits(x);
    return (int) tab[(u >> 15) & 255] + 301 * (int) (u >> 23) - 38227;
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

// x^(3/4): piecewise-linear mantissa fit x exponent table (reference pow34.c:132-154)
__device__ __forceinline__ float hx_pow34(const float *a, const float *b, const float *ex, float x)
{
    unsigned u = hx_f2bits(x);
    float m = hx_bits2f((u & 0x7FFFFFu) | (127u << 23));
    unsigned seg = (u >> 19) & 15, e = (u >> 23) & 255;
    return (m * b[seg] + a[seg]) * ex[e];
}

// ---- certified band sums -------------------------------------------------------------------------------------
// The reference adds a band's non-negative terms t_0 .. t_(n-1) left to right in fp32 (l3math.c:521-537) and hands the
// sum s to mbLogC (l3math.c:228-242), which reads only its exponent and top 8 mantissa bits.  Any other fp32 summation
// order t of the same terms, every term passing through at most D additions, satisfies
//     |s - S| <= ((1 + u)^(n-1) - 1) S,   |t - S| <= ((1 + u)^D - 1) S      (S the exact sum, u = 2^-24; no underflow
// in fp32 addition), hence s lies in [t (1 - c u), t (1 + c u)] for c = n + D + slack.  1.0e-12f + x and the bucket
// (bits >> 15) are monotone in x: if both ends of the interval land in the same bucket, the reference's mbLogC value is
// proven without forming s; otherwise the band's lane adds the terms in line order as before.  The same holds for the
// quotient of two such sums (inverse_sf2) with the interval of the quotient.  tests/test_cert_sums.py checks the bound on
// 10^6 random and adversarial term vectors against the strict sum; HMP3AMD_EXACT_SUMS=1 forces the strict sum everywhere.
//
// Layout: the 64 lanes each own a run of at most W consecutive lines of one band (HxParams::lane_run), add their own terms,
// and a segmented inclusive scan over the neighbouring lanes of a band (at most 16, d = lanes back to the band's first one)
// leaves the band's total in its last lane: row_shr 1/2/4/8 inside the 16-lane rows, then the previous row's last lane
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into row 2) for a band that straddles a row boundary.
// Depth of a term: (W - 1) + 6 additions.
#define HX_DPPF(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ float hx_seg_scan(float v, int d, int lane)
{
    float t;
    t = HX_DPPF(v, 0x111, 0xf); v += (d >= 1) ? t : 0.0f;
    t = HX_DPPF(v, 0x112, 0xf); v += (d >= 2) ? t : 0.0f;
    t = HX_DPPF(v, 0x114, 0xf); v += (d >= 4) ? t : 0.0f;
    t = HX_DPPF(v, 0x118, 0xf); v += (d >= 8) ? t : 0.0f;
    const bool cross = d > (lane & 15);         // the band's first lane is in the row before this one
    t = HX_DPPF(v, 0x142, 0xa); v += cross ? t : 0.0f;
    t = HX_DPPF(v, 0x143, 0x4); v += cross ? t : 0.0f;
    return v;
}
// the value of lane src4 / 4 (LDS crossbar, no memory access)
__device__ __forceinline__ float hx_lane_read(int src4, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v))); }
// c u for a band of n terms summed in runs of W: n - 1 + (W - 1 + 6) roundings, + 12 for the second-order terms and the
// roundings of the interval's own arithmetic
__device__ __forceinline__ float hx_cert_delta(int n, int W) { return (float) (n + W + 16) * 5.9604644775390625e-08f; }
// do both ends of [t (1 - du), t (1 + du)], after the reference's + 1.0e-12f, fall into one mbLogC bucket?
__device__ __forceinline__ bool hx_cert_mblog(float t, float du)
{
    const float e = t * du;
    return (hx_f2bits(1.0e-12f + (t - e)) >> 15) == (hx_f2bits(1.0e-12f + (t + e)) >> 15);
}
// the same for mbLogC(x / q) of two certified sums (IEEE division: monotone in both arguments)
__device__ __forceinline__ bool hx_cert_mblog_ratio(float x, float q, float du)
{
    const float ex = x * du, eq = q * du;
    return (hx_f2bits((x - ex) / (q + eq)) >> 15) == (hx_f2bits((x + ex) / (q - eq)) >> 15);
}

// sequential (reference-order) sum of term[ch][start .. start+n)
// Every scalefactor band starts on an even line and has an even width (ISO Table B.8), so the
// terms are fetched as 8-byte pairs, four pairs in flight, and added strictly in line order.
__device__ __forceinline__ float band_sum(const float *t, int n, float acc)
{
    // Blocks of eight pairs, software pipelined (the next block's loads are in flight while the
    // sixteen dependent adds of this one retire), then tails of 4 / 2 / 1 pairs.  Lanes with a
    // narrower band simply drop out of the loops (exec mask), no per-element predicates.
    const float2 *t2 = reinterpret_cast<const float2 *>(t);
    const int m = n >> 1;
    int j = 0;
    if (m >= 8) {
        float2 c[8], nx[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = t2[k];
        for (j = 8; j + 8 <= m; j += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) nx[k] = t2[j + k];
#pragma unroll
            for (int k = 0; k < 8; k++) { acc += c[k].x; acc += c[k].y; }
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = nx[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += c[k].x; acc += c[k].y; }
    }
    if (j + 4 <= m) {
        float2 a = t2[j], b = t2[j + 1], c = t2[j + 2], d = t2[j + 3];
        acc += a.x; acc += a.y; acc += b.x; acc += b.y;
        acc += c.x; acc += c.y; acc += d.x; acc += d.y;
        j += 4;
    }
    if (j + 2 <= m) {
        float2 a = t2[j], b = t2[j + 1];
        acc += a.x; acc += a.y; acc += b.x; acc += b.y;
        j += 2;
    }
    if (j < m) { float2 a = t2[j]; acc += a.x; acc += a.y; }
    return acc;
}
// two independent sums over the same band of two term arrays (same order each), pipelined like
// band_sum; the two add chains interleave
__device__ __forceinline__ void band_sum2(const float *t, const float *u, int n, float *s0, float *s1)
{
    const float2 *t2 = reinterpret_cast<const float2 *>(t), *u2 = reinterpret_cast<const float2 *>(u);
    const int m = n >> 1;
    float a0 = 0.0f, a1 = 0.0f;
    int j = 0;
    if (m >= 8) {
        float2 c[8], d[8], nc[8], nd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { c[k] = t2[k]; d[k] = u2[k]; }
        for (j = 8; j + 8 <= m; j += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) { nc[k] = t2[j + k]; nd[k] = u2[j + k]; }
#pragma unroll
            for (int k = 0; k < 8; k++) { a0 += c[k].x; a1 += d[k].x; a0 += c[k].y; a1 += d[k].y; }
#pragma unroll
            for (int k = 0; k < 8; k++) { c[k] = nc[k]; d[k] = nd[k]; }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { a0 += c[k].x; a1 += d[k].x; a0 += c[k].y; a1 += d[k].y; }
    }
    if (j + 4 <= m) {
        float2 c[4], d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { c[k] = t2[j + k]; d[k] = u2[j + k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) { a0 += c[k].x; a1 += d[k].x; a0 += c[k].y; a1 += d[k].y; }
        j += 4;
    }
    if (j + 2 <= m) {
        float2 c0 = t2[j], c1 = t2[j + 1], d0 = u2[j], d1 = u2[j + 1];
        a0 += c0.x; a1 += d0.x; a0 += c0.y; a1 += d0.y;
        a0 += c1.x; a1 += d1.x; a0 += c1.y; a1 += d1.y;
        j += 2;
    }
    if (j < m) { float2 c0 = t2[j], d0 = u2[j]; a0 += c0.x; a1 += d0.x; a0 += c0.y; a1 += d0.y; }
    *s0 = a0;
    *s1 = a1;
}

