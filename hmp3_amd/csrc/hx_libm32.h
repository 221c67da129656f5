// hx_libm32.h - single-precision logf / log10f with the results of GNU libc 2.35 (x86-64), for the first-generation
// allocator (hx_alloc1.inc).
//
// Why: the reference's bitallo1.cpp is C++, so its log10(float) / log(float) calls are libm's log10f / logf (checked in
// the object code), whose results are not correctly rounded: the encoder's output depends on the libm the reference is
// linked with.  The oracle and the reference run on glibc 2.35 (the image's libm.so.6); these functions restate that
// library's published algorithms so that the device agrees with them bit for bit:
//   logf   sysdeps/ieee754/flt-32/e_logf.c + logf_data.c (from ARM Optimized Routines): 16-entry table of
//          {1/c, log c}, a degree-3 polynomial in r = z/c - 1, everything in double, one rounding to float;
//   log10f sysdeps/ieee754/flt-32/e_log10f.c: exponent and mantissa split, log10(2) in two pieces,
//          z = y * log10_2lo + ivln10 * logf(m), result z + y * log10_2hi, in float.
// The table values were read out of the image's libm.so.6 (tools/capture_glibc_logf.py prints them) and are the
// constants of logf_data.c.  tests/test_host_and_abi.py compares the host build of this header with libm on a large
// sample; tools/check_libm32.c does it for every float.  (glibc selects an FMA build of logf on CPUs that have the
// instruction; the double result then differs in its last bits, which reaches the float result for no input that
// the exhaustive run found.)  Only positive finite normal arguments occur here (band energies and maxima
// > 1e-12); zero, negative, infinite and NaN arguments follow the same branches as glibc for completeness.
#pragma once
#include <stdint.h>
#include <string.h>
#ifndef HX_HD
#define HX_HD __host__ __device__ __forceinline__
#endif

HX_HD float hx_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
HX_HD uint32_t hx_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

HX_HD float hx_logf(float x)
{
    // {invc, logc} of the sixteen subintervals of [0x3f330000, 2 * 0x3f330000) as IEEE-754 bit patterns
    const uint64_t T[16][2] = {
        {0x3ff661ec79f8f3beull, 0xbfd57bf7808caadeull}, {0x3ff571ed4aaf883dull, 0xbfd2bef0a7c06ddbull},
        {0x3ff49539f0f010b0ull, 0xbfd01eae7f513a67ull}, {0x3ff3c995b0b80385ull, 0xbfcb31d8a68224e9ull},
        {0x3ff30d190c8864a5ull, 0xbfc6574f0ac07758ull}, {0x3ff25e227b0b8ea0ull, 0xbfc1aa2bc79c8100ull},
        {0x3ff1bb4a4a1a343full, 0xbfba4e76ce8c0e5eull}, {0x3ff12358f08ae5baull, 0xbfb1973c5a611cccull},
        {0x3ff0953f419900a7ull, 0xbfa252f438e10c1eull}, {0x3ff0000000000000ull, 0x0000000000000000ull},
        {0x3fee608cfd9a47acull, 0x3faaa5aa5df25984ull}, {0x3feca4b31f026aa0ull, 0x3fbc5e53aa362eb4ull},
        {0x3feb2036576afce6ull, 0x3fc526e57720db08ull}, {0x3fe9c2d163a1aa2dull, 0x3fcbc2860d224770ull},
        {0x3fe886e6037841edull, 0x3fd1058bc8a07ee1ull}, {0x3fe767dcf5534862ull, 0x3fd4043057b6ee09ull}};
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = hx_f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -1.0f / 0.0f;                   // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                        // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / 0.0f;     // log(negative) or NaN: NaN
        ix = hx_f2u(x * 0x1p23f);                               // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int) ((tmp >> (23 - 4)) % 16);
    const int k = (int32_t) tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    double invc, logc;
    memcpy(&invc, &T[i][0], 8);
    memcpy(&logc, &T[i][1], 8);
    const double z = (double) hx_u2f(iz);
    const double r = z * invc - 1;
    const double y0 = logc + (double) k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float) y;
}

HX_HD float hx_log10f(float x)
{
    const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
    int32_t hx = (int32_t) hx_f2u(x), k = 0;
    if (hx < 0x00800000) {
        if ((hx & 0x7fffffff) == 0) return -two25 / 0.0f;       // log10(+-0) = -inf
        if (hx < 0) return (x - x) / (x - x);                   // log10(negative) = NaN
        k -= 25; x *= two25;                                    // subnormal: scale up
        hx = (int32_t) hx_f2u(x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    const int32_t i = (int32_t) (((uint32_t) k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float) (k + i);
    const float m = hx_u2f((uint32_t) hx);
    const float z = y * log10_2lo + ivln10 * hx_logf(m);
    return z + y * log10_2hi;
}
