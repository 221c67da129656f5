/* cert_sums_check.c - CPU property test of the stream walk's certified band sums (hmp3_amd/csrc/hx_dev.h, "certified band
 * sums"; hx_alloc.hip noise_sweep / inverse_sf2; hx_front.hip msmetric_unit).  Test infrastructure: restates, in plain C and fp32, the
 * reduction the kernels run - a lane adds its run of at most W terms (both tree shapes the kernels use), a segmented
 * Hillis-Steele scan over the band's lanes (row_shr 1/2/4/8 inside 16-lane rows, row_bcast:15 / row_bcast:31 across a row
 * boundary) - and checks, for random and adversarial vectors of non-negative terms, that the reference's strict left-to-right
 * fp32 sum (l3math.c:521-537) lies inside the interval the kernels certify with, and that a certified bucket is the strict
 * sum's bucket of mbLogC (l3math.c:228-242: exponent and top 8 mantissa bits).  The same for the quotient of two sums and for
 * the stereo metric's sums (hx_front.hip msmetric_unit).
 *
 *   cert_sums_check <vectors> <seed>      exit 0 = every vector inside its interval; prints the straddle rate
 * Build: gcc -O2 -ffp-contract=off -o cert_sums_check cert_sums_check.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng_s;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static double urand(void) { return (double) (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* hx_cert_delta / hx_cert_mblog / hx_cert_mblog_ratio of hx_dev.h, operation for operation */
static float cert_delta(int n, int W) { return (float) (n + W + 16) * 5.9604644775390625e-08f; }
static int cert_mblog(float t, float du, float *lo, float *hi)
{
    volatile float e = t * du;
    volatile float a = t - e, b = t + e;
    *lo = a; *hi = b;
    volatile float ya = 1.0e-12f + a, yb = 1.0e-12f + b;
    return (f2u(ya) >> 15) == (f2u(yb) >> 15);
}

/* the lanes' partial sums: tree of pairs over at most 10 terms (sweep_run) or pair after pair (sweep_run_stored) */
static float lane_sum(const float *t, int cnt, int W, int shape)
{
    float v[10] = {0};
    for (int k = 0; k < cnt; k++) v[k] = t[k];
    if (shape == 0) {
        volatile float p01 = v[0] + v[1], p23 = v[2] + v[3], p45 = v[4] + v[5], p67 = v[6] + v[7], p89 = v[8] + v[9];
        volatile float q0 = p01 + p23, q1 = p45 + p67;
        volatile float r = q0 + q1;
        volatile float s = r + p89;
        return s;
    }
    volatile float acc = 0.0f;
    for (int k = 0; k < W; k += 2) { volatile float p = v[k] + v[k + 1]; acc = acc + p; }
    return acc;
}

/* hx_seg_scan on 64 lanes: v = the lanes' partial sums, d = lanes back to the segment's first lane */
static void seg_scan(float *v, const int *d)
{
    static const int sh[4] = {1, 2, 4, 8};
    float t[64];
    for (int s = 0; s < 4; s++) {
        for (int l = 0; l < 64; l++) t[l] = ((l & 15) >= sh[s]) ? v[l - sh[s]] : 0.0f;      /* row_shr: nothing crosses a row */
        for (int l = 0; l < 64; l++) { volatile float x = v[l] + ((d[l] >= sh[s]) ? t[l] : 0.0f); v[l] = x; }
    }
    for (int l = 0; l < 64; l++) t[l] = ((l >> 4) == 1 || (l >> 4) == 3) ? v[(l & ~15) - 1] : 0.0f;     /* row_bcast:15, rows 1 and 3 */
    for (int l = 0; l < 64; l++) { volatile float x = v[l] + ((d[l] > (l & 15)) ? t[l] : 0.0f); v[l] = x; }
    for (int l = 0; l < 64; l++) t[l] = ((l >> 4) == 2) ? v[31] : 0.0f;                                 /* row_bcast:31, row 2 */
    for (int l = 0; l < 64; l++) { volatile float x = v[l] + ((d[l] > (l & 15)) ? t[l] : 0.0f); v[l] = x; }
}

static float strict_sum(const float *t, int n)
{
    volatile float s = 0.0f;
    for (int i = 0; i < n; i++) s = s + t[i];
    return s;
}

/* the band's total as the kernels form it: n terms in runs of W starting at lane l0, other lanes hold other bands' (random) sums */
static float tree_sum(const float *t, int n, int W, int l0, int shape)
{
    float v[64];
    int d[64];
    const int c = (n + W - 1) / W;
    for (int l = 0; l < 64; l++) { v[l] = (float) urand() * 1.0e6f; d[l] = 0; }
    for (int k = 0; k < c; k++) {
        const int cnt = (n - k * W < W) ? n - k * W : W;
        v[l0 + k] = lane_sum(t + k * W, cnt, W, shape);
        d[l0 + k] = k;
    }
    /* the lanes around the band belong to other segments: give them plausible d */
    for (int l = l0 + c, k = 0; l < 64; l++, k++) d[l] = k & 3;
    for (int l = l0 - 1, k = 0; l >= 0; l--, k++) d[l] = 0;
    seg_scan(v, d);
    return v[l0 + c - 1];
}

static float rand_term(int family, int i, int n, float scale)
{
    switch (family) {
    case 0: return scale * (float) urand();                                                /* uniform */
    case 1: return u2f((uint32_t) (rnd() % 0x7F000000u)) * 1.0e-10f;                        /* any exponent, wide dynamic range */
    case 2: return (i == 0 ? scale : scale * 5.9604645e-08f * (1.0f + 9.765625e-4f));       /* a large first term, the rest just above half an ulp of it: every strict add rounds up */
    case 3: return (i == 0 ? scale : scale * 2.9802322e-08f);                               /* ... just below: every strict add is lost */
    case 4: return (i == n - 1 ? scale : scale * 5.9604645e-08f * (1.0f + 9.765625e-4f));   /* the large term last */
    case 5: return u2f((uint32_t) (rnd() % 0x00FFFFFFu));                                   /* subnormals and the smallest normals */
    case 6: return scale * (1.0f + (float) (rnd() & 3) * 1.1920929e-07f);                   /* near-equal terms: ties and carries */
    case 7: return (rnd() & 3) ? 0.0f : scale * (float) urand();                            /* mostly zeros */
    default: { const double e = urand() * 40.0 - 20.0; return (float) (pow(10.0, e)) * (float) urand(); }    /* log-uniform */
    }
}

/* ---- the stereo metric's band sums (hx_front.hip msmetric_unit; reference bitallo3.cpp:695-742): el = 100 + sum l^2, er = 100 + sum r^2
 * and the signed t = sum l r feed four mbLogC arguments only: el + er, max(el, er), es + ed and max(es, ed) with es = (el + er) + 2 t,
 * ed = (el + er) - 2 t.  The kernels certify all four buckets from tree sums and intervals (the signed sum's half-width from
 * sum |l r|) and run the strict loop otherwise.  Returns 0 = certified and equal to the strict buckets, 1 = not certified, 2 = WRONG. */
static unsigned bucket(float x) { return f2u(x) >> 15; }
static int metric_case(const float *l, const float *r, int n, int W, int l0, int shape)
{
    static float a[192], b[192], c[192], m[192];
    volatile float el = 100.0f, er = 100.0f, t = 0.0f;
    for (int k = 0; k < n; k++) {
        volatile float x = l[k] * l[k], y = r[k] * r[k], z = l[k] * r[k];
        a[k] = x; b[k] = y; c[k] = z; m[k] = fabsf(z);
        el = el + x; er = er + y; t = t + z;
    }
    volatile float es = el + er, ed = es;
    volatile float t2 = t + t;
    es = es + t2; ed = ed - t2;
    volatile float p1 = el + er, p3 = es + ed;
    const float p2 = el > er ? el : er, p4 = es > ed ? es : ed;
    /* the kernels' side: tree sums of the four term vectors */
    const float SA = tree_sum(a, n, W, l0, shape), SB = tree_sum(b, n, W, l0, shape), SM = tree_sum(m, n, W, l0, shape);
    float SC;
    {   /* the signed sum through the same tree (tree_sum works on any floats) */
        SC = tree_sum(c, n, W, l0, shape);
    }
    const float du = cert_delta(n + 1, W);       /* (the 100 in front is one more term, one more addition) */
    volatile float tel = 100.0f + SA, ter = 100.0f + SB;
    volatile float e1 = tel * du, e2 = ter * du, e3 = SM * du;
    volatile float el_lo = tel - e1, el_hi = tel + e1, er_lo = ter - e2, er_hi = ter + e2;
    /* (a sum of non-negative terms that starts at 100 never falls below 100: rounding is monotone) */
    if (el_lo < 100.0f) el_lo = 100.0f;
    if (er_lo < 100.0f) er_lo = 100.0f;
    volatile float t_lo = SC - e3, t_hi = SC + e3;
    volatile float tl2 = t_lo + t_lo, th2 = t_hi + t_hi;
    volatile float p1_lo = el_lo + er_lo, p1_hi = el_hi + er_hi;
    const float p2_lo = el_lo > er_lo ? el_lo : er_lo, p2_hi = el_hi > er_hi ? el_hi : er_hi;
    volatile float es_lo = p1_lo + tl2, es_hi = p1_hi + th2, ed_lo = p1_lo - th2, ed_hi = p1_hi - tl2;
    volatile float p3_lo = es_lo + ed_lo, p3_hi = es_hi + ed_hi;
    float p4_lo = es_lo > ed_lo ? es_lo : ed_lo;
    const float p4_hi = es_hi > ed_hi ? es_hi : ed_hi;
    /* (one of es, ed is p1 plus something non-negative, rounded: max(es, ed) >= p1) */
    if (p4_lo < p1_lo) p4_lo = p1_lo;
    const int ok = p3_lo > 0.0f && p4_lo > 0.0f && bucket(p1_lo) == bucket(p1_hi) && bucket(p2_lo) == bucket(p2_hi) && bucket(p3_lo) == bucket(p3_hi) && bucket(p4_lo) == bucket(p4_hi);
    /* the enclosures themselves must hold whether or not they certify */
    if (!(el_lo <= el && el <= el_hi && er_lo <= er && er <= er_hi && t_lo <= t && t <= t_hi)) return 2;
    if (!ok) return 1;
    if (bucket(p1) != bucket(p1_lo) || bucket(p2) != bucket(p2_lo) || bucket(p3) != bucket(p3_lo) || bucket(p4) != bucket(p4_lo)) return 2;
    return 0;
}

int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 1000000;
    rng_s = argc > 2 ? strtoull(argv[2], 0, 10) * 2654435761u + 88172645463325252ull : 88172645463325252ull;
    long bad = 0, straddle = 0, evals = 0, badq = 0, straddleq = 0, badm = 0, straddlem = 0, evalm = 0;
    static float t[192], q[192];
    for (long it = 0; it < N; it++) {
        const int W = 2 + 2 * (int) (rnd() % 5);                       /* 2 .. 10 */
        int n = 2 * (1 + (int) (rnd() % (8 * W < 96 ? 8 * W : 96)));    /* even, at most 16 lanes and 192 terms (the widest band is 192) */
        if (n > 16 * W) n = 16 * W;
        if (n > 192) n = 192;
        const int c = (n + W - 1) / W;
        const int l0 = (int) (rnd() % (65 - c));
        const int family = (int) (rnd() % 9), shape = (int) (rnd() & 1);
        const float scale = (float) pow(10.0, urand() * 24.0 - 12.0);
        for (int i = 0; i < n; i++) { t[i] = rand_term(family, i, n, scale); q[i] = rand_term((int) (rnd() % 9), i, n, scale * 3.0f); }
        /* (the noise terms are squares: non-negative; make sure) */
        for (int i = 0; i < n; i++) { t[i] = fabsf(t[i]); q[i] = fabsf(q[i]); if (!(t[i] < 1.0e30f)) t[i] = 1.0e30f; if (!(q[i] < 1.0e30f)) q[i] = 1.0e30f; }
        const float s = strict_sum(t, n), tt = tree_sum(t, n, W, l0, shape);
        const float du = cert_delta(n, W);
        float lo, hi;
        const int ok = cert_mblog(tt, du, &lo, &hi);
        evals++;
        if (!(lo <= s && s <= hi)) {
            if (bad < 10) fprintf(stderr, "outside: n %d W %d family %d strict %a tree %a [%a, %a]\n", n, W, family, s, tt, lo, hi);
            bad++;
        }
        if (!ok) straddle++;
        else {
            volatile float ys = 1.0e-12f + s, yl = 1.0e-12f + lo;
            if ((f2u(ys) >> 15) != (f2u(yl) >> 15)) { if (bad < 10) fprintf(stderr, "certified bucket differs: n %d W %d family %d\n", n, W, family); bad++; }
        }
        /* quotient of two sums (inverse_sf2: mbLogC(sxx / sqq)) */
        const float sq = strict_sum(q, n), tq = tree_sum(q, n, W, l0, shape);
        if (sq > 0.0f && tq > 0.0f) {
            volatile float ex = tt * du, eq = tq * du;
            volatile float xl = tt - ex, xh = tt + ex, ql = tq - eq, qh = tq + eq;
            volatile float rl = xl / qh, rh = xh / ql, rs = s / sq;
            if (!(rl <= rs && rs <= rh)) { if (badq < 10) fprintf(stderr, "quotient outside: n %d W %d strict %a [%a, %a]\n", n, W, rs, rl, rh); badq++; }
            if ((f2u(rl) >> 15) != (f2u(rh) >> 15)) straddleq++;
            else if ((f2u(rs) >> 15) != (f2u(rl) >> 15)) { if (badq < 10) fprintf(stderr, "certified quotient bucket differs\n"); badq++; }
        }
    }
    /* stereo metric: correlated channel pairs of every kind */
    long kindm[8] = {0}, kindn[8] = {0};
    for (long it = 0; it < N / 2; it++) {
        static float l[192], r[192];
        const int W = 2 + 2 * (int) (rnd() % 5);
        int n = 2 * (1 + (int) (rnd() % 96));
        if (n > 16 * W) n = 16 * W;
        const int c = (n + W - 1) / W, l0 = (int) (rnd() % (65 - c)), shape = (int) (rnd() & 1);
        const int kind = (int) (rnd() % 8);
        const float scale = (float) pow(10.0, urand() * 9.0 - 3.0);
        const double rho = kind == 0 ? 1.0 : kind == 1 ? -1.0 : kind == 2 ? 0.0 : kind == 3 ? 0.999 : kind == 4 ? -0.999 : urand() * 2.0 - 1.0;
        for (int k = 0; k < n; k++) {
            const double u = urand() * 2.0 - 1.0, v = urand() * 2.0 - 1.0;
            double a = u, b = rho * u + (1.0 - fabs(rho)) * v;
            if (kind == 6) { a = (k & 1) ? 0.0 : u; b = (k & 1) ? v : 0.0; }       /* disjoint support */
            if (kind == 7) { a *= 1.0e-4; }                                         /* one channel nearly silent */
            l[k] = (float) (a * scale); r[k] = (float) (b * scale);
        }
        const int res = metric_case(l, r, n, W, l0, shape);
        evalm++;
        if (res == 2) { if (badm < 10) fprintf(stderr, "metric: wrong certificate or enclosure, n %d W %d kind %d\n", n, W, kind); badm++; }
        else if (res == 1) { straddlem++; kindm[kind]++; }
        kindn[kind]++;
    }
    for (int k = 0; k < 8; k++) printf("  metric kind %d: not certified %.2f %%\n", k, 100.0 * kindm[k] / (kindn[k] ? kindn[k] : 1));
    printf("metric bands %ld  wrong %ld  not certified %ld (%.3f %%)\n", evalm, badm, straddlem, 100.0 * straddlem / (evalm ? evalm : 1));
    printf("vectors %ld  outside %ld  straddles %ld (%.3f %%)  quotient outside %ld  quotient straddles %ld (%.3f %%)\n", evals, bad, straddle,
           100.0 * straddle / evals, badq, straddleq, 100.0 * straddleq / evals);
    return (bad || badq || badm) ? 1 : 0;
}
