/* hxo_front.c - ORACLE (test infrastructure): polyphase filterbank, hybrid MDCT, alias
 * reduction, transient detector, MDCT-energy psychoacoustic model.
 * Restates sbt.c:57-310, hwin.c:147-322, emdct.c:104-303, detect.c:53-142, emap.c:61-121,
 * spdsmr.c:64-320 with the reference's evaluation order (bit-exact). */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "hxo_int.h"

/* ------------------------------------------------------------------------------------------
 * 32-band analysis for one time slot.  v[0] is the newest of 512 samples.
 * Windowing (sbt.c:57-109): output k folds the taps at ages A(k)+64j and B(k)+64j, each
 * partial sum accumulated over j = 0..7 in order, then added.
 * DCT (sbt.c:134-259): four "split" passes (even inputs pass through, odd inputs become a
 * running difference from the top) followed by five butterfly passes.
 */
static void dct32_split(int m, int n, const float *x, float *f)
{
    int blk, i, h = n / 2;
    for (blk = 0; blk < m; blk++, x += n, f += n) {
        f[h + h - 1] = x[n - 1];
        f[h - 1] = x[n - 2];
        for (i = h - 2; i >= 0; i--) {
            f[h + i] = x[2 * i + 1] - f[h + i + 1];
            f[i] = x[2 * i];
        }
    }
}

static void dct32_bfly(int m, int n, const float *x, float *f, const float *c)
{
    int blk, j, h = n / 2;
    for (blk = 0; blk < m; blk++, x += n, f += n)
        for (j = 0; j < h; j++) {
            float tmp = c[j] * x[j + h];
            float t = x[j];
            f[j] = t + tmp;
            f[n - 1 - j] = t - tmp;
        }
}

static void analysis_slot(const hxo_params *p, const float *v, float *out, int stride)
{
    float a[32], b[32], s1, s2;
    int k, j;
    s1 = 0.0f;
    for (j = 0; j < 8; j++) s1 += hxo_anwin(16 + 64 * j) * v[16 + 64 * j];
    b[0] = s1;
    for (k = 1; k < 32; k++) {
        int A = (k <= 16) ? 16 + k : 80 - k;
        int B = (k <= 16) ? 16 - k : 16 + k;
        s1 = s2 = 0.0f;
        for (j = 0; j < 8; j++) {
            s1 += hxo_anwin(A + 64 * j) * v[A + 64 * j];
            s2 += hxo_anwin(B + 64 * j) * v[B + 64 * j];
        }
        b[k] = s1 + s2;
    }
    dct32_split(1, 32, b, a);
    dct32_split(2, 16, a, b);
    dct32_split(4, 8, b, a);
    dct32_split(8, 4, a, b);
    dct32_bfly(16, 2, b, a, p->dct_coef + 30);
    dct32_bfly(8, 4, a, b, p->dct_coef + 28);
    dct32_bfly(4, 8, b, a, p->dct_coef + 24);
    dct32_bfly(2, 16, a, b, p->dct_coef + 16);
    for (k = 0; k < 16; k++) {              /* last pass writes subband-major (sbt.c:205-221) */
        float tmp = p->dct_coef[k] * b[k + 16];
        out[stride * k] = b[k] + tmp;
        out[stride * (31 - k)] = b[k] - tmp;
    }
}

/* sbt.c:293-310: 18 slots of one granule; vbuf[0] = newest sample of the granule */
void hxo_polyphase_granule(const hxo_params *p, const float *vbuf, float *samp)
{
    int t;
    for (t = 0; t < 18; t++) analysis_slot(p, vbuf + 576 - 32 * (t + 1), samp + t, 18);
}

/* hwin.c:282-294 */
void hxo_freq_invert(float *y, int nsb)
{
    int i, j;
    for (j = 0; j < nsb; j += 2)
        for (i = 0; i < 18; i += 2) y[(1 + j) * 18 + 1 + i] = -y[(1 + j) * 18 + 1 + i];
}

/* emdct.c:104-188: 18-point cosine transform, in -> 18 folded values */
static void mdct18(const hxo_params *p, const float *f, float *y)
{
    const float *w = p->m18_w, *w2 = p->m18_w2;
    const float (*c)[4] = p->m18_c;
    float a[9], b[9], g1, g2, ap, bp, a8p, b8p;
    int q;
    for (q = 0; q < 4; q++) {
        g1 = w[q] * f[q];
        g2 = w[17 - q] * f[17 - q];
        ap = g1 + g2;
        bp = w2[q] * (g1 - g2);
        g1 = w[8 - q] * f[8 - q];
        g2 = w[9 + q] * f[9 + q];
        a8p = g1 + g2;
        b8p = w2[8 - q] * (g1 - g2);
        a[q] = ap + a8p;
        a[5 + q] = ap - a8p;
        b[q] = bp + b8p;
        b[5 + q] = bp - b8p;
    }
    g1 = w[4] * f[4];
    g2 = w[13] * f[13];
    a[4] = g1 + g2;
    b[4] = w2[4] * (g1 - g2);

    y[0] = 0.5f * (a[0] + a[1] + a[2] + a[3] + a[4]);
    y[1] = 0.5f * (b[0] + b[1] + b[2] + b[3] + b[4]);
    y[2] = c[1][0] * a[5] + c[1][1] * a[6] + c[1][2] * a[7] + c[1][3] * a[8];
    y[3] = c[1][0] * b[5] + c[1][1] * b[6] + c[1][2] * b[7] + c[1][3] * b[8] - y[1];
    y[1] = y[1] - y[0];
    y[2] = y[2] - y[1];
    y[4] = c[2][0] * a[0] + c[2][1] * a[1] + c[2][2] * a[2] + c[2][3] * a[3] - a[4];
    y[5] = c[2][0] * b[0] + c[2][1] * b[1] + c[2][2] * b[2] + c[2][3] * b[3] - b[4] - y[3];
    y[3] = y[3] - y[2];
    y[4] = y[4] - y[3];
    y[6] = c[3][0] * (a[5] - a[7] - a[8]);
    y[7] = c[3][0] * (b[5] - b[7] - b[8]) - y[5];
    y[5] = y[5] - y[4];
    y[6] = y[6] - y[5];
    y[8] = c[4][0] * a[0] + c[4][1] * a[1] + c[4][2] * a[2] + c[4][3] * a[3] + a[4];
    y[9] = c[4][0] * b[0] + c[4][1] * b[1] + c[4][2] * b[2] + c[4][3] * b[3] + b[4] - y[7];
    y[7] = y[7] - y[6];
    y[8] = y[8] - y[7];
    y[10] = c[5][0] * a[5] + c[5][1] * a[6] + c[5][2] * a[7] + c[5][3] * a[8];
    y[11] = c[5][0] * b[5] + c[5][1] * b[6] + c[5][2] * b[7] + c[5][3] * b[8] - y[9];
    y[9] = y[9] - y[8];
    y[10] = y[10] - y[9];
    y[12] = 0.5f * (a[0] + a[2] + a[3]) - a[1] - a[4];
    y[13] = 0.5f * (b[0] + b[2] + b[3]) - b[1] - b[4] - y[11];
    y[11] = y[11] - y[10];
    y[12] = y[12] - y[11];
    y[14] = c[7][0] * a[5] + c[7][1] * a[6] + c[7][2] * a[7] + c[7][3] * a[8];
    y[15] = c[7][0] * b[5] + c[7][1] * b[6] + c[7][2] * b[7] + c[7][3] * b[8] - y[13];
    y[13] = y[13] - y[12];
    y[14] = y[14] - y[13];
    y[16] = c[8][0] * a[0] + c[8][1] * a[1] + c[8][2] * a[2] + c[8][3] * a[3] + a[4];
    y[17] = c[8][0] * b[0] + c[8][1] * b[1] + c[8][2] * b[2] + c[8][3] * b[3] + b[4] - y[15];
    y[15] = y[15] - y[14];
    y[16] = y[16] - y[15];
    y[17] = y[17] - y[16];
}

/* hwin.c:147-181 */
void hxo_hybrid_long(const hxo_params *p, const float *x1, const float *x2, float *yout,
                     int btype, int nlong, int clear_flag)
{
    const float *w = p->win[btype];
    float y[18];
    int i, j;
    for (i = 0; i < nlong; i++) {
        for (j = 0; j < 9; j++) {
            y[j] = w[26 - j] * x2[8 - j] + w[27 + j] * x2[9 + j];
            y[9 + j] = w[j] * x1[j] + w[17 - j] * x1[17 - j];
        }
        mdct18(p, y, yout);
        x1 += 18; x2 += 18; yout += 18;
    }
    if (clear_flag) memset(yout, 0, sizeof(float) * 18 * (32 - nlong));
}

/* emdct.c:252-303: three 6-point transforms, window w of the triple goes to c[192*w] */
static void mdct6x3(const hxo_params *p, const float *f, float *c)
{
    const float *v = p->m6_v, *v2 = p->m6_v2;
    float buf[18], *a = buf, g1, g2, a02, b02;
    int w, q;
    for (w = 0; w < 3; w++, a += 6, f += 6)
        for (q = 0; q < 3; q++) {
            g1 = v[q] * f[q];
            g2 = v[5 - q] * f[5 - q];
            a[q] = g1 + g2;
            a[3 + q] = v2[q] * (g1 - g2);
        }
    a = buf;
    for (w = 0; w < 3; w++, a += 6, c += 192) {
        a02 = (a[0] + a[2]);
        b02 = (a[3] + a[5]);
        c[0] = a02 + a[1];
        c[1] = b02 + a[4];
        c[2] = p->m6_c87 * (a[0] - a[2]);
        c[3] = p->m6_c87 * (a[3] - a[5]) - c[1];
        c[1] = c[1] - c[0];
        c[2] = c[2] - c[1];
        c[4] = a02 - a[1] - a[1];
        c[5] = b02 - a[4] - a[4] - c[3];
        c[3] = c[3] - c[2];
        c[4] = c[4] - c[3];
        c[5] = c[5] - c[4];
    }
}

/* hwin.c:228-278: short blocks, output [3 windows][192] */
void hxo_hybrid_short(const hxo_params *p, const float *x1, const float *x2, float *yout, int n)
{
    const float *w = p->win[2];
    float y[18];
    int i, q;
    for (i = 0; i < n; i++) {
        for (q = 0; q < 3; q++) {
            y[q] = w[8 - q] * x1[14 - q] + w[9 + q] * x1[15 + q];
            y[3 + q] = w[q] * x1[6 + q] + w[5 - q] * x1[11 - q];
            y[6 + q] = w[8 - q] * x2[2 - q] + w[9 + q] * x2[3 + q];
            y[9 + q] = w[q] * x1[12 + q] + w[5 - q] * x1[17 - q];
            y[12 + q] = w[8 - q] * x2[8 - q] + w[9 + q] * x2[9 + q];
            y[15 + q] = w[q] * x2[q] + w[5 - q] * x2[5 - q];
        }
        mdct6x3(p, y, yout);
        x1 += 18; x2 += 18; yout += 6;
    }
    for (i = 0; i < 6 * (32 - n); i++) {
        yout[i] = 0.0f;
        yout[i + 192] = 0.0f;
        yout[i + 2 * 192] = 0.0f;
    }
}

/* hwin.c:298-322 */
void hxo_antialias(const hxo_params *p, float *x, int n)
{
    int i, k;
    n--;
    for (k = 0; k < n; k++, x += 18)
        for (i = 0; i < 8; i++) {
            float a = x[17 - i], b = x[18 + i];
            x[17 - i] = a * p->csa[0][i] + b * p->csa[1][i];
            x[18 + i] = b * p->csa[0][i] - a * p->csa[1][i];
        }
    for (i = 0; i < 8; i++) x[17 - i] = x[17 - i] * p->csa[0][i];
}

/* detect.c:53-142: energy of subbands 4..17 per slot pair -> mB; attack = rise over the
   maximum of the six preceding values */
int hxo_attack_detect(const float *sample, int eng[32], int short_flag_prev)
{
    int i, j, k, m = 0;
    memmove(eng, eng + 9, 23 * sizeof(int));
    for (k = 0, j = 23; k < 9; k++, j++) {
        const float *y = sample + 18 * 4 + 2 * k;
        float sum = 7.0e4f, x;
        for (i = 0; i < 14; i++, y += 18) {
            x = y[0] * y[0]; sum += x;
            x = y[1] * y[1]; sum += x;
        }
        eng[j] = hxo_mblog(sum);
    }
    for (j = short_flag_prev ? 18 : 17; j < 29; j++) {
        int a0 = HXO_MAX(eng[j - 6], eng[j - 7]);
        int a1 = HXO_MAX(eng[j - 4], eng[j - 5]);
        int a2 = HXO_MAX(eng[j - 2], eng[j - 3]);
        int a;
        a1 = HXO_MAX(a1, a0);
        a = HXO_MAX(a1, a2);
        m = HXO_MAX(m, eng[j] - a);
    }
    return m;
}

/* detect.c:147-228, the MPEG-2 variant: subbands 8..27, rise over the four preceding values */
int hxo_attack_detect_lsf(const float *sample, int eng[32], int short_flag_prev)
{
    int i, j, k, m = 0;
    memmove(eng, eng + 9, 23 * sizeof(int));
    for (k = 0, j = 23; k < 9; k++, j++) {
        const float *y = sample + 18 * 8 + 2 * k;
        float sum = 7.0e4f, x;
        for (i = 0; i < 20; i++, y += 18) {
            x = y[0] * y[0]; sum += x;
            x = y[1] * y[1]; sum += x;
        }
        eng[j] = hxo_mblog(sum);
    }
    for (j = short_flag_prev ? 18 : 17; j < 29; j++) {
        int a1 = HXO_MAX(eng[j - 4], eng[j - 5]);
        int a2 = HXO_MAX(eng[j - 2], eng[j - 3]);
        int a = HXO_MAX(a1, a2);
        m = HXO_MAX(m, eng[j] - a);
    }
    return m;
}

/* test taps: when set, hxo_psy_long also stores etab and the unclamped thresholds */
float *hxo_tap_etab, *hxo_tap_thr;

/* emap.c:96-121 + spdsmr.c:188-320 */
void hxo_psy_long(const hxo_params *p, const float *xr, float *esave, hxo_sigmask *sm, int block_type)
{
    const hxo_psytab *pt = &p->psyL;
    const float *w = pt->w;
    float e[64], xtab[64], stab[64] = {0}, etab[64];
    int mbetab[64];
    int i, j, k, n, q, m, npart, npart2;
    int snr, snr0, totsnr, nsnr, d, d0, dm0, dm, snrvar, dv, itmp;
    const float alpha = 0.30f;

    for (i = 0, j = 0; j < pt->npart_e; j++) {
        float s = 0.0f;
        for (k = 0; k < pt->nsum[j]; k++, i++) s += xr[i] * xr[i];
        e[j] = s;
    }
    for (; j < 64; j++) e[j] = 0.0f;

    npart = pt->npart;
    npart2 = (npart + 1) & (~1);
    for (i = 0; i < npart2; i++) {
        float t = w[i] + e[i];
        int mbe;
        etab[i] = t;
        mbe = hxo_mblog(t);
        mbetab[i] = mbe;
        xtab[i] = hxo_mbexp((int) (alpha * mbe));
    }
    nsnr = snrvar = totsnr = snr0 = 0;
    k = 128;
    for (i = 0; i < npart; i++) {
        float s = 0.1f;
        q = pt->off[i];
        n = pt->cnt[i];
        for (j = 0; j < n; j++, k++) s += w[k] * xtab[q + j];
        s = (0.03f * 0.1f * 0.35f) * hxo_mbexp((int) ((1.0f / alpha) * hxo_mblog(s))) + w[i];
        stab[i] = s;
        snr = mbetab[i] - hxo_mblog(w[i] + s);
        if (snr > 0) nsnr++;
        totsnr += HXO_MAX(-200, snr);
        snrvar += abs(snr - snr0);
        snr0 = snr;
    }
    d = 0;
    if (nsnr > 0) {
        d0 = hxo_round(1.3f * (totsnr / npart) - 850);
        itmp = snrvar / npart;
        dv = HXO_MIN(500 - itmp, 0);
        d = d0 + dv;
        d = HXO_MAX(d, -2000);
        d = HXO_MIN(d, 600);
    }
    d += 300;
    dm0 = (300 - d) >> 4;
    for (m = 0, i = 0; i < npart; i += 2, m++) {
        float a, s1, s2, t, x, emax, s;
        dm = HXO_MAX(dm0 * HXO_MAX(m - 13, 0), 0);
        a = hxo_mbexp(d + dm);
        if (hxo_tap_etab) {
            hxo_tap_etab[i] = etab[i]; hxo_tap_etab[i + 1] = etab[i + 1];
            hxo_tap_thr[i] = a * stab[i]; hxo_tap_thr[i + 1] = a * stab[i + 1];
        }
        s1 = a * stab[i];
        t = esave[i];
        esave[i] = (float) (2.0 * s1);
        if (block_type != 3 && s1 > t) {    /* pre-echo control */
            x = 0.1f * s1;
            s1 = t;
            if (s1 < x) s1 = x;
        }
        s2 = a * stab[i + 1];
        t = esave[i + 1];
        esave[i + 1] = 2.0f * s2;
        if (block_type != 3 && s2 > t) {
            x = 0.1f * s2;
            s2 = t;
            if (s2 < x) s2 = x;
        }
        emax = etab[i];
        if (emax < etab[i + 1]) emax = etab[i + 1];
        s = (etab[i] * s1 + etab[i + 1] * s2) / emax;
        sm[m].sig = etab[i] + etab[i + 1];
        sm[m].mask = s;
    }
}

/* emap.c:61-93 + spdsmr.c:64-184; xr is [3][192], sm is [3][12] */
void hxo_psy_short(const hxo_params *p, const float *xr, float *esave, hxo_sigmask *sm, int block_type_prev)
{
    const hxo_psytab *pt = &p->psyS;
    const float *w = pt->w;
    float e[3][64], mask[3][12];
    int i, j, k, n, q, m, npart, mpart, wn;

    for (i = 0, j = 0; j < pt->npart_e; j++) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        for (k = 0; k < pt->nsum[j]; k++, i++) {
            s0 += xr[i] * xr[i];
            s1 += xr[192 + i] * xr[192 + i];
            s2 += xr[384 + i] * xr[384 + i];
        }
        e[0][j] = s0; e[1][j] = s1; e[2][j] = s2;
    }
    for (; j < 64; j++) e[0][j] = e[1][j] = e[2][j] = 0.0f;

    npart = pt->npart;
    k = 0;
    for (m = 0, i = 0; i < npart; i += 2, m++) {
        float s[3], ts[3];
        q = pt->off[i]; n = pt->cnt[i];
        s[0] = s[1] = s[2] = 0.5f;
        for (j = 0; j < n; j++, k++)
            for (wn = 0; wn < 3; wn++) s[wn] += w[k] * e[wn][q + j];
        q = pt->off[i + 1]; n = pt->cnt[i + 1];
        ts[0] = ts[1] = ts[2] = 0.5f;
        for (j = 0; j < n; j++, k++)
            for (wn = 0; wn < 3; wn++) ts[wn] += w[k] * e[wn][q + j];
        for (wn = 0; wn < 3; wn++) mask[wn][m] = sm[12 * wn + m].mask = s[wn] + ts[wn];
    }
    mpart = (npart + 1) >> 1;
    for (i = 0; i < mpart; i++) {
        float m0 = esave[i], m1 = (float) (2.0 * mask[0][i]), m2 = (float) (2.0 * mask[1][i]), t, tmp;
        esave[i] = (float) (2.0 * mask[2][i]);
        if (block_type_prev == 2) {
            t = mask[0][i];
            if (t > m0) { tmp = 0.1f * t; mask[0][i] = (m0 > tmp) ? m0 : tmp; }
        }
        t = mask[1][i];
        if (t > m1) { tmp = 0.1f * t; mask[1][i] = (m1 > tmp) ? m1 : tmp; }
        t = mask[2][i];
        if (t > m2) { tmp = 0.1f * t; mask[2][i] = (m2 > tmp) ? m2 : tmp; }
        sm[i].mask = mask[0][i];
        sm[12 + i].mask = mask[1][i] + 0.1f * mask[0][i];
        sm[24 + i].mask = mask[2][i] + 0.1f * mask[1][i];
        sm[i].sig = sm[12 + i].sig = sm[24 + i].sig = 0.0f;
    }
}
