/* hxo_alloc.c - ORACLE (test infrastructure): long-block bit allocation / quantisation.
 * Restates CBitAllo3 (bitallo3.cpp:484-3149) and its vector helpers (l3math.c:432-1114)
 * in the reference's evaluation order (bit-exact). */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <assert.h>
#include "hxo_int.h"

#define G_OFFSET 8
#define GMIN_OFFSET 70
#define PART23 4021

static const int pretable[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2};
static const int dNthres[22] = {250, 250, 250, 250, 250, 250, 250, 250, 250, 250, 250,
                                250, 250, 250, 250, 300, 400, 400, 500, 500, 600, 600};

/* working set of one BitAllo call (members of CBitAllo3, pub/bitallo3.h:91-197) */
typedef struct {
    hxo_encoder *e;
    const hxo_params *p;
    float (*xr)[576];
    int (*ix)[576];
    unsigned char (*signx)[576];
    int nchan, ms_flag, block_type;
    int maxBits, maxTargetBits, minTargetBits, PoolBits, TargetBits, deltaMNR, activeBands;
    int snr[2][22], Noise0[2][22], Noise[2][22], NT[2][22];
    float xsxx[2][22], xsxxms[2][22], x34max[2][22];
    int ix10xmax[2][22], gzero[2][22], gmin[2][22], gsf[2][22], sf[2][22], G[2], active_sf[2][22];
    float x34[2][576];
    int preemp[2], scale[2];
    int up[2][22], lo[2][22];       /* current sf limits per channel (psf_upper/lower_limit) */
    hxo_huffsel hs[2];
} ba_t;

/* sf limits by [scalefac_scale][preflag] (bitallo3.cpp:87-162): the format allows 4-bit
   scalefactors below sfb 11 and 3-bit from 11 on; preflag adds pretab */
static int sf_limit_hi(int scale, int pre, int i)      /* sf_limit[0..3], inclusive */
{
    int step = scale ? 4 : 2;
    int base = (i < 11) ? (scale ? 62 : 31) : (scale ? 30 : 15);
    if (pre && i >= 11 && i < 21) base += step * pretable[i];
    return base;
}
static int sf_limit_lo(int scale, int pre, int i)      /* sf_limit[4..5] / sf_lower_limit */
{
    int step = scale ? 4 : 2;
    return (pre && i >= 11 && i < 21) ? step * pretable[i] : 0;
}
static int sf_upper(int scale, int pre, int i)         /* sf_upper_limit[scale][pre] */
{
    int step = scale ? 4 : 2;
    int base = (i < 11) ? (scale ? 60 : 30) : (scale ? 28 : 14);
    if (pre && i >= 11 && i < 21) base += step * pretable[i];
    return base;
}
static void set_limits(ba_t *b, int ch, int scale, int pre)
{
    int i;
    for (i = 0; i < 22; i++) { b->up[ch][i] = sf_upper(scale, pre, i); b->lo[ch][i] = sf_limit_lo(scale, pre, i); }
}

/* ---- l3math.c vector helpers ---- */
/* l3math.c:512-541: quantise a band at gain step gsf, dequantise, return noise power in mB */
static int noise_actual(const hxo_params *p, const float *x34, const float *x, int gsf, int n, int logn)
{
    float sxx = 0.0f, igain = p->look_34igain[gsf], gain = p->look_gain[gsf], xhat, tmp;
    int i, qx;
    for (i = 0; i < n; i++) {
        tmp = (igain * x34[i] + (0.0f - 0.0946f));
        qx = (int) (tmp + copysignf(0.5f, tmp));
        if (qx >= 0 && qx < 256) xhat = gain * p->look_ix43[qx];
        else xhat = (float) (gain * pow(qx, (4.0 / 3.0)));
        tmp = x[i] - xhat;
        sxx += tmp * tmp;
    }
    return hxo_mblog(1.0e-12f + sxx) - logn;
}

/* l3math.c:656-671 */
static int quant_plain(const hxo_params *p, const float *x34, int *ix, int gsf, int n)
{
    float igain = p->look_34igain[gsf];
    int i, m = 0;
    for (i = 0; i < n; i++) {
        ix[i] = (int) (igain * x34[i] + (0.5f - 0.0946f));
        if (ix[i] > m) m = ix[i];
    }
    return m;
}

/* l3math.c:675-694: magnitude-dependent rounding offset */
static int quant_opt(const hxo_params *p, const float *x34, int *ix, int gsf, int n)
{
    float igain = p->look_34igain[gsf], t;
    int i, iq, m = 0;
    for (i = 0; i < n; i++) {
        t = igain * x34[i] + (0.5f - 0.4375f);
        iq = (int) t;
        if (iq > 31) iq = 31;
        ix[i] = (int) (t - hxo_quant_off[iq]);
        if (ix[i] > m) m = ix[i];
    }
    return m;
}

/* l3math.c:698-727: same with the first offset replaced (used for the -HF band) */
static int quant_opt2(const hxo_params *p, const float *x34, int *ix, int gsf, int n, float qadjust)
{
    float igain = p->look_34igain[gsf], t;
    int i, iq, m = 0;
    for (i = 0; i < n; i++) {
        t = igain * x34[i] + (0.5f - 0.4375f);
        iq = (int) t;
        if (iq > 31) iq = 31;
        if (iq < 0) iq = 0;
        ix[i] = (int) (t - (iq == 0 ? qadjust : hxo_quant_off[iq]));
        if (ix[i] > m) m = ix[i];
    }
    return m;
}

/* l3math.c:753-769 / 772-793 */
static void ixmax_quant(const hxo_params *p, const float *x34max, int *ixmax, const int *gsf, int n)
{
    int i, iq;
    float t;
    for (i = 0; i < n; i++) {
        t = p->look_34igain[gsf[i]] * x34max[i] + (0.5f - 0.4375f);
        iq = (int) t;
        if (iq > 31) iq = 31;
        ixmax[i] = (int) (t - hxo_quant_off[iq]);
    }
}
static void ix10xmax_quant(const hxo_params *p, const float *x34max, int *ixmax, const int *gsf, int n)
{
    int i, iq;
    float t;
    for (i = 0; i < n; i++) {
        t = p->look_34igain[gsf[i]] * x34max[i] + (0.5f - 0.4375f);
        iq = (int) t;
        if (iq > 31) iq = 31;
        ixmax[i] = (int) (10.0f * (t - hxo_quant_off[iq]) + (0.5f - 5.0f));
    }
}

/* l3math.c:1088-1114 */
static int inverse_gsf_xfer(const hxo_params *p, const int *qx, const float *x, int n)
{
    float sqq = 0, sxx = 0, q;
    int i;
    for (i = 0; i < n; i++) {
        if (qx[i] < 256) q = p->look_ix43[qx[i]];
        else q = (float) (pow(qx[i], (4.0 / 3.0)));
        sqq += q * q;
        sxx += x[i] * x[i];
    }
    return 54 * hxo_mblog(sxx / sqq) + (8 << 13);
}

static int imax(const int *x, int n)
{
    int i, m = 0;
    for (i = 0; i < n; i++) if (x[i] > m) m = x[i];
    return m;
}

/* ---- bitallo3.cpp:682-754: L/R vs M/S energy-compaction metric with hysteresis ---- */
int hxo_ms_metric_long(hxo_encoder *e, const float x[2][576])
{
    const hxo_params *p = &e->p;
    int i, j, k = 0, n, cm = 0;
    for (i = 0; i < p->nsf[0]; i++) {
        float el = 100.0f, er = 100.0f, t = 0.0f, es, ed, a, b, c;
        int mblr, mbsd, psd;
        n = p->nBand_l[i];
        for (j = 0; j < n; j++, k++) {
            a = x[0][k] * x[0][k];
            b = x[1][k] * x[1][k];
            c = x[0][k] * x[1][k];
            el += a; er += b; t += c;
        }
        es = ed = el + er;
        t = t + t;
        es = es + t;
        ed = ed - t;
        mblr = hxo_mblog(el + er) - hxo_mblog(el > er ? el : er);
        mbsd = hxo_mblog(es + ed) - hxo_mblog(es > ed ? es : ed);
        psd = HXO_MAX(75 - abs(mblr - 120), 0);
        mbsd = HXO_MIN(mbsd, (mbsd >> 1) + 120);
        mbsd += psd;
        cm += p->nBand_l[i] * (mblr - mbsd);
    }
    cm += e->s.ms_correlation_memory;
    e->s.ms_correlation_memory = (cm > 0) ? 5000 : -5000;
    return cm;
}

/* ---- bitallo3.cpp:1069-1126: pull noise targets toward their width-weighted mean ---- */
static void adjust_nt(ba_t *b)
{
    static const int sthres[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 100, 100, 100, 200, 300, 300, 300};
    const hxo_params *p = b->p;
    int ch, i, f = p->test1;
    if (f == 0) return;
    for (ch = 0; ch < b->nchan; ch++) {
        int na = 1, a = 0, nab = 1, ab = 0, d, dmax;
        for (i = 0; i < p->nsf[ch]; i++)
            if (b->snr[ch][i] > sthres[i]) {
                a += b->NT[ch][i]; na++;
                ab += p->nBand_l[i] * b->NT[ch][i];
                nab += p->nBand_l[i];
            }
        a = a / na;
        ab = ab / nab;
        if (na < 5) continue;
        for (i = 0; i < p->nsf[ch]; i++)
            if (b->snr[ch][i] > sthres[i]) {
                dmax = HXO_MAX(b->snr[ch][i] - 400, 0);
                d = (f * (ab - b->NT[ch][i])) >> 4;
                d = HXO_MIN(d, dmax);
                b->NT[ch][i] = b->NT[ch][i] + d;
            }
    }
}

static void gzero_gmin(ba_t *b, const int nband[2])
{
    const hxo_params *p = b->p;
    int ch, i;
    for (ch = 0; ch < b->nchan; ch++) {
        const float *y = b->x34[ch];
        for (i = 0; i < nband[ch]; i++) {
            int n = p->nBand_l[i], j;
            float m = 0.0f;
            for (j = 0; j < n; j++) if (y[j] > m) m = y[j];
            b->x34max[ch][i] = m;
            b->gzero[ch][i] = HXO_MAX(0, hxo_round((0.017716950f * hxo_mblog(m) + (104.585000f - 100.0f + 8.0f))));
            b->gmin[ch][i] = HXO_MAX(0, b->gzero[ch][i] - GMIN_OFFSET);
            y += n;
        }
    }
}

static int drop_guard(int noise0, int nt)   /* bitallo3.cpp:857-864 */
{
    int tsnr = noise0 - nt;
    if (tsnr < 300) {
        tsnr = 187 + ((3 * tsnr) >> 3) - tsnr;
        nt -= tsnr;
    }
    return nt;
}

/* bitallo3.cpp:816-898 (L/R) */
static void startup_lr(ba_t *b, hxo_sigmask sm[][36])
{
    const hxo_params *p = b->p;
    hxo_state *s = &b->e->s;
    int ch, i, j, n, mnr = s->MNR + 100;
    for (ch = 0; ch < b->nchan; ch++) {
        float *x = b->xr[ch];
        unsigned char *sg = b->signx[ch];
        for (i = 0; i < p->nsf3[ch]; i++) {
            float sxx = 0.0f;
            n = p->nBand_l[i];
            for (j = 0; j < n; j++) {
                if (x[j] >= 0.0f) sg[j] = 0;
                else { sg[j] = 1; x[j] = -x[j]; }
                sxx += x[j] * x[j];
            }
            b->xsxx[ch][i] = sxx;
            x += n; sg += n;
        }
    }
    b->activeBands = 0;
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < p->nsf[ch]; i++) {
            b->Noise0[ch][i] = hxo_mblog(b->xsxx[ch][i]) - p->look_log_cbwmb[i];
            if (b->Noise0[ch][i] < -2000) b->NT[ch][i] = b->Noise0[ch][i] + 1000;
            else {
                int mask = hxo_mblog(sm[ch][i].mask) - p->look_log_cbwmb[i];
                b->activeBands += p->nBand_l[i];
                b->NT[ch][i] = drop_guard(b->Noise0[ch][i], mask - mnr + p->taperNT[i]);
            }
            b->snr[ch][i] = b->Noise0[ch][i] - b->NT[ch][i];
        }
    adjust_nt(b);
    for (ch = 0; ch < b->nchan; ch++) hxo_pow34(b->xr[ch], b->x34[ch], p->nbmax3[ch]);
    gzero_gmin(b, p->nsf3);
}

/* bitallo3.cpp:902-1066 (M/S: xr becomes |L+R|, |L-R| without the 1/sqrt2) */
static void startup_ms(ba_t *b, hxo_sigmask sm[][36])
{
    const hxo_params *p = b->p;
    hxo_state *s = &b->e->s;
    int i, j, n = 0, k = 0, mnr;
    if (p->vbr_flag == 0 && s->call_count > 10 && (b->TargetBits - b->minTargetBits) < 100)
        s->MNR = HXO_MIN(s->MNR + 50, 2050);
    mnr = s->MNR;
    b->activeBands = 0;
    for (i = 0; i < p->nsf[0]; i++, k += n) {
        float sl = 0.0f, sr = 0.0f, ss = 0.0f, sd = 0.0f, x0, x1;
        int cbw = p->look_log_cbwmb[i], N0L, N0R, NTL, NTR;
        n = p->nBand_l[i];
        for (j = k; j < k + n; j++) {
            sl += b->xr[0][j] * b->xr[0][j];
            sr += b->xr[1][j] * b->xr[1][j];
        }
        for (j = k; j < k + n; j++) {           /* l3math.c:905-930 */
            x0 = (b->xr[0][j] + b->xr[1][j]);
            x1 = (b->xr[0][j] - b->xr[1][j]);
            b->signx[0][j] = b->signx[1][j] = 0;
            if (x0 < 0.0f) { b->signx[0][j] = 1; x0 = -x0; }
            if (x1 < 0.0f) { b->signx[1][j] = 1; x1 = -x1; }
            b->xr[0][j] = x0;
            b->xr[1][j] = x1;
        }
        for (j = k; j < k + n; j++) {
            ss += b->xr[0][j] * b->xr[0][j];
            sd += b->xr[1][j] * b->xr[1][j];
        }
        b->xsxx[0][i] = sl; b->xsxx[1][i] = sr;
        b->xsxxms[0][i] = ss; b->xsxxms[1][i] = sd;
        N0L = hxo_mblog(sl) - cbw;
        if (N0L < -2000) NTL = 10000;
        else {
            NTL = drop_guard(N0L, (hxo_mblog(sm[0][i].mask) - cbw) - mnr + p->taperNT[i]);
            b->activeBands += n;
        }
        N0R = hxo_mblog(sr) - cbw;
        if (N0R < -2000) NTR = 10000;
        else {
            NTR = drop_guard(N0R, (hxo_mblog(sm[1][i].mask) - cbw) - mnr + p->taperNT[i]);
            b->activeBands += n;
        }
        b->NT[0][i] = NTL; b->NT[1][i] = NTR;
        b->snr[0][i] = N0L - NTL; b->snr[1][i] = N0R - NTR;
        b->Noise0[0][i] = hxo_mblog(ss) - cbw;
        b->Noise0[1][i] = hxo_mblog(sd) - cbw;
    }
    if (p->hf_flag) {
        n = p->nBand_l[21];
        for (j = k; j < k + n; j++) {
            float x0 = (b->xr[0][j] + b->xr[1][j]), x1 = (b->xr[0][j] - b->xr[1][j]);
            b->signx[0][j] = b->signx[1][j] = 0;
            if (x0 < 0.0f) { b->signx[0][j] = 1; x0 = -x0; }
            if (x1 < 0.0f) { b->signx[1][j] = 1; x1 = -x1; }
            b->xr[0][j] = x0; b->xr[1][j] = x1;
        }
    }
    adjust_nt(b);
    for (i = 0; i < p->nsf[0]; i++) {
        int NTL = b->NT[0][i], NTR = b->NT[1][i], Nsum = b->Noise0[0][i], Ndiff = b->Noise0[1][i];
        int xNT = HXO_MIN(NTL, NTR) + 300;
        b->NT[1][i] = b->NT[0][i] = xNT;
        if (Ndiff < xNT) {
            b->NT[0][i] = hxo_logsubber(xNT, Ndiff);
            if (i < 16) b->NT[0][i] -= 200;
        }
        if (Nsum < xNT) b->NT[1][i] = hxo_logsubber(xNT, Nsum);
        b->snr[0][i] = Nsum - b->NT[0][i];
        b->snr[1][i] = Ndiff - b->NT[1][i];
    }
    hxo_pow34(b->xr[0], b->x34[0], p->nbmax2[0]);
    hxo_pow34(b->xr[1], b->x34[1], p->nbmax2[1]);
    gzero_gmin(b, p->nsf2);
}

/* bitallo3.cpp:1130-1160: closed-form first guess of the per-band gain step */
static void seek_initial(ba_t *b)
{
    const hxo_params *p = b->p;
    hxo_state *s = &b->e->s;
    int ch, i;
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < p->nsf[ch]; i++) {
            float g4, d, g;
            s->NTadjust[ch][i] = HXO_MAX(s->NTadjust[ch][i], -400);
            s->NTadjust[ch][i] = HXO_MIN(s->NTadjust[ch][i], 400);
            g4 = 0.017716950f * hxo_mblog(b->x34max[ch][i]) + (88.411238f - 100.0f + 8.0f);
            d = (1.00f / 110.5f) * (1800 - 8 * i - (b->Noise0[ch][i] - b->NT[ch][i] + s->NTadjust[ch][i]));
            g = g4 + d;
            b->gsf[ch][i] = hxo_round(g);
            b->gsf[ch][i] = HXO_MIN(b->gsf[ch][i], b->gzero[ch][i]);
            b->gsf[ch][i] = HXO_MAX(b->gsf[ch][i], b->gmin[ch][i]);
        }
}

/* bitallo3.cpp:1164-1296: measure the real noise and walk gsf (<= 20 steps) toward the target */
static void seek_actual(ba_t *b)
{
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int ch, i, k;
    for (ch = 0; ch < b->nchan; ch++) {
        const float *y34 = b->x34[ch], *y = b->xr[ch];
        for (i = 0; i < p->nsf[ch]; i++) {
            int NTarget = b->NT[ch][i], n = p->nBand_l[i], s = b->gsf[ch][i];
            if (b->Noise0[ch][i] > NTarget) {
                int logn = p->look_log_cbwmb[i];
                int noise = noise_actual(p, y34, y, s, n, logn);
                int dn = noise - NTarget;
                st->NTadjust[ch][i] = st->NTadjust[ch][i] + (dn >> 3);
                if (dn > 100) {                     /* decrease_noise */
                    int t = s - 1, absmin = abs(dn), tnmin = noise, smin = s, niter = HXO_MIN(t, 20);
                    for (k = 0; k < niter; k++) {
                        int tn = noise_actual(p, y34, y, t, n, logn), ad = abs(tn - NTarget);
                        if (ad < absmin) { absmin = ad; tnmin = tn; smin = t; }
                        if (tn <= NTarget) break;
                        t--;
                    }
                    noise = tnmin; s = smin;
                } else if (dn < -100) {             /* increase_noise */
                    int t = s, absmin = abs(dn), tnmin = noise, smin = s;
                    for (k = 0; k < 20; k++) {
                        int tn, ad;
                        t++;
                        tn = noise_actual(p, y34, y, t, n, logn);
                        ad = abs(tn - NTarget);
                        if (ad < absmin) { absmin = ad; tnmin = tn; smin = t; }
                        if (tn >= NTarget) break;
                    }
                    noise = tnmin; s = smin;
                }
                b->gsf[ch][i] = s;
                b->Noise[ch][i] = noise;
            } else {
                b->gsf[ch][i] = b->gzero[ch][i] + 5;
                b->Noise[ch][i] = b->Noise0[ch][i];
            }
            y34 += n; y += n;
        }
    }
}

/* bitallo3.cpp:1793-1862 */
static void sf_final(ba_t *b, int ch)
{
    const hxo_params *p = b->p;
    int i, s, sp0 = 0, sp1 = 0, sp2 = 0, sp3 = 0, scale, pre;
    if (!p->h_id) {     /* fnc_sf_final_MPEG2 (bitallo3.cpp:1860-1888): no pre-emphasis in an LSF granule */
        for (i = 0; i < p->nsf[ch]; i++)
            if (b->active_sf[ch][i]) sp0 |= (sf_limit_hi(0, 0, i) - b->sf[ch][i]);
        b->preemp[ch] = 0;
        b->scale[ch] = (sp0 >= 0) ? 0 : 1;
        return;
    }
    for (i = 0; i < p->nsf[ch]; i++)
        if (b->active_sf[ch][i]) {
            s = b->sf[ch][i];
            sp0 |= (sf_limit_hi(0, 0, i) - s);
            sp1 |= (sf_limit_hi(0, 1, i) - s);
            sp2 |= (sf_limit_hi(1, 0, i) - s);
            sp3 |= (sf_limit_hi(1, 1, i) - s);
            sp1 |= (s - sf_limit_lo(0, 1, i));
            sp3 |= (s - sf_limit_lo(1, 1, i));
        }
    if (sp0 >= 0) { scale = 0; pre = 0; }
    else if (sp1 >= 0) { scale = 0; pre = 1; }
    else if (sp2 >= 0) { scale = 1; pre = 0; }
    else if (sp3 >= 0) { scale = 1; pre = 1; }
    else { scale = 1; pre = 0; }
    b->preemp[ch] = pre;
    b->scale[ch] = scale;
}

/* bitallo3.cpp:1892-2019 (ms = 0) and :2022-2170 (ms = 1): global gain, scalefactors */
static int scale_factors(ba_t *b, int ms)
{
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int ch, i, Gtmp, Gtmpmin = 999, s, d, dN, dsf;
    Gtmp = -1;
    if (ms && st->hf_quant) Gtmp = st->gsf_hf;
    for (ch = 0; ch < b->nchan; ch++) {
        if (!ms) Gtmp = st->gsf_hf_stereo[ch];
        for (i = 0; i < p->nsf[ch]; i++) {
            b->gsf[ch][i] = HXO_MAX(b->gsf[ch][i], b->gmin[ch][i]);
            b->active_sf[ch][i] = 0;
            if (b->gsf[ch][i] < b->gzero[ch][i]) {
                b->active_sf[ch][i] = -1;
                Gtmp = HXO_MAX(Gtmp, b->gsf[ch][i]);
            }
        }
        if (Gtmp < 0) {     /* nothing to code in this channel */
            for (i = 0; i < p->nsf[ch]; i++) {
                b->sf[ch][i] = 0;
                b->gsf[ch][i] = b->gzero[ch][i];
                Gtmp = HXO_MAX(Gtmp, b->gsf[ch][i]);
            }
            b->preemp[ch] = 0;
            b->scale[ch] = 0;
            b->G[ch] = Gtmp;
            if (ms) Gtmpmin = HXO_MIN(Gtmpmin, 100);
            else if (Gtmpmin > 100) Gtmpmin = 100;
            set_limits(b, ch, 0, 0);
            /* NOTE: in the M/S variant the running Gtmp is NOT reset on this path (bitallo3.cpp:2060-2074) */
            continue;
        }
        for (i = 0; i < p->nsf[ch]; i++) {
            if (ms) b->sf[ch][i] = (Gtmp - b->gsf[ch][i]) & b->active_sf[ch][i];
            else { b->sf[ch][i] = 0; if (b->active_sf[ch][i]) b->sf[ch][i] = Gtmp - b->gsf[ch][i]; }
        }
        sf_final(b, ch);
        if (b->scale[ch] == 0) {
            dsf = 2;
            for (i = 0; i < p->nsf[ch]; i++) {
                if (ms) {
                    if (b->active_sf[ch][i]) {
                        if ((b->gzero[ch][i] - b->gsf[ch][i]) < 5) b->sf[ch][i]++;
                        else if ((i < 11) && (b->Noise[ch][i] > b->NT[ch][i])) b->sf[ch][i]++;
                        b->sf[ch][i] &= (~1);
                    }
                } else {
                    if ((i < 11) && (b->Noise[ch][i] > b->NT[ch][i])) b->sf[ch][i]++;
                    b->sf[ch][i] &= (~1);
                }
            }
        } else {
            dsf = 4;
            for (i = 0; i < p->nsf[ch]; i++) {
                if (ms && !b->active_sf[ch][i]) continue;
                s = b->sf[ch][i] & (~3);
                d = b->sf[ch][i] - s;
                dN = b->Noise[ch][i] - b->NT[ch][i] + 150 * d;
                if (dN > dNthres[i]) s = s + 4;
                else if (ms && (b->gzero[ch][i] - b->gsf[ch][i] - d) < 5) s = s + 4;
                b->sf[ch][i] = ms ? s : (s & b->active_sf[ch][i]);
            }
        }
        set_limits(b, ch, b->scale[ch], b->preemp[ch]);
        for (i = 0; i < p->nsf[ch]; i++) {      /* vect_limits, l3math.c:853-866 */
            if (b->sf[ch][i] > b->up[ch][i]) b->sf[ch][i] = b->up[ch][i];
            else if (b->sf[ch][i] < b->lo[ch][i]) b->sf[ch][i] = b->lo[ch][i];
        }
        for (i = 0; i < p->nsf[ch]; i++)
            if (b->active_sf[ch][i]) {
                b->gsf[ch][i] = Gtmp - b->sf[ch][i];
                if (b->gsf[ch][i] < 0) {
                    b->gsf[ch][i] += dsf;
                    b->sf[ch][i] -= dsf;
                    assert(b->sf[ch][i] >= b->lo[ch][i]);
                }
                if (b->gsf[ch][i] >= b->gzero[ch][i]) {
                    b->gsf[ch][i] = b->gzero[ch][i] + 5;
                    b->sf[ch][i] = b->lo[ch][i];
                }
            }
        b->G[ch] = Gtmp;
        if (ms) { Gtmpmin = HXO_MIN(Gtmpmin, Gtmp); Gtmp = -1; }
        else if (Gtmp < Gtmpmin) Gtmpmin = Gtmp;
    }
    return Gtmpmin;
}

/* bitallo3.cpp:1348-1396: try cheaper scalefactors for sfb < 13 that still meet the target */
static void big_lucky_noise(ba_t *b)
{
    const hxo_params *p = b->p;
    int ch, i, m;
    for (ch = 0; ch < b->nchan; ch++) {
        int sdelta = 2 * (1 + b->scale[ch]), GG = b->G[ch];
        const float *y34 = b->x34[ch], *y = b->xr[ch];
        m = HXO_MIN(13, p->nsf[ch]);
        for (i = 0; i < m; i++) {
            int n = p->nBand_l[i];
            if (b->active_sf[ch][i] && (b->gsf[ch][i] < (b->gzero[ch][i] - 5))) {
                int smin = b->sf[ch][i], g0 = b->gzero[ch][i] - 4, s = b->up[ch][i], s0, g, noise;
                int logn = p->look_log_cbwmb[i];
                s = HXO_MIN(b->sf[ch][i] - sdelta, s);
                s0 = b->lo[ch][i];
                for (; s >= s0; s -= sdelta) {
                    g = GG - s;
                    if (g >= g0) break;
                    noise = noise_actual(p, y34, y, g, n, logn);
                    if (noise <= b->NT[ch][i]) { b->Noise[ch][i] = noise; smin = s; }
                }
                b->sf[ch][i] = smin;
                b->gsf[ch][i] = HXO_MAX(GG - smin, 0);
            }
            y34 += n; y += n;
        }
    }
}

static void do_quant(ba_t *b, int opt)      /* bitallo3.cpp:1540-1585 */
{
    const hxo_params *p = b->p;
    int ch, i;
    for (ch = 0; ch < b->nchan; ch++) {
        const float *x = b->x34[ch];
        int *qx = b->ix[ch];
        for (i = 0; i < p->nsf[ch]; i++) {
            int n = p->nBand_l[i];
            b->e->s.ixmax[ch][i] = opt ? quant_opt(p, x, qx, b->gsf[ch][i], n) : quant_plain(p, x, qx, b->gsf[ch][i], n);
            x += n; qx += n;
        }
    }
}

static int count_bits_n(ba_t *b, const int ncb[2])      /* bitallo3.cpp:1740-1780 */
{
    hxo_state *st = &b->e->s;
    int ch, bits = 0;
    for (ch = 0; ch < b->nchan; ch++) {
        st->huff_bits[ch] = hxo_count_bits(b->p, st->ixmax[ch], b->ix[ch], ncb[ch], 1, b->block_type, &b->hs[ch]);
        bits += st->huff_bits[ch];
    }
    return bits;
}
#define count_bits(b) count_bits_n(b, (b)->p->nsf2)
#define count_bits_dual(b) count_bits_n(b, (b)->p->nsf3)

/* -HF support: bitallo3.cpp:1635-1737, 2421-2563 */
static void clear_hf(ba_t *b, int nch)
{
    const hxo_params *p = b->p;
    int ch, i;
    for (ch = 0; ch < nch; ch++)
        for (i = 0; i < p->nBand_l[21]; i++) b->ix[ch][p->startBand_l[21] + i] = 0;
}

static void sparse_quad_counted(int *qx, int n, int level)
{
    int i, c = 0, scnt = 0, m;
    for (i = 0; i < n; i++) c += qx[i];
    c = (level * c) >> 4;
    if (c <= 0) return;
    for (i = n - 4; i >= 0; i -= 4) {
        m = qx[i] + qx[i + 1] + qx[i + 2] + qx[i + 3];
        if (m == 1) {
            qx[i] = qx[i + 1] = qx[i + 2] = qx[i + 3] = 0;
            scnt++;
            if (scnt >= c) break;
        }
    }
}

static void quant_hf(ba_t *b)
{
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int ch, o = p->startBand_l[21], n = p->nBand_l[21];
    for (ch = 0; ch < b->nchan; ch++)
        if (st->hf_quant_stereo[ch]) {
            st->ixmax[ch][21] = quant_opt2(p, b->x34[ch] + o, b->ix[ch] + o, b->G[ch], n, -.30f);
            sparse_quad_counted(b->ix[ch] + o, n, 4);
        }
}

static void quant_hf_ms(ba_t *b)
{
    const hxo_params *p = b->p;
    int o = p->startBand_l[21], n = p->nBand_l[21];
    b->e->s.ixmax[0][21] = quant_opt2(p, b->x34[0] + o, b->ix[0] + o, b->G[0], n, -.30f);
}

/* decide whether band 21 can ride on the global gain (bitallo3.cpp:2421-2493 / 2496-2563) */
static int hf_adjust_ch(ba_t *b, int ch, int *gsf_hf_out)
{
    const hxo_params *p = b->p;
    int i, gmax0 = 0, gmax1 = 0, gmax, gtar, gtar2, gset;
    if (b->gzero[ch][21] <= 8) return 0;
    for (i = 0; i < 11; i++)
        if (b->gsf[ch][i] < b->gzero[ch][i] && b->gsf[ch][i] > gmax0) gmax0 = b->gsf[ch][i];
    for (i = 11; i < p->nsf[ch]; i++)
        if (b->gsf[ch][i] < b->gzero[ch][i] && b->gsf[ch][i] > gmax1) gmax1 = b->gsf[ch][i];
    gtar = HXO_MAX(0, b->gzero[ch][21] - 5);
    gtar2 = HXO_MAX(0, b->gzero[ch][21] - 7);
    gmax = HXO_MAX(gmax0, gmax1);
    if (gtar >= gmax) { *gsf_hf_out = gtar2; return 1; }
    if (gmax0 > gmax1) {
        gset = HXO_MAX(gtar, gmax1);
        if (b->gzero[ch][21] > gset) {
            for (i = 0; i < 11; i++)
                if (b->gsf[ch][i] < b->gzero[ch][i] && b->gsf[ch][i] > gset) b->gsf[ch][i] = gset;
            return 1;
        }
    }
    return 0;
}

static void hf_adjust(ba_t *b)
{
    hxo_state *st = &b->e->s;
    int ch;
    st->gsf_hf_stereo[0] = st->gsf_hf_stereo[1] = -1;
    for (ch = 0; ch < b->nchan; ch++)
        if (hf_adjust_ch(b, ch, &st->gsf_hf_stereo[ch])) st->hf_quant_stereo[ch] = 1;
    st->hf_quant = st->hf_quant_stereo[0] | st->hf_quant_stereo[1];
}

static void hf_adjust_ms(ba_t *b)
{
    hxo_state *st = &b->e->s;
    if (hf_adjust_ch(b, 0, &st->gsf_hf)) st->hf_quant = 1;
}

static void hf_reset_lr(ba_t *b)
{
    hxo_state *st = &b->e->s;
    st->hf_quant = 0;
    st->hf_quant_stereo[0] = st->hf_quant_stereo[1] = 0;
    st->gsf_hf_stereo[0] = st->gsf_hf_stereo[1] = -1;
    st->ixmax[0][21] = st->ixmax[1][21] = 0;
}

/* bitallo3.cpp:2216-2306: flatten isolated peaks in the top bands (L/R only) */
static void trade_dual(ba_t *b)
{
    static const float qo[16] = {0.09460f, 0.02799f, 0.01671f, 0.01192f, 0.00927f, 0.00758f, 0.00641f, 0.00556f,
                                 0.00490f, 0.00439f, 0.00397f, 0.00362f, 0.00333f, 0.00309f, 0.00287f, 0.00269f};
    static const int target_table[16] = {0, 1, 2, 3, 3, 5, 5, 7, 7, 7, 7, 15, 15, 15, 15, 15};
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int ch, i, g, k0, k1, ixmax0, ixtarget;
    float xg, eixmax, feixmax, fetot, factor, ftmp;
    for (ch = 0; ch < b->nchan; ch++) {
        ixmax_quant(p, b->x34max[ch], st->ixmax[ch], b->gsf[ch], p->nsf[ch]);
        ix10xmax_quant(p, b->x34max[ch], b->ix10xmax[ch], b->gsf[ch], p->nsf[ch]);
        for (i = p->nsf[ch] - 1; i >= 11; i--) {
            if (b->ix10xmax[ch][i] > 16) break;
            if (st->ixmax[ch][i] == 2) {
                xg = 1.7717f * hxo_dblog(b->x34max[ch][i] * (1.0f / (1.5f + 0.02799f)));
                g = (int) (xg + 1.0f);
                b->gsf[ch][i] = g + G_OFFSET;
            }
        }
        k1 = i + 1;
        if (k1 < 9) continue;
        k0 = (3 * k1) >> 2;
        if (k0 < 11) k0 = 11;
        if (k0 >= k1) continue;
        ixmax0 = imax(st->ixmax[ch] + k0, k1 - k0);
        if (ixmax0 <= 2) continue;
        fetot = 0; feixmax = 0;
        for (i = k0; i < k1; i++) {
            ftmp = p->rnBand_l[i] * b->xsxx[ch][i];
            fetot += ftmp;
            feixmax += ftmp * b->ix10xmax[ch][i];
        }
        eixmax = feixmax / (1.0f + fetot);
        ixtarget = (int) (0.1f * eixmax + 0.65f);
        if (ixtarget < 2) ixtarget = 2;
        if (ixmax0 <= ixtarget) continue;
        if (ixtarget > 15) continue;
        ixtarget = target_table[ixtarget];
        factor = 1.0f / ((ixtarget + 0.5f) + qo[ixtarget]);
        for (i = k0; i < k1; i++)
            if (st->ixmax[ch][i] > ixtarget) {
                xg = 1.7717f * hxo_dblog(b->x34max[ch][i] * factor);
                g = (int) (xg + 1.0f);
                b->gsf[ch][i] = g + G_OFFSET;
            }
    }
}

/* Rate-loop path statistics for the tools (tools/oracle_rate_stats.py): [0] granules allocated, [1] granules that entered
   increase_bits past its threshold test, [2] its first-pass iterations, [3] its step-backs, [4] granules that entered
   decrease_bits, [5] its rounds, [6] limit_bits rounds, [7] quantise-and-count passes in all.  Not part of the encoder. */
long long hxo_rate_stats[8];

/* bitallo3.cpp:2569-2722 */
static int increase_bits(ba_t *b, int bits0, int ms)
{
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int i, k, ch, bits = bits0, g[2][22], thres = b->minTargetBits - (b->minTargetBits >> 4), pass;
    if (bits0 > thres) return bits0;
    hxo_rate_stats[1]++;
    for (ch = 0; ch < 2; ch++) for (i = 0; i < p->nsf[ms ? 0 : ch]; i++) g[ch][i] = b->gsf[ch][i];
    for (pass = 0; pass < 2; pass++) {
        for (k = 0; k < (pass ? 1 : 10); k++) {
            hxo_rate_stats[pass ? 3 : 2]++; hxo_rate_stats[7]++;
            for (ch = 0; ch < b->nchan; ch++)
                for (i = 0; i < p->nsf[ch]; i++) {
                    if (pass) b->gsf[ch][i] = g[ch][i] + 1;
                    else b->gsf[ch][i] = g[ch][i] = HXO_MAX(g[ch][i] - 1, b->gmin[ch][i]);
                }
            if (ms) {
                st->hf_quant = 0; st->ixmax[0][21] = 0; st->gsf_hf = -1;
                clear_hf(b, 1);
                if (p->hf_flag) hf_adjust_ms(b);
                scale_factors(b, 1);
                do_quant(b, 1);
                st->ixmax[0][21] = 0;
                if (st->hf_quant) quant_hf_ms(b);
                bits = count_bits(b);
            } else {
                if (p->hf_flag & 2) { hf_reset_lr(b); hf_adjust(b); }
                scale_factors(b, 0);
                do_quant(b, 1);
                if (st->hf_quant) quant_hf(b);
                bits = count_bits_dual(b);
            }
            if (!pass && bits >= thres) break;
        }
        if (bits <= b->maxTargetBits) break;     /* else fall back one step (second pass) */
    }
    return bits;
}

/* bitallo3.cpp:2814-2855: raise all noise targets until the bits fit */
static int decrease_bits(ba_t *b, int bits0)
{
    const hxo_params *p = b->p;
    int i, k, ch, bits = bits0, deltaN, f;
    f = (250 * 1024) / (b->activeBands + 10);
    deltaN = (f * (bits0 - b->maxTargetBits)) >> 10;
    deltaN = HXO_MAX(deltaN, 40);
    b->deltaMNR = 0;
    hxo_rate_stats[4]++;
    for (k = 0; k < 10; k++) {
        hxo_rate_stats[5]++; hxo_rate_stats[7]++;
        b->deltaMNR += deltaN;
        for (ch = 0; ch < b->nchan; ch++) for (i = 0; i < p->nsf[ch]; i++) b->NT[ch][i] += deltaN;
        seek_actual(b);
        scale_factors(b, 0);
        do_quant(b, 0);
        bits = count_bits(b);
        if (bits <= b->maxTargetBits) break;
        deltaN = (f * (bits - b->maxTargetBits)) >> 10;
        deltaN = HXO_MAX(deltaN, 40);
    }
    return bits;
}

/* bitallo3.cpp:2725-2746 / 2749-2773 */
static int limit_bits(ba_t *b, int part23)
{
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int i, k, ch, bits = 0;
    for (k = 0; k < 100; k++) {
        hxo_rate_stats[6]++; hxo_rate_stats[7]++;
        for (ch = 0; ch < b->nchan; ch++) {
            if (part23 && st->huff_bits[ch] <= PART23) continue;
            for (i = 0; i < p->nsf[ch]; i++) b->gsf[ch][i] = HXO_MIN(127, b->gsf[ch][i] + 1);
        }
        scale_factors(b, 0);
        do_quant(b, 0);
        bits = count_bits(b);
        if (part23) { if ((st->huff_bits[0] <= PART23) && (st->huff_bits[1] <= PART23)) break; }
        else if (bits <= b->maxBits) break;
    }
    return bits;
}

/* bitallo3.cpp:1471-1536: refine scalefactors of bands whose largest value is 1 or 2 */
static void inverse_sf2(ba_t *b)
{
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int ch, i, n, t, s;
    for (ch = 0; ch < b->nchan; ch++) {
        int Gscale = b->G[ch] << 13, sh = b->scale[ch] ? 14 : 13;
        const float *y = b->xr[ch];
        const int *qx = b->ix[ch];
        for (i = 0; i < p->nsf[ch]; i++) {
            n = p->nBand_l[i];
            if ((st->ixmax[ch][i] == 1) || (st->ixmax[ch][i] == 2)) {
                t = inverse_gsf_xfer(p, qx, y, n);
                s = ((Gscale - t + (1 << sh)) & (~((1 << (sh + 1)) - 1))) >> 13;
                s = HXO_MIN(s, b->up[ch][i]);
                s = HXO_MAX(s, b->lo[ch][i]);
                b->sf[ch][i] = s;
            }
            y += n; qx += n;
        }
    }
}

/* bitallo3.cpp:2948-3046 (ms = 0) / 3048-3149 (ms = 1) */
static int allocate(ba_t *b, int ms)
{
    const hxo_params *p = b->p;
    hxo_state *st = &b->e->s;
    int ch, bits, bits0;
    if (p->hf_flag) {
        if (ms) { st->hf_quant = 0; st->ixmax[0][21] = st->ixmax[1][21] = 0; st->gsf_hf = -1; }
        else hf_reset_lr(b);
        clear_hf(b, b->nchan);
    }
    seek_initial(b);
    seek_actual(b);
    if (ms) { if (p->hf_flag) hf_adjust_ms(b); }
    else { trade_dual(b); if (p->hf_flag & 2) hf_adjust(b); }
    scale_factors(b, ms);
    big_lucky_noise(b);
    do_quant(b, 1);
    hxo_rate_stats[0]++; hxo_rate_stats[7]++;
    if (ms) { st->ixmax[0][21] = 0; if (st->hf_quant) quant_hf_ms(b); bits0 = bits = count_bits(b); }
    else { if (st->hf_quant) quant_hf(b); bits0 = bits = count_bits_dual(b); }
    if (bits < b->minTargetBits && st->MNR < 2000) bits = increase_bits(b, bits, ms);
    if (ms) { st->hf_quant = 0; st->ixmax[0][21] = 0; st->gsf_hf = -1; }
    else if (p->hf_flag) hf_reset_lr(b);
    if (bits > b->maxTargetBits) { clear_hf(b, ms ? 1 : b->nchan); bits = decrease_bits(b, bits); }
    if (bits > b->maxBits) { clear_hf(b, ms ? 1 : b->nchan); bits = limit_bits(b, 0); }
    if (bits > PART23)
        for (ch = 0; ch < b->nchan; ch++)
            if (st->huff_bits[ch] > PART23) { clear_hf(b, ms ? 1 : b->nchan); bits = limit_bits(b, 1); break; }
    inverse_sf2(b);
    return bits0;
}

/* bitallo3.cpp:2897-2944: CBR closed loop on the long-term mask-to-noise target */
static void mnr_feedback(ba_t *b, int activeBands, int bits, int block_type)
{
    hxo_state *st = &b->e->s;
    const hxo_params *p = b->p;
    if (block_type == 2) return;
    if (st->call_count > 10) {
        float deltaNB = 150.0f / (0.20f * (activeBands + 10));
        int mnr = (int) (0.05 * deltaNB * (bits - b->TargetBits));
        int mnr2 = (int) (0.05 * deltaNB * HXO_MAX((bits - b->maxBits), 0));
        int dPool = HXO_MAX(b->TargetBits - bits, 0);
        int dBits = (((2044 + 8 * 5) - b->PoolBits) >> 4) - dPool, mnrp, mnr0, dmnr, maxdmnr;
        dBits = HXO_MAX(dBits, 0);
        dBits = HXO_MIN(dBits, 200);
        mnrp = (int) (deltaNB * dBits);
        mnr0 = (int) (0.2 * deltaNB * HXO_MAX((b->minTargetBits - bits), 0));
        dmnr = mnr + mnr2 + mnrp - mnr0;
        maxdmnr = HXO_MAX(st->MNR - p->initialMNR, b->TargetBits >> 3);
        dmnr = HXO_MIN(dmnr, maxdmnr);
        if (b->deltaMNR) dmnr = HXO_MAX(dmnr, (b->deltaMNR >> 1));
        st->MNR = st->MNR - dmnr;
        st->MNR = HXO_MIN(st->MNR, 2000);
        if (bits > (b->TargetBits + 2000)) st->MNR = HXO_MIN(st->MNR, p->initialMNR);
    }
}

static void null_gr(hxo_gr *g, int block_type)
{
    g->global_gain = 0;
    g->window_switching_flag = (block_type != 0);
    g->block_type = block_type;
    g->mixed_block_flag = 0;
    g->preflag = g->scalefac_scale = 0;
    g->table_select[0] = g->table_select[1] = g->table_select[2] = 0;
    g->big_values = g->region0_count = g->region1_count = g->count1table_select = 0;
    g->aux_nquads = g->aux_bits = g->aux_not_null = 0;
    g->aux_nreg[0] = g->aux_nreg[1] = g->aux_nreg[2] = 0;
}

/* bitallo3.cpp:484-678 */
void hxo_bitallo_long(hxo_encoder *e, float xr[2][576], hxo_sigmask sm[2][36],
                      int min_bits, int target_bits, int max_bits, int bit_pool,
                      hxo_scalefact sf_out[2], hxo_gr gr[2], int ms_flag)
{
    static ba_t bb;     /* large; the oracle is single-threaded by design */
    ba_t *b = &bb;
    hxo_state *st = &e->s;
    const hxo_params *p = &e->p;
    int i, j, ch, FeedbackBits, tbits, t, block_type = gr[0].block_type;

    memset(b, 0, sizeof(*b));
    b->e = e; b->p = p;
    b->block_type = block_type;
    st->call_count++;
    b->deltaMNR = 0;
    if (block_type == 1) {
        if (st->MNR > p->initialMNR) {
            st->MNR = (st->MNR + p->initialMNR) >> 1;
            st->MNR = HXO_MIN(st->MNR, p->initialMNR + 500);
        }
    } else if (block_type == 3) {
        st->MNR = (st->MNR + p->initialMNR) >> 1;
        st->MNR = HXO_MIN(st->MNR, p->initialMNR + 500);
        memset(st->ix, 0, p->nchan * 576 * sizeof(int));
    }
    if (block_type == 2) {
        int MNR0 = st->MNR;
        if (p->vbr_flag == 0) {
            MNR0 = st->MNR - (HXO_MAX(st->MNR - p->initialMNR, 0) >> 1) - (HXO_MAX(st->MNR - p->initialMNR - 400, 0) >> 2);
            MNR0 = HXO_MAX(p->initialMNR + 400, MNR0);
        } else MNR0 = p->initialMNR + 400;
        FeedbackBits = hxo_bitallo_short(e, xr, sm, min_bits, target_bits, max_bits, bit_pool, sf_out, gr, ms_flag, MNR0);
        /* mnr_feedback returns immediately for block_type 2 */
        (void) FeedbackBits;
        return;
    }
    b->ms_flag = ms_flag;
    b->xr = xr; b->signx = st->signx; b->ix = st->ix;
    b->nchan = p->nchan;
    b->maxBits = HXO_MIN(4000 * b->nchan, max_bits);
    b->minTargetBits = min_bits;
    if (b->minTargetBits < 0) b->minTargetBits = 0;
    b->TargetBits = target_bits;
    b->PoolBits = bit_pool;
    if (p->vbr_flag == 0) {
        st->PoolFraction = HXO_MIN(st->PoolFraction + 50, 614);
        if (block_type != 0) st->PoolFraction = 0;
    }
    tbits = ((st->PoolFraction * b->PoolBits) >> 10);
    if (p->vbr_flag == 0) {
        t = HXO_MAX((2050 - 500) + p->initialMNR - st->MNR, 200);
        tbits = HXO_MIN(tbits, t);
    }
    b->maxTargetBits = b->TargetBits + tbits;
    b->maxTargetBits = HXO_MIN(b->maxBits, b->maxTargetBits);
    if (st->MNR < -200) b->minTargetBits = HXO_MAX(b->minTargetBits, (3 * b->TargetBits) >> 2);
    b->maxTargetBits = HXO_MAX(b->minTargetBits, b->maxTargetBits);
    b->minTargetBits = HXO_MIN(b->minTargetBits, b->maxTargetBits - 100);

    if (ms_flag) startup_ms(b, sm); else startup_lr(b, sm);

    if (b->activeBands <= 0) {
        for (i = 0; i < b->nchan; i++) {
            null_gr(&gr[i], block_type);
            for (j = 0; j < 21; j++) sf_out[i].l[j] = 0;
        }
        return;
    }
    FeedbackBits = allocate(b, ms_flag);
    if (p->vbr_flag == 0) mnr_feedback(b, b->activeBands, FeedbackBits, block_type);

    /* output_sf, bitallo3.cpp:760-812 */
    for (ch = 0; ch < b->nchan; ch++) {
        for (i = 0; i < p->nsf[ch]; i++) b->sf[ch][i] >>= (b->scale[ch] == 0) ? 1 : 2;
        if (b->preemp[ch])
            for (i = 11; i < p->nsf[ch]; i++) {
                b->sf[ch][i] -= pretable[i];
                assert(b->sf[ch][i] >= 0);
            }
        for (i = 0; i < 21; i++) sf_out[ch].l[i] = b->sf[ch][i];
    }
    if (ms_flag) { b->G[0] -= 2; b->G[1] -= 2; }
    for (i = 0; i < b->nchan; i++) {
        gr[i].global_gain = b->G[i] + (4 * 32 + 14);
        if (gr[i].global_gain > 255) gr[i].global_gain = 255;
        gr[i].window_switching_flag = (block_type != 0);
        gr[i].block_type = block_type;
        gr[i].mixed_block_flag = 0;
        gr[i].preflag = b->preemp[i];
        gr[i].scalefac_scale = b->scale[i];
        gr[i].aux_bits = st->huff_bits[i];
        gr[i].aux_not_null = st->huff_bits[i];
        hxo_huffsel_to_gr(p, &b->hs[i], &gr[i]);
    }
}
