/* hxo_short.c - ORACLE (test infrastructure): short-block (3 windows x 192 lines) allocator,
 * its Huffman region/bit counter and the reorder to bitstream order.
 * Restates CBitAlloShort (bitallos.cpp:128-1503), its subdivide2/output_subdivide2
 * (bitallosc.cpp:296-487) and CountBits*Short (cnts.c:95-303) in the reference's evaluation
 * order (bit-exact). */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "hxo_int.h"

#define GMIN_OFFSET 70
#define PART23 4021

typedef struct {
    hxo_encoder *e;
    const hxo_params *p;
    float (*xr)[3][192];
    int nchan, ms_flag, MNR;
    int maxBits, maxTargetBits, minTargetBits, PoolBits, TargetBits, activeBands, FeedbackBits;
    int huff_bits[2], nsf[2];
    int ix[2][3][192];
    unsigned char signx[2][3][192];
    float xsxx[2][3][16], x34max[2][3][16], x34[2][3][192];
    int Noise0[2][3][16], NT[2][3][16], Noise[2][3][16], snr[2][3][16];
    int ixmax[2][3][16], gzero[2][3][16], gmin[2][3][16], gsf[2][3][16], sf[2][3][16], active_sf[2][3][16];
    int subblock_gain[2][3], G[2][3], GG[2], scale[2];
    struct { int table[4]; int cbreg[3]; int nbig, nquads, bits; } save[2];
} sba_t;

void hxo_short_init(hxo_encoder *e)
{
    hxo_params *p = &e->p;
    int i;
    /* bitallos.cpp:128-200: band limits are passed in long-block lines */
    p->nsfs = hxo_sfbs_limit(p->tix, p->band_limit / 3 - 10);
    p->nbmax_s = p->startBand_s[p->nsfs];
    for (i = 0; i < 12; i++) p->look_log_cbwmb_s[i] = (int) (100.0f * hxo_dblog((float) p->nBand_s[i]));
    e->s.s_call_count = 0;
}

/* bitallos.cpp:377-416 */
int hxo_ms_metric_short(hxo_encoder *e, const float xx[2][576])
{
    const hxo_params *p = &e->p;
    const float (*x)[3][192] = (const float (*)[3][192]) xx;
    int i, j, k, n, w, d = 0;
    for (w = 0; w < 3; w++) {
        k = 0;
        for (i = 0; i < p->nsfs; i++) {
            float s0 = 0.0f, s1 = 0.0f, a, b;
            n = p->nBand_s[i];
            for (j = 0; j < n; j++, k++) {
                a = x[0][w][k] * x[0][w][k];
                b = x[1][w][k] * x[1][w][k];
                s0 += (a + b);
                a = (float) fabs(a - b);
                s1 += a;
            }
            if (s1 > 0.80 * s0) d++;
            if (s1 > 0.95 * s0) d += 2;
        }
    }
    return (p->nsfs - d) * 1024;     /* (the reference shifts; the value may be negative) */
}

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

static int drop_guard(int noise0, int nt)
{
    int tsnr = noise0 - nt;
    if (tsnr < 300) { tsnr = 187 + ((3 * tsnr) >> 3) - tsnr; nt -= tsnr; }
    return nt;
}

/* bitallos.cpp:695-741 */
static void adjust_nt(sba_t *b)
{
    int ch, w, i, na = 1, a = 0;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++)
            for (i = 0; i < b->nsf[ch]; i++)
                if (b->snr[ch][w][i] > 0) { a += b->NT[ch][w][i]; na++; }
    a = a / na;
    if (a <= 500) return;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++)
            for (i = 0; i < b->nsf[ch]; i++)
                if (b->snr[ch][w][i] > 0) b->NT[ch][w][i] = (b->NT[ch][w][i] + a) >> 1;
}

static void pow34_gzero(sba_t *b)
{
    const hxo_params *p = b->p;
    int ch, w, i, j;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++) {
            const float *y;
            hxo_pow34(b->xr[ch][w], b->x34[ch][w], p->nbmax_s);
            y = b->x34[ch][w];
            for (i = 0; i < b->nsf[ch]; i++) {
                int n = p->nBand_s[i];
                float m = 0.0f;
                for (j = 0; j < n; j++) if (y[j] > m) m = y[j];
                b->x34max[ch][w][i] = m;
                b->gzero[ch][w][i] = HXO_MAX(0, hxo_round((0.017716950f * hxo_mblog(m) + (104.585000f - 100.0f + 8.0f))));
                b->gmin[ch][w][i] = HXO_MAX(0, b->gzero[ch][w][i] - GMIN_OFFSET);
                y += n;
            }
        }
}

/* bitallos.cpp:475-570 */
static void startup_lr(sba_t *b, hxo_sigmask sm[][3][12])
{
    const hxo_params *p = b->p;
    int ch, w, i, j;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++) {
            float *x = b->xr[ch][w];
            unsigned char *s = b->signx[ch][w];
            for (i = 0; i < b->nsf[ch]; i++) {
                int n = p->nBand_s[i];
                float sxx = 0.0f;
                for (j = 0; j < n; j++) {
                    if (x[j] >= 0.0f) s[j] = 0; else { s[j] = 1; x[j] = -x[j]; }
                    sxx += x[j] * x[j];
                }
                b->xsxx[ch][w][i] = sxx;
                x += n; s += n;
            }
        }
    b->activeBands = 0;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++)
            for (i = 0; i < b->nsf[ch]; i++) {
                int cbw = p->look_log_cbwmb_s[i];
                b->Noise0[ch][w][i] = hxo_mblog(b->xsxx[ch][w][i]) - cbw;
                if (b->Noise0[ch][w][i] < -2000) {
                    b->NT[ch][w][i] = b->Noise0[ch][w][i] + 1000;
                    b->snr[ch][w][i] = -1000;
                } else {
                    int mask = (hxo_mblog(sm[ch][w][i].mask) - cbw);
                    b->NT[ch][w][i] = drop_guard(b->Noise0[ch][w][i], mask - b->MNR);
                    b->snr[ch][w][i] = b->Noise0[ch][w][i] - b->NT[ch][w][i];
                    b->activeBands += p->nBand_s[i];
                }
            }
    adjust_nt(b);
    pow34_gzero(b);
}

/* bitallos.cpp:573-692 */
static void startup_ms(sba_t *b, hxo_sigmask sm[][3][12])
{
    const hxo_params *p = b->p;
    int w, i, j;
    b->activeBands = 0;
    for (w = 0; w < 3; w++) {
        float *x0 = b->xr[0][w], *x1 = b->xr[1][w];
        unsigned char *s0 = b->signx[0][w], *s1 = b->signx[1][w];
        for (i = 0; i < b->nsf[0]; i++) {
            int n = p->nBand_s[i], cbw = p->look_log_cbwmb_s[i], N0L, N0R, NTL, NTR, Nsum, Ndiff, xNT;
            float sl = 0.0f, sr = 0.0f, ss = 0.0f, sd = 0.0f;
            for (j = 0; j < n; j++) { sl += x0[j] * x0[j]; sr += x1[j] * x1[j]; }
            for (j = 0; j < n; j++) {
                float a = (x0[j] + x1[j]), d = (x0[j] - x1[j]);
                s0[j] = s1[j] = 0;
                if (a < 0.0f) { s0[j] = 1; a = -a; }
                if (d < 0.0f) { s1[j] = 1; d = -d; }
                x0[j] = a; x1[j] = d;
            }
            for (j = 0; j < n; j++) { ss += x0[j] * x0[j]; sd += x1[j] * x1[j]; }
            b->xsxx[0][w][i] = sl; b->xsxx[1][w][i] = sr;
            N0L = hxo_mblog(sl) - cbw;
            if (N0L < -2000) NTL = 10000;
            else { NTL = drop_guard(N0L, (hxo_mblog(sm[0][w][i].mask) - cbw) - b->MNR); b->activeBands += n; }
            N0R = hxo_mblog(sr) - cbw;
            if (N0R < -2000) NTR = 10000;
            else { NTR = drop_guard(N0R, (hxo_mblog(sm[1][w][i].mask) - cbw) - b->MNR); b->activeBands += n; }
            b->Noise0[0][w][i] = Nsum = hxo_mblog(ss) - cbw;
            b->Noise0[1][w][i] = Ndiff = hxo_mblog(sd) - cbw;
            xNT = HXO_MIN(NTR, NTL) + 300;
            b->NT[1][w][i] = b->NT[0][w][i] = xNT;
            if (Ndiff < xNT) { b->NT[0][w][i] = hxo_logsubber(xNT, Ndiff); b->NT[0][w][i] -= 200; }
            if (Nsum < xNT) { b->NT[1][w][i] = hxo_logsubber(xNT, Nsum); b->NT[1][w][i] -= 200; }
            b->snr[0][w][i] = b->Noise0[0][w][i] - b->NT[0][w][i];
            b->snr[1][w][i] = b->Noise0[1][w][i] - b->NT[1][w][i];
            x0 += n; x1 += n; s0 += n; s1 += n;
        }
    }
    adjust_nt(b);
    pow34_gzero(b);
}

/* bitallos.cpp:744-769 */
static void seek_initial(sba_t *b)
{
    int ch, w, i;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++)
            for (i = 0; i < b->nsf[ch]; i++) {
                float g4 = 0.017716950f * hxo_mblog(b->x34max[ch][w][i]) + (88.411238f - 100.0f + 8.0f);
                float d = (1.00f / 110.5f) * (1800 - (2 * 8) * i - (b->Noise0[ch][w][i] - b->NT[ch][w][i]));
                float g = g4 + d;
                int gs = hxo_round(g);
                gs = HXO_MIN(gs, b->gzero[ch][w][i]);
                gs = HXO_MAX(gs, b->gmin[ch][w][i]);
                b->gsf[ch][w][i] = gs;
            }
}

/* bitallos.cpp:772-898 */
static void seek_actual(sba_t *b)
{
    const hxo_params *p = b->p;
    int ch, w, i, k;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++) {
            const float *y34 = b->x34[ch][w], *y = b->xr[ch][w];
            for (i = 0; i < b->nsf[ch]; i++) {
                int NTarget = b->NT[ch][w][i], n = p->nBand_s[i], s = b->gsf[ch][w][i];
                if (b->Noise0[ch][w][i] > NTarget) {
                    int logn = p->look_log_cbwmb_s[i];
                    int noise = noise_actual(p, y34, y, s, n, logn), dn = noise - NTarget;
                    if (dn > 100) {
                        int t = s - 1, absmin = abs(dn), tnmin = noise, smin = s, niter = HXO_MIN(t, 20);
                        for (k = 0; k < niter; k++) {
                            int tn = noise_actual(p, y34, y, t, n, logn), ad = abs(tn - NTarget);
                            if (ad < absmin) { absmin = ad; tnmin = tn; smin = t; }
                            if (tn <= NTarget) break;
                            t--;
                        }
                        noise = tnmin; s = smin;
                    } else if (dn < -100) {
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
                    b->gsf[ch][w][i] = s;
                    b->Noise[ch][w][i] = noise;
                } else {
                    b->gsf[ch][w][i] = b->gzero[ch][w][i] + 5;
                    b->Noise[ch][w][i] = b->Noise0[ch][w][i];
                }
                y34 += n; y += n;
            }
        }
}

/* bitallos.cpp:1142-1276: per-window gain (subblock_gain in steps of 8) + scalefactors */
static void scale_factors(sba_t *b)
{
    int ch, w, i, Gtmp, s, d, dN;
    for (ch = 0; ch < b->nchan; ch++) {
        int sp0 = 0;
        b->scale[ch] = 0;
        for (w = 0; w < 3; w++) {
            Gtmp = -1;
            for (i = 0; i < b->nsf[ch]; i++) {
                b->gsf[ch][w][i] = HXO_MAX(b->gsf[ch][w][i], b->gmin[ch][w][i]);
                b->active_sf[ch][w][i] = 0;
                if (b->gsf[ch][w][i] < b->gzero[ch][w][i]) {
                    b->active_sf[ch][w][i] = -1;
                    Gtmp = HXO_MAX(Gtmp, b->gsf[ch][w][i]);
                }
            }
            b->G[ch][w] = Gtmp;
        }
        b->GG[ch] = HXO_MAX(b->G[ch][0], b->G[ch][1]);
        b->GG[ch] = HXO_MAX(b->GG[ch], b->G[ch][2]);
        for (w = 0; w < 3; w++) {
            Gtmp = b->G[ch][w];
            if (Gtmp < 0) {
                b->subblock_gain[ch][w] = 0;
                for (i = 0; i < b->nsf[ch]; i++) { b->sf[ch][w][i] = 0; b->gsf[ch][w][i] = b->gzero[ch][w][i]; }
            } else {
                b->subblock_gain[ch][w] = (b->GG[ch] - Gtmp) & (~7);
                b->subblock_gain[ch][w] = HXO_MIN(b->subblock_gain[ch][w], 7 * 8);
                Gtmp = b->GG[ch] - b->subblock_gain[ch][w];
                b->G[ch][w] = Gtmp;
                for (i = 0; i < b->nsf[ch]; i++) {
                    b->sf[ch][w][i] = 0;
                    if (b->active_sf[ch][w][i]) b->sf[ch][w][i] = Gtmp - b->gsf[ch][w][i];
                }
            }
        }
        for (i = 0; i < b->nsf[ch]; i++)            /* fnc_sf_final, bitallos.cpp:1013-1039 */
            for (w = 0; w < 3; w++)
                if (b->active_sf[ch][w][i]) sp0 |= ((i < 6 ? 31 : 15) - b->sf[ch][w][i]);
        b->scale[ch] = (sp0 >= 0) ? 0 : 1;
        for (w = 0; w < 3; w++) {
            if (b->G[ch][w] < 0) continue;
            for (i = 0; i < b->nsf[ch]; i++) {
                if (b->scale[ch] == 0) {
                    if (b->Noise[ch][w][i] > b->NT[ch][w][i]) b->sf[ch][w][i]++;
                    b->sf[ch][w][i] = HXO_MIN(b->G[ch][w], b->sf[ch][w][i]);
                    b->sf[ch][w][i] &= (~1);
                } else {
                    s = b->sf[ch][w][i] & (~3);
                    d = b->sf[ch][w][i] - s;
                    dN = b->Noise[ch][w][i] - b->NT[ch][w][i] + 150 * d;
                    if (dN > 250) { s = s + 4; s = HXO_MIN(b->G[ch][w], s) & (~3); }
                    b->sf[ch][w][i] = s;
                }
            }
        }
        for (w = 0; w < 3; w++) {
            if (b->G[ch][w] < 0) continue;
            for (i = 0; i < b->nsf[ch]; i++) {      /* vect_limits against sf_upper_limit[scale] / 0 */
                int up = (i < 6) ? (b->scale[ch] ? 60 : 30) : (b->scale[ch] ? 28 : 14);
                if (b->sf[ch][w][i] > up) b->sf[ch][w][i] = up;
                else if (b->sf[ch][w][i] < 0) b->sf[ch][w][i] = 0;
            }
        }
        for (w = 0; w < 3; w++) {
            if (b->G[ch][w] < 0) continue;
            for (i = 0; i < b->nsf[ch]; i++)
                if (b->active_sf[ch][w][i]) {
                    b->gsf[ch][w][i] = b->G[ch][w] - b->sf[ch][w][i];
                    if (b->gsf[ch][w][i] >= b->gzero[ch][w][i]) { b->gsf[ch][w][i] = b->gzero[ch][w][i]; b->sf[ch][w][i] = 0; }
                }
        }
    }
}

/* bitallos.cpp:943-998: opt = quantB (first rounding offset replaced by -0.30), else plain */
static void do_quant(sba_t *b, int opt)
{
    const hxo_params *p = b->p;
    int ch, w, i, j;
    for (ch = 0; ch < b->nchan; ch++)
        for (w = 0; w < 3; w++) {
            const float *x = b->x34[ch][w];
            int *qx = b->ix[ch][w];
            for (i = 0; i < b->nsf[ch]; i++) {
                int n = p->nBand_s[i], m = 0;
                float igain = p->look_34igain[b->gsf[ch][w][i]];
                for (j = 0; j < n; j++) {
                    if (opt) {
                        float t = igain * x[j] + (0.5f - 0.4375f);
                        int iq = (int) t;
                        if (iq > 31) iq = 31;
                        if (iq < 0) iq = 0;
                        qx[j] = (int) (t - (iq == 0 ? -.30f : hxo_quant_off[iq]));
                    } else qx[j] = (int) (igain * x[j] + (0.5f - 0.0946f));
                    if (qx[j] > m) m = qx[j];
                }
                b->ixmax[ch][w][i] = m;
                x += n; qx += n;
            }
        }
}

/* candidate tables by region maximum: shared with the long-block counter */
typedef struct { int ncand; int t[4]; int tmax; } cand_t;
void hxo_huff_candidates(int rmax, int *ncand, int t[4], int *tmax);
int hxo_huff_pair_len(int t, int x, int y);

static int count_region(const cand_t *c, int ix[3][192], int off, int n, int *index)
{
    int bb[4] = {0, 0, 0, 0}, i, k, w, bits;
    *index = 0;
    if (c->ncand == 0 || n <= 0) return 0;
    for (w = 0; w < 3; w++)
        for (i = 0; i < n; i += 2)
            for (k = 0; k < c->ncand; k++) bb[k] += hxo_huff_pair_len(c->t[k], ix[w][off + i], ix[w][off + i + 1]);
    for (k = 0; k < 4; k++) bb[k] &= 0xFFFF;
    if (bb[0] < bb[1]) { bits = bb[0]; *index = 0; } else { bits = bb[1]; *index = 1; }
    if (c->ncand == 4) {
        if (bb[2] <= bits) { bits = bb[2]; *index = 2; }
        if (bb[3] <= bits) { bits = bb[3]; *index = 3; }
    }
    return bits;
}

/* bitallosc.cpp:296-428: region 0 = 3 short sfb, region 1 = rest of the big values; count1
   region = whole sfb's in bitstream (sfb-major) order, padded to a multiple of 4 */
static int count_bits_ch(sba_t *b, int ch)
{
    const hxo_params *p = b->p;
    int (*ixmax)[16] = b->ixmax[ch];
    int (*ix)[192] = b->ix[ch];
    int ncb = b->nsf[ch], cb[3], i, j, w, k, n0, nbig, nquads, bits, idx, r0 = 0, r1 = 0, qa = 0, qb = 0;
    int re[576 + 4];
    cand_t c0, c1;
    cb[0] = 3;
    for (i = ncb - 1; i >= 0; i--) if (ixmax[0][i] > 0 || ixmax[1][i] > 0 || ixmax[2][i] > 0) break;
    cb[2] = i + 1;
    for (; i >= 0; i--) if (ixmax[0][i] > 1 || ixmax[1][i] > 1 || ixmax[2][i] > 1) break;
    cb[1] = i + 1;
    cb[1] = HXO_MAX(cb[1], 3);
    cb[2] = HXO_MAX(cb[2], cb[1]);
    nbig = p->startBand_s[cb[1]];
    for (i = 0; i < cb[0]; i++) for (w = 0; w < 3; w++) if (r0 < ixmax[w][i]) r0 = ixmax[w][i];
    for (; i < cb[1]; i++) for (w = 0; w < 3; w++) if (r1 < ixmax[w][i]) r1 = ixmax[w][i];
    hxo_huff_candidates(r0, &c0.ncand, c0.t, &c0.tmax);
    hxo_huff_candidates(r1, &c1.ncand, c1.t, &c1.tmax);
    n0 = p->startBand_s[cb[0]];
    bits = count_region(&c0, ix, 0, n0, &idx);
    b->save[ch].table[0] = c0.t[idx];
    bits += count_region(&c1, ix, n0, nbig - n0, &idx);
    b->save[ch].table[1] = c1.t[idx];
    b->save[ch].table[2] = 0;
    k = 0;
    for (i = cb[1]; i < cb[2]; i++)
        for (w = 0; w < 3; w++)
            for (j = p->startBand_s[i]; j < p->startBand_s[i + 1]; j++) re[k++] = ix[w][j];
    re[k] = re[k + 1] = re[k + 2] = 0;
    k = (k + 3) & (~3);
    nquads = k >> 2;
    for (i = 0; i < nquads; i++) {
        const int *v = re + 4 * i;
        int pop = v[0] + v[1] + v[2] + v[3];
        qa += hxo_quada_len(((v[0] << 3) + (v[1] << 2) + (v[2] << 1) + v[3]) & 15) + pop;
        qb += 4 + pop;
    }
    if (nquads > 0) { if (qa < qb) { bits += qa; idx = 0; } else { bits += qb; idx = 1; } } else idx = 0;
    b->save[ch].table[3] = idx;
    b->save[ch].cbreg[0] = cb[0]; b->save[ch].cbreg[1] = cb[1]; b->save[ch].cbreg[2] = cb[2];
    b->save[ch].nbig = nbig;
    b->save[ch].nquads = nquads;
    b->save[ch].bits = bits;
    return bits;
}

static int count_bits(sba_t *b)
{
    int ch, bits = 0;
    for (ch = 0; ch < b->nchan; ch++) { b->huff_bits[ch] = count_bits_ch(b, ch); bits += b->huff_bits[ch]; }
    return bits;
}

static void bump_gsf(sba_t *b, int delta, int only_over)
{
    int ch, w, i;
    for (ch = 0; ch < b->nchan; ch++) {
        if (only_over && b->huff_bits[ch] <= PART23) continue;
        for (w = 0; w < 3; w++)
            for (i = 0; i < b->nsf[ch]; i++)
                b->gsf[ch][w][i] = (delta < 0) ? HXO_MAX(b->gsf[ch][w][i] - 1, 0) : HXO_MIN(127, b->gsf[ch][w][i] + 1);
    }
}

/* bitallos.cpp:1450-1503 with :1279-1447 */
static void allocate(sba_t *b)
{
    int ch, w, i, k, bits, f, deltaN;
    if (b->MNR < -200) b->minTargetBits = HXO_MAX(b->minTargetBits, (3 * b->TargetBits) >> 2);
    seek_initial(b);
    seek_actual(b);
    scale_factors(b);
    do_quant(b, 1);
    b->FeedbackBits = bits = count_bits(b);
    if (bits < b->minTargetBits)
        for (k = 0; k < 10; k++) {
            bump_gsf(b, -1, 0);
            scale_factors(b); do_quant(b, 1); bits = count_bits(b);
            if (bits >= b->minTargetBits) break;
        }
    if (bits > b->maxTargetBits) {
        f = (250 * 1024) / (b->activeBands + 10);
        deltaN = HXO_MAX((f * (bits - b->maxTargetBits)) >> 10, 40);
        for (k = 0; k < 10; k++) {
            for (ch = 0; ch < b->nchan; ch++) for (w = 0; w < 3; w++) for (i = 0; i < b->nsf[ch]; i++) b->NT[ch][w][i] += deltaN;
            seek_actual(b);
            scale_factors(b); do_quant(b, 0); bits = count_bits(b);
            if (bits <= b->maxTargetBits) break;
            deltaN = HXO_MAX((f * (bits - b->maxTargetBits)) >> 10, 40);
        }
    }
    if (bits > b->maxBits)
        for (k = 0; k < 100; k++) {
            bump_gsf(b, 1, 0);
            scale_factors(b); do_quant(b, 0); bits = count_bits(b);
            if (bits <= b->maxBits) break;
        }
    if (bits > PART23 && (b->huff_bits[0] > PART23 || b->huff_bits[1] > PART23))
        for (k = 0; k < 100; k++) {
            bump_gsf(b, 1, 1);
            scale_factors(b); do_quant(b, 0); bits = count_bits(b);
            if ((b->huff_bits[0] <= PART23) && (b->huff_bits[1] <= PART23)) break;
        }
}

/* bitallos.cpp:202-369 */
int hxo_bitallo_short(hxo_encoder *e, float xr[2][576], hxo_sigmask smarg[2][36],
                      int min_bits, int target_bits, int max_bits, int bit_pool,
                      hxo_scalefact sf_out[2], hxo_gr gr[2], int ms_flag, int MNR)
{
    static sba_t bb;
    sba_t *b = &bb;
    const hxo_params *p = &e->p;
    hxo_sigmask (*sm)[3][12] = (hxo_sigmask (*)[3][12]) smarg;
    int ch, i, j, w, k, n;
    memset(b, 0, sizeof(*b));
    b->e = e; b->p = p;
    b->MNR = MNR;
    if (!p->h_id) b->MNR = HXO_MIN(b->MNR, 850);      /* bitallos.cpp:214-217 */
    e->s.s_call_count++;
    b->ms_flag = ms_flag;
    b->xr = (float (*)[3][192]) xr;
    b->nchan = p->nchan;
    b->nsf[0] = b->nsf[1] = p->nsfs;
    b->maxBits = HXO_MIN(4000 * b->nchan, max_bits);
    b->minTargetBits = min_bits;
    if (b->minTargetBits < 0) b->minTargetBits = 0;
    b->TargetBits = target_bits;
    b->PoolBits = bit_pool;
    b->maxTargetBits = b->TargetBits + ((614 * b->PoolBits) >> 10);
    b->maxTargetBits = (b->maxBits + b->maxTargetBits) >> 1;
    b->maxTargetBits = HXO_MIN(b->maxBits, b->maxTargetBits);
    if (ms_flag) startup_ms(b, sm); else startup_lr(b, sm);
    if (b->activeBands <= 0) {
        for (ch = 0; ch < b->nchan; ch++) {
            hxo_gr *g = &gr[ch];
            g->global_gain = 0; g->window_switching_flag = 1; g->block_type = 2; g->mixed_block_flag = 0;
            g->preflag = g->scalefac_scale = 0;
            g->table_select[0] = g->table_select[1] = g->table_select[2] = 0;
            g->subblock_gain[0] = g->subblock_gain[1] = g->subblock_gain[2] = 0;
            g->big_values = g->region0_count = g->region1_count = g->count1table_select = 0;
            g->aux_nquads = g->aux_bits = g->aux_not_null = 0;
            g->aux_nreg[0] = g->aux_nreg[1] = g->aux_nreg[2] = 0;
            for (w = 0; w < 3; w++) for (j = 0; j < 12; j++) sf_out[ch].s[w][j] = 0;
        }
        return 0;
    }
    allocate(b);
    if (ms_flag) { b->GG[0] -= 2; b->GG[1] -= 2; }
    b->GG[0] = HXO_MAX(b->GG[0], 0);
    b->GG[1] = HXO_MAX(b->GG[1], 0);
    for (ch = 0; ch < b->nchan; ch++) {
        hxo_gr *g = &gr[ch];
        int n0, n1, n2;
        g->global_gain = b->GG[ch] + (4 * 32 + 14);
        if (g->global_gain > 255) g->global_gain = 255;
        g->window_switching_flag = 1; g->block_type = 2; g->mixed_block_flag = 0; g->preflag = 0;
        g->scalefac_scale = b->scale[ch];
        g->aux_bits = b->huff_bits[ch];
        g->aux_not_null = b->huff_bits[ch];
        for (w = 0; w < 3; w++) g->subblock_gain[w] = b->subblock_gain[ch][w] >> 3;
        if (b->save[ch].bits <= 0) {            /* output_subdivide2, bitallosc.cpp:431-487 */
            g->table_select[0] = g->table_select[1] = g->table_select[2] = 0;
            g->big_values = g->region0_count = g->region1_count = 0;
            g->aux_nreg[0] = g->aux_nreg[1] = g->aux_nreg[2] = 0;
            g->aux_nquads = 0; g->count1table_select = 0;
        } else {
            g->table_select[0] = b->save[ch].table[0];
            g->table_select[1] = b->save[ch].table[1];
            g->table_select[2] = b->save[ch].table[2];
            g->count1table_select = b->save[ch].table[3];
            g->big_values = 3 * (b->save[ch].nbig >> 1);
            g->region0_count = g->region1_count = 0;
            n0 = p->startBand_s[b->save[ch].cbreg[0]];
            n1 = p->startBand_s[b->save[ch].cbreg[1]];
            n2 = p->startBand_s[b->save[ch].cbreg[2]];
            if (n2 > b->save[ch].nbig) n2 = b->save[ch].nbig;
            if (n1 > n2) n1 = n2;
            if (n0 > n1) n0 = n1;
            n2 = n2 - n1; n1 = n1 - n0;
            g->aux_nreg[0] = 3 * (n0 >> 1);
            g->aux_nreg[1] = 3 * (n1 >> 1);
            g->aux_nreg[2] = 3 * (n2 >> 1);
            g->aux_nquads = b->save[ch].nquads;
        }
    }
    for (ch = 0; ch < b->nchan; ch++)           /* output_sf, bitallos.cpp:419-471 */
        for (w = 0; w < 3; w++) {
            for (i = 0; i < b->nsf[ch]; i++) b->sf[ch][w][i] >>= (b->scale[ch] == 0) ? 1 : 2;
            for (i = 0; i < 12; i++) sf_out[ch].s[w][i] = b->sf[ch][w][i];
        }
    for (ch = 0; ch < b->nchan; ch++) {         /* reorder to sfb-major bitstream order */
        memset(e->s.ix[ch], 0, 576 * sizeof(int));
        k = 0;
        n = b->save[ch].cbreg[2];
        for (i = 0; i < n; i++)
            for (w = 0; w < 3; w++)
                for (j = p->startBand_s[i]; j < p->startBand_s[i + 1]; j++) {
                    e->s.ix[ch][k] = b->ix[ch][w][j];
                    e->s.signx[ch][k] = b->signx[ch][w][j];
                    k++;
                }
    }
    return b->FeedbackBits;
}
