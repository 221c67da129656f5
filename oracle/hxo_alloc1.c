/* hxo_alloc1.c - ORACLE (test infrastructure): the Helix encoder's first-generation bit allocator, which the
 * reference still uses where the newer one cannot go: joint stereo with intensity coding (MPEG-1 below 96 kbps,
 * MPEG-2 below 48 kbps total) and dual-channel mode.  Long blocks only, CBR only.  Restates
 * bitallo1.cpp:100-1813 (class CBitAllo1); every floating-point expression in the reference's order and
 * precision.  The reference file is C++: log10 / log / sqrt of a float argument there resolve to the float
 * overloads (log10f, logf, sqrtf of libm - checked in the object code: smr_adj, smr_adj_joint, compute_x34,
 * fnc_noise2 and fnc_noise2_cb call them), and the arithmetic around them stays in float where every operand is. */
#include <math.h>
#include <string.h>
#include "hxo_int.h"

#define G_OFFSET 8
#define GMIN_OFFSET 70

static const float sparse_mpeg1[21] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f,
                                       0.20f, 0.30f, 0.40f, 0.50f, 0.60f, 0.70f, 0.80f, 0.90f, 1.0f, 1.5f};
static const float sparse_mpeg2[21] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f,
                                       0.20f, 0.30f, 0.40f, 0.50f, 0.50f, 0.60f, 0.70f, 0.80f, 0.90f};
static const int pretab[22] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0};
static const int pre2[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 17, 17, 17, 17, 19, 19, 21, 21, 21, 19};
static const int pre4[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 35, 35, 35, 35, 39, 39, 43, 43, 43, 39};

/* bitallo1.cpp:107-211 BitAlloInit, :444-543 table generators */
void hxo_a1_init(hxo_encoder *e)
{
    const hxo_params *p = &e->p;
    hxo_a1 *a = &e->s.a1;
    hxo_a1tab *t = &e->a1t;
    int i, j, k, ix;
    double pi, x0, xh, xl, eps, tt, dh, dl, cum, ave_noise;

    memset(a, 0, sizeof(*a));
    t->ill_is_pos = p->h_id ? 7 : 999;
    for (i = 0; i < 21; i++) t->nBand[i] = hxo_sfb_long_edge(p->tix, i + 1) - hxo_sfb_long_edge(p->tix, i);
    for (k = 0, i = 0; i < 21; i++) { t->startBand[i] = k; k += t->nBand[i]; }
    t->startBand[21] = k;
    t->nsf[0] = hxo_sfbl_limit(p->tix, p->band_limit);
    t->nsf[1] = hxo_sfbl_limit(p->tix, p->band_limit_stereo);
    for (i = 0; i < 21; i++) t->look_log_cbw[i] = (float) (10.0 * log10((double) t->nBand[i]));
    /* gen_noise_estimator */
    cum = 0.0f;
    for (ix = 0; ix < 256; ix++) {
        tt = ix + 0.5; xh = tt * pow(tt, 1.0 / 3.0);
        tt = ix; x0 = tt * pow(tt, 1.0 / 3.0);
        tt = ix - 0.5; xl = tt * pow(fabs(tt), 1.0 / 3.0);
        dh = xh - x0; dl = xl - x0;
        eps = (dh * dh * dh - dl * dl * dl) / (3.0 * (xh - xl));
        cum += eps;
        ave_noise = cum / (ix + 1);
        t->look_f_ix[ix] = (float) eps;
        t->look_f_ixmax[ix] = (float) (10.0 * log10(ave_noise));
    }
    cum = 0.0;
    for (i = 0; i < 256; i++) {
        ix = 32 * i + 16;
        tt = ix + 0.5; xh = tt * pow(tt, 1.0 / 3.0);
        tt = ix; x0 = tt * pow(tt, 1.0 / 3.0);
        tt = ix - 0.5; xl = tt * pow(fabs(tt), 1.0 / 3.0);
        dh = xh - x0; dl = xl - x0;
        eps = (dh * dh * dh - dl * dl * dl) / (3.0 * (xh - xl));
        cum += eps;
        ave_noise = cum / (i + 1);
        t->look_f_big_ix[i] = (float) (eps);
        t->look_f_big_ixmax[i] = (float) (10.0 * log10(ave_noise));
    }
    /* gen_bit_estimator */
    t->look_bits[0] = 0;
    for (ix = 1; ix < 256; ix++) t->look_bits[ix] = (int) (16 * (1.4427 * log((double) (ix + 1)) + (ix - 0.6) / ix));
    a->bitadjust = -100;
    a->bitadjust_save[1] = a->bitadjust_save[0] = -100;
    t->gz_con1 = (float) (16.0 / (3.0 * log(2.0)));
    t->gz_con2 = (float) (1 - (16.0 / (3.0 * log(2.0))) * log(.5946) + G_OFFSET);
    t->gz_con0 = (float) (exp((0.99 - t->gz_con2) / t->gz_con1));
    for (i = 0; i < 2; i++) for (j = 0; j < t->nsf[i]; j++) a->gsf[i][j] = a->gsf_save[i][j] = 35;
    a->call_count = 0;
    a->running_a = (1.0f / 20.0f);
    /* gen_atan */
    if (p->h_id) {
        pi = 4.0 * atan(1.0);
        for (i = 0; i < 34; i++) t->look_is_pos[i] = (int) (((12.0 / pi) * atan(sqrt(i / 32.0))) + .25);
    } else {
        for (i = 0; i < 34; i++) {
            k = (int) (-log((i + .0001) / 32.0) / log(2.0) + 0.5);
            if (k < 0) k = 0;
            if (k > 3) k = 3;
            t->look_is_pos[i] = k + k;
        }
    }
    t->con707 = (float) (1.0 / sqrt(2.0));
    memcpy(t->Ssb, p->h_id ? sparse_mpeg1 : sparse_mpeg2, sizeof(t->Ssb));
    a->ave_alpha_nmr = 40.0f;
}

/* bitallo1.cpp:385-431 */
int hxo_a1_ms_metric(hxo_encoder *e, const float x[2][576])
{
    const hxo_a1tab *t = &e->a1t;
    int i, j, k = 0, n, d = 0;
    float a, b, s0, s1;
    for (i = 0; i < t->nsf[0]; i++) {
        n = t->nBand[i];
        s0 = s1 = 0.0f;
        for (j = 0; j < n; j++, k++) {
            a = x[0][k] * x[0][k];
            b = x[1][k] * x[1][k];
            s0 += (a + b);
            a = (float) fabs(a - b);
            s1 += a;
        }
        if (s1 > 0.80 * s0) d++;
        if (s1 > 0.95 * s0) d += 2;
    }
    return t->nsf[0] - 3 * d;
}

typedef struct {
    const hxo_params *p;
    const hxo_a1tab *t;
    hxo_a1 *a;
    int nchan, ms_flag, is_flag;
    float (*xr)[576];
    int (*ix)[576];
    int max_bits, min_bits, max_cnt_bits, target_bits, target0_bits, target0_min, target0_max;
    int huff_bits[2], scale[2], preemp[2], G[2];
    float x34mm, x34[2][576], xsxx[2][21], mask[2][21], x34max[2][21], noise[2][21];
    int ixmax[2][21], gzero[2][21], gmin[2][22], lastGsf[2][21];
    float dGdB, dBG;
    int noise_fn;
    hxo_huffsel hs[2];
} a1_t;

/* bitallo1.cpp:634-668 / :670-907 */
static void smr_adj(a1_t *b, hxo_sigmask sm[][36], unsigned char signx[][576])
{
    const hxo_a1tab *t = b->t;
    hxo_a1 *a = b->a;
    float (*xr)[576] = b->xr;
    const int *nsf = t->nsf;
    int i, j, k, n, ch;
    float r, aa, bb;

    if (!b->is_flag) {
        for (ch = 0; ch < b->nchan; ch++) {
            k = 0;
            for (i = 0; i < nsf[ch]; i++) {
                b->xsxx[ch][i] = 1.0e-12f;
                n = t->nBand[i];
                for (j = 0; j < n; j++, k++) {
                    signx[ch][k] = 0;
                    if (xr[ch][k] < 0.0f) { signx[ch][k] = 1; xr[ch][k] = -xr[ch][k]; }
                    b->xsxx[ch][i] += xr[ch][k] * xr[ch][k];
                }
            }
        }
        for (ch = 0; ch < b->nchan; ch++)
            for (i = 0; i < nsf[ch]; i++) {
                r = sm[ch][i].sig / (sm[ch][i].mask * (0.1f + 0.0001f * b->xsxx[ch][i]));
                if (r < 1.0e-10f) b->mask[ch][i] = 100.0f;
                else b->mask[ch][i] = (float) (-10.0 * log10f(r) - t->look_log_cbw[i]);
            }
        return;
    }
    /* joint stereo with an intensity part */
    if (b->ms_flag == 0) {
        for (ch = 0; ch < b->nchan; ch++) {
            k = 0;
            for (i = 0; i < nsf[1]; i++) {
                b->xsxx[ch][i] = 1.0e-12f;
                n = t->nBand[i];
                for (j = 0; j < n; j++, k++) {
                    signx[ch][k] = 0;
                    if (xr[ch][k] < 0.0f) { signx[ch][k] = 1; xr[ch][k] = -xr[ch][k]; }
                    b->xsxx[ch][i] += xr[ch][k] * xr[ch][k];
                }
            }
        }
    } else {
        k = 0;
        for (i = 0; i < nsf[1]; i++) {
            b->xsxx[0][i] = b->xsxx[1][i] = 1.0e-12f;
            n = t->nBand[i];
            for (j = 0; j < n; j++, k++) {
                b->xsxx[0][i] += xr[0][k] * xr[0][k];
                b->xsxx[1][i] += xr[1][k] * xr[1][k];
                aa = t->con707 * xr[0][k];
                bb = t->con707 * xr[1][k];
                xr[0][k] = aa + bb;
                xr[1][k] = aa - bb;
                signx[0][k] = signx[1][k] = 0;
                if (xr[0][k] < 0.0f) { signx[0][k] = 1; xr[0][k] = -xr[0][k]; }
                if (xr[1][k] < 0.0f) { signx[1][k] = 1; xr[1][k] = -xr[1][k]; }
            }
        }
    }
    if (b->ms_flag) {       /* thin out the side channel */
        k = t->startBand[5];
        for (i = 5; i < nsf[1]; i++) {
            n = t->nBand[i];
            for (j = 0; j < n; j += 2, k += 2) {
                aa = xr[1][k] * xr[1][k] + xr[1][k + 1] * xr[1][k + 1];
                bb = t->Ssb[i] * (aa + xr[0][k] * xr[0][k] + xr[0][k + 1] * xr[0][k + 1]);
                if (aa < bb) xr[1][k] = xr[1][k + 1] = 0.0f;
            }
        }
    }
    /* intensity part: L + R carried in channel 0, rescaled to the bands' joint energy */
    for (i = nsf[1]; i < nsf[0]; i++) {
        b->xsxx[0][i] = b->xsxx[1][i] = 1.0e-12f;
        n = t->nBand[i];
        k = t->startBand[i];
        r = 1.0f;
        for (j = 0; j < n; j++, k++) {
            b->xsxx[0][i] += xr[0][k] * xr[0][k];
            b->xsxx[1][i] += xr[1][k] * xr[1][k];
            xr[0][k] = xr[0][k] + xr[1][k];
            r += xr[0][k] * xr[0][k];
            signx[0][k] = 0;
            if (xr[0][k] < 0.0f) { signx[0][k] = 1; xr[0][k] = -xr[0][k]; }
        }
        if (b->p->h_id) {
            r = (float) (sqrt((b->xsxx[0][i] + b->xsxx[1][i] + 2.0 * sqrtf(b->xsxx[0][i] * b->xsxx[1][i])) / r));
            if (r > 1.5f) r = 1.5f;
        } else {
            aa = (b->xsxx[0][i] > b->xsxx[1][i]) ? b->xsxx[0][i] : b->xsxx[1][i];
            r = (float) (sqrtf(aa / r));
            if (r > 1.2f) r = 1.2f;
        }
        k = t->startBand[i];
        for (j = 0; j < n; j++, k++) xr[0][k] = r * xr[0][k];
    }
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < nsf[1]; i++) {
            r = sm[ch][i].sig / (sm[ch][i].mask * (0.1f + 0.0001f * b->xsxx[ch][i]));
            if (r < 1.0e-10f) b->mask[ch][i] = 100.0f;
            else b->mask[ch][i] = (float) (-10.0 * log10f(r) - t->look_log_cbw[i]);
        }
    for (i = nsf[1]; i < nsf[0]; i++) {
        r = (sm[0][i].sig + sm[1][i].sig) / ((sm[0][i].mask + sm[1][i].mask) * (0.1f + 0.0001f * (b->xsxx[0][i] + b->xsxx[1][i])));
        if (r < 1.0e-10f) b->mask[0][i] = 100.0f;
        else b->mask[0][i] = (float) (-10.0 * log10f(r) - t->look_log_cbw[i]);
    }
    /* the intensity position of each band rides in the right channel's scalefactor */
    for (i = nsf[1]; i < nsf[0]; i++) {
        if (b->xsxx[0][i] <= b->xsxx[1][i]) {
            k = (int) (32.0f * b->xsxx[0][i] / b->xsxx[1][i] + 0.5f);
            a->sf[1][i] = t->look_is_pos[k];
            if (!b->p->h_id && a->sf[1][i] != 0) a->sf[1][i] -= 1;
        } else {
            k = (int) (32.0f * b->xsxx[1][i] / b->xsxx[0][i] + 0.5f);
            a->sf[1][i] = b->p->h_id ? 6 - t->look_is_pos[k] : t->look_is_pos[k];
        }
    }
    if (b->ms_flag)
        for (i = 0; i < nsf[1]; i++) b->mask[1][i] = b->mask[0][i] = 0.5f * (b->mask[0][i] + b->mask[1][i]);
}

/* bitallo1.cpp:589-631 */
static void compute_x34(a1_t *b)
{
    const hxo_a1tab *t = b->t;
    int i, j, k, n, ch;
    for (ch = 0; ch < b->nchan; ch++) hxo_pow34(b->xr[ch], b->x34[ch], t->startBand[t->nsf[ch]]);
    b->x34mm = 0.0f;
    for (ch = 0; ch < b->nchan; ch++) {
        k = 0;
        for (i = 0; i < t->nsf[ch]; i++) {
            b->x34max[ch][i] = 0.0f;
            n = t->nBand[i];
            for (j = 0; j < n; j++, k++) if (b->x34max[ch][i] < b->x34[ch][k]) b->x34max[ch][i] = b->x34[ch][k];
            if (b->x34mm < b->x34max[ch][i]) b->x34mm = b->x34max[ch][i];
            if (b->x34max[ch][i] < t->gz_con0) b->gzero[ch][i] = 0;
            else b->gzero[ch][i] = (int) (t->gz_con1 * logf(b->x34max[ch][i]) + t->gz_con2);
            b->gmin[ch][i] = HXO_MAX(0, b->gzero[ch][i] - GMIN_OFFSET);
        }
    }
}

static void fnc_ixmax(a1_t *b)
{
    int i, ch;
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < b->t->nsf[ch]; i++)
            b->ixmax[ch][i] = (int) ((0.5f - 0.0946f) + b->x34max[ch][i] * b->p->look_34igain[b->a->gsf[ch][i]]);
}

static int fnc_bit_est(a1_t *b)
{
    int i, ixm, ch, n = 0, bits;
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < b->t->nsf[ch]; i++) {
            ixm = b->ixmax[ch][i];
            if (ixm < 256) bits = b->t->look_bits[ixm];
            else if (ixm < 512) bits = 16 * 11;
            else if (ixm < 2048) bits = 16 * 13;
            else bits = 16 * 15;
            n += b->t->nBand[i] * bits;
        }
    return n >> 4;
}

/* bitallo1.cpp:959-1037 (target = target0_bits) and :1040-1133 (target follows the noise level) */
static int fnc_bit_seek(a1_t *b, int second)
{
    hxo_a1 *a = b->a;
    const int *nsf = b->t->nsf;
    int i, j, nbits, delta_bits, mindelta, dG, gz_flag, ch, target = b->target0_bits;
    if (second) {
        target = (int) (b->target0_bits + 0.5f * b->dBG * (a->alpha_nmr - a->ave_alpha_nmr));
        if (target > b->target0_max) target = b->target0_max;
        else if (target < b->target0_min) target = b->target0_min;
    }
    fnc_ixmax(b);
    nbits = fnc_bit_est(b);
    delta_bits = nbits - target;
    if (delta_bits > 0) {
        for (i = 0; i < 10; i++) {
            if (delta_bits <= 0) break;
            dG = (int) (b->dGdB * delta_bits);
            if (dG < 1) dG = 1;
            for (ch = 0; ch < b->nchan; ch++)
                for (j = 0; j < nsf[ch]; j++) {
                    a->gsf[ch][j] += dG;
                    if (a->gsf[ch][j] > b->gzero[ch][j]) a->gsf[ch][j] = b->gzero[ch][j];
                }
            fnc_ixmax(b);
            nbits = fnc_bit_est(b);
            delta_bits = nbits - target;
        }
        return nbits;
    }
    mindelta = target >> 2;
    if (mindelta < 100) mindelta = 100;
    delta_bits = -delta_bits;
    if (delta_bits < mindelta) return nbits;
    for (i = 0; i < 10; i++) {
        dG = (int) (b->dGdB * delta_bits);
        if (dG < 1) dG = 1;
        gz_flag = 0;
        for (ch = 0; ch < b->nchan; ch++)
            for (i = 0; i < nsf[ch]; i++) {     /* (the reference reuses the outer loop's counter here) */
                a->gsf[ch][i] -= dG;
                if (a->gsf[ch][i] < 0) a->gsf[ch][i] = 0;
                gz_flag |= a->gsf[ch][i];
            }
        fnc_ixmax(b);
        nbits = fnc_bit_est(b);
        delta_bits = target - nbits;
        if (delta_bits < mindelta) break;
        if (gz_flag == 0) break;
    }
    return nbits;
}

static float noise_of_ixmax(const hxo_a1tab *t, int ixm, int gsf)
{
    if (ixm < 256) return t->look_f_ixmax[ixm] + 1.505f * gsf;
    ixm >>= 5;
    if (ixm > 255) ixm = 255;
    return t->look_f_big_ixmax[ixm] + 1.505f * gsf;
}

static void fnc_noise(a1_t *b)
{
    int i, ch;
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < b->t->nsf[ch]; i++) b->noise[ch][i] = noise_of_ixmax(b->t, b->ixmax[ch][i], b->a->gsf[ch][i]);
}

static void fnc_noise_cb(a1_t *b, int i, int ch)
{
    int ixm;
    b->ixmax[ch][i] = ixm = (int) ((0.5f - 0.0946f + 0.002f) + b->x34max[ch][i] * b->p->look_34igain[b->a->gsf[ch][i]]);
    b->noise[ch][i] = noise_of_ixmax(b->t, ixm, b->a->gsf[ch][i]);
}

static void fnc_noise2_cb(a1_t *b, int i, int ch)
{
    const hxo_a1tab *t = b->t;
    int j, k, n, ixm, gsf = b->a->gsf[ch][i];
    float sum, igain;
    if (gsf == b->lastGsf[ch][i]) return;
    b->lastGsf[ch][i] = gsf;
    k = t->startBand[i];
    n = t->nBand[i];
    igain = b->p->look_34igain[gsf];
    sum = 0.0f;
    for (j = 0; j < n; j++, k++) {
        ixm = (int) ((0.5f - 0.0946f) + b->x34[ch][k] * igain);
        if (ixm < 256) sum += t->look_f_ix[ixm];
        else {
            ixm >>= 5;
            if (ixm > 255) ixm = 255;
            sum += t->look_f_big_ix[ixm];
        }
    }
    b->noise[ch][i] = (float) (10.0f * log10f(sum) - t->look_log_cbw[i] + 1.505f * gsf);
}

static void fnc_noise2(a1_t *b)
{
    int i, ch;
    for (ch = 0; ch < b->nchan; ch++) for (i = 0; i < b->t->nsf[ch]; i++) fnc_noise2_cb(b, i, ch);
}

static void fnc_noise2_init(a1_t *b)
{
    int i, ch;
    for (ch = 0; ch < b->nchan; ch++) for (i = 0; i < 21; i++) b->lastGsf[ch][i] = -9999;
}

static void noise_cb(a1_t *b, int cb, int ch) { if (b->noise_fn) fnc_noise2_cb(b, cb, ch); else fnc_noise_cb(b, cb, ch); }

static void fnc_ix_quant(a1_t *b)
{
    const hxo_a1tab *t = b->t;
    int i, j, k, n, ch;
    float igain;
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < t->nsf[ch]; i++) {
            if (b->a->gsf[ch][i] == b->lastGsf[ch][i]) continue;
            b->lastGsf[ch][i] = b->a->gsf[ch][i];
            n = t->nBand[i];
            k = t->startBand[i];
            if (b->ixmax[ch][i] <= 0) { for (j = 0; j < n; j++, k++) b->ix[ch][k] = 0; }
            else {
                igain = b->p->look_34igain[b->a->gsf[ch][i]];
                for (j = 0; j < n; j++, k++) b->ix[ch][k] = (int) ((0.5f - 0.0946f) + b->x34[ch][k] * igain);
            }
        }
}

/* bitallo1.cpp:1291-1395 */
static int fnc_noise_seek(a1_t *b)
{
    hxo_a1 *a = b->a;
    const int *nsf = b->t->nsf;
    int i, cb, ch, dg, dgmax, gsf0, gsf00, n = 0;
    float av, dn, dn0, asum = 0.0f;
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < nsf[ch]; i++)
            if ((a->gsf[ch][i] > 0) && (a->gsf[ch][i] < b->gzero[ch][i])) { asum += b->noise[ch][i] - b->mask[ch][i]; n++; }
    if (n <= 1) return 0;
    av = asum / n;
    a->alpha_nmr = av;
    dgmax = 0;
    for (ch = 0; ch < b->nchan; ch++)
        for (cb = 0; cb < nsf[ch]; cb++) {
            dn = b->noise[ch][cb] - b->mask[ch][cb] - av;
            if (dn > 1.0) {
                if (a->gsf[ch][cb] <= 0) continue;
                dn0 = dn;
                gsf00 = gsf0 = a->gsf[ch][cb];
                for (i = 0; i < 50; i++) {
                    if (a->gsf[ch][cb] <= 0) break;
                    dg = (int) (0.5f * dn + 0.5f);
                    if (dg <= 0) break;
                    a->gsf[ch][cb] -= dg;
                    if (a->gsf[ch][cb] < 0) a->gsf[ch][cb] = 0;
                    noise_cb(b, cb, ch);
                    dn = b->noise[ch][cb] - b->mask[ch][cb] - av;
                    if (dn < -1.0f) { dn = dn0 = 0.5f * dn0; a->gsf[ch][cb] = gsf0; continue; }
                    dn0 = dn;
                    gsf0 = a->gsf[ch][cb];
                }
                dg = gsf00 - a->gsf[ch][cb];
                if (dg > dgmax) dgmax = dg;
            } else if (dn < -1.0f) {
                if (a->gsf[ch][cb] >= b->gzero[ch][cb]) continue;
                dn0 = dn;
                gsf00 = gsf0 = a->gsf[ch][cb];
                for (i = 0; i < 50; i++) {
                    if (a->gsf[ch][cb] >= b->gzero[ch][cb]) break;
                    dg = (int) (-0.5f * dn);
                    if (dg <= 0) break;
                    a->gsf[ch][cb] += dg;
                    if (a->gsf[ch][cb] >= b->gzero[ch][cb]) a->gsf[ch][cb] = b->gzero[ch][cb];
                    noise_cb(b, cb, ch);
                    dn = b->noise[ch][cb] - b->mask[ch][cb] - av;
                    if (dn > 1.0f) { dn = dn0 = 0.5f * dn0; a->gsf[ch][cb] = gsf0; continue; }
                    dn0 = dn;
                    gsf0 = a->gsf[ch][cb];
                }
                dg = a->gsf[ch][cb] - gsf00;
                if (dg > dgmax) dgmax = dg;
            }
        }
    return dgmax;
}

/* bitallo1.cpp:1427-1611 */
static void fnc_sf_final(a1_t *b, int ch)
{
    int *sf = b->a->sf[ch];
    const int nsf = b->t->nsf[ch];
    int i, n = HXO_MIN(11, nsf), pre_flag = 0, scale_flag = 0;
    for (i = 0; i < n; i++) if (sf[i] > 31) { scale_flag = 1; break; }
    if (!b->p->h_id) {      /* MPEG-2: no pre-emphasis */
        if (scale_flag == 0) for (i = 11; i < nsf; i++) if (sf[i] > 15) { scale_flag = 1; break; }
        if (scale_flag == 0) {
            for (i = 0; i < n; i++) if (sf[i] > 31) sf[i] = 31;
            for (i = 11; i < nsf; i++) if (sf[i] > 15) sf[i] = 15;
        } else {
            for (i = 0; i < n; i++) if (sf[i] > 63) sf[i] = 63;
            for (i = 11; i < nsf; i++) if (sf[i] > 31) sf[i] = 31;
        }
        b->preemp[ch] = 0;
        b->scale[ch] = scale_flag;
        return;
    }
    if (scale_flag == 0) for (i = 11; i < nsf; i++) if (sf[i] > pre2[i]) { scale_flag = 1; break; }
    if (scale_flag == 0) {
        for (i = 11; i < nsf; i++) if (sf[i] > 15) { pre_flag = 1; break; }
        if (pre_flag) for (i = 11; i < nsf; i++) if ((sf[i] >> 1) < pretab[i]) { pre_flag = 0; break; }
    } else {
        for (i = 11; i < nsf; i++) if (sf[i] > 31) { pre_flag = 1; break; }
        if (pre_flag) for (i = 11; i < nsf; i++) if ((sf[i] >> 2) < pretab[i]) { pre_flag = 0; break; }
    }
    if (scale_flag == 0) {
        for (i = 0; i < n; i++) if (sf[i] > 31) sf[i] = 31;
        if (pre_flag == 0) { for (i = 11; i < nsf; i++) if (sf[i] > 15) sf[i] = 15; }
        else { for (i = 11; i < nsf; i++) if (sf[i] > pre2[i]) sf[i] = pre2[i]; }
    } else {
        for (i = 0; i < n; i++) if (sf[i] > 63) sf[i] = 63;
        if (pre_flag == 0) { for (i = 11; i < nsf; i++) if (sf[i] > 31) sf[i] = 31; }
        else { for (i = 11; i < nsf; i++) if (sf[i] > pre4[i]) sf[i] = pre4[i]; }
    }
    b->preemp[ch] = pre_flag;
    b->scale[ch] = scale_flag;
}

/* bitallo1.cpp:1614-1690 */
static int fnc_scale_factors(a1_t *b)
{
    hxo_a1 *a = b->a;
    int i, ch, Gtmp, Gtmpmin = 999;
    for (ch = 0; ch < b->nchan; ch++) {
        const int nsf = b->t->nsf[ch];
        Gtmp = -1;
        for (i = 0; i < nsf; i++) {
            a->gsf[ch][i] = HXO_MAX(a->gsf[ch][i], b->gmin[ch][i]);
            if ((b->ixmax[ch][i] > 0) && (a->gsf[ch][i] > Gtmp)) Gtmp = a->gsf[ch][i];
        }
        if (Gtmp < 0) {
            for (i = 0; i < nsf; i++) {
                a->sf[ch][i] = 0;
                a->gsf[ch][i] = b->gzero[ch][i];
                if (a->gsf[ch][i] > Gtmp) Gtmp = a->gsf[ch][i];
            }
            b->preemp[ch] = 0;
            b->scale[ch] = 0;
            b->G[ch] = Gtmp;
            if (100 < Gtmpmin) Gtmpmin = 100;
            continue;
        }
        for (i = 0; i < nsf; i++) {
            a->sf[ch][i] = 0;
            if (b->ixmax[ch][i] > 0) a->sf[ch][i] = Gtmp - a->gsf[ch][i];
        }
        fnc_sf_final(b, ch);
        if (b->scale[ch] == 0) for (i = 0; i < nsf; i++) a->sf[ch][i] &= (~1);
        else for (i = 0; i < nsf; i++) a->sf[ch][i] &= (~3);
        for (i = 0; i < nsf; i++) {
            a->gsf[ch][i] = Gtmp - a->sf[ch][i];
            if (a->gsf[ch][i] > b->gzero[ch][i]) a->gsf[ch][i] = b->gzero[ch][i];
        }
        b->G[ch] = Gtmp;
        if (Gtmp < Gtmpmin) Gtmpmin = Gtmp;
    }
    return Gtmpmin;
}

static int quant_and_count(a1_t *b)
{
    int bits = 0, ch;
    fnc_ixmax(b);
    fnc_ix_quant(b);
    for (ch = 0; ch < b->nchan; ch++)
        bits += b->huff_bits[ch] = hxo_count_bits(b->p, b->ixmax[ch], b->ix[ch], b->t->nsf[ch], 1, 0, &b->hs[ch]);
    return bits;
}

/* bitallo1.cpp:1693-1811 */
static int allo_2(a1_t *b)
{
    hxo_a1 *a = b->a;
    int i, j, ch, nbits, dsf, GG, dG, bits, gz_flag, tmp;
    fnc_noise2_init(b);
    b->noise_fn = 0;
    nbits = fnc_bit_seek(b, 0);
    for (i = 0; i < 4; i++) {
        fnc_noise(b);
        dsf = fnc_noise_seek(b);
        if (dsf <= 0) break;
        nbits = fnc_bit_seek(b, 0);
        if (dsf < 2) break;
    }
    b->noise_fn = 1;
    for (i = 0; i < 4; i++) {
        fnc_noise2(b);
        dsf = fnc_noise_seek(b);
        if (dsf <= 0) break;
        nbits = fnc_bit_seek(b, 1);
        if (dsf < 2) break;
    }
    fnc_noise2_init(b);
    fnc_scale_factors(b);
    bits = quant_and_count(b);
    a->bitadjust = a->bitadjust + ((bits - nbits - a->bitadjust) >> 3);
    if ((tmp = b->min_bits - bits) > 0) {
        if (tmp > 200) tmp = 200;
        a->bitadjust = a->bitadjust - (tmp >> 2);
    }
    for (j = 0; j < 3; j++) {       /* spend more if below the minimum */
        if ((b->min_bits - bits) < 50) break;
        dG = (int) (b->dGdB * (b->min_bits - bits));
        if (dG < 1) dG = 1;
        gz_flag = 0;
        for (ch = 0; ch < b->nchan; ch++)
            for (i = 0; i < b->t->nsf[ch]; i++) {
                a->gsf[ch][i] -= dG;
                if (a->gsf[ch][i] < 0) a->gsf[ch][i] = 0;
                gz_flag |= a->gsf[ch][i];
            }
        fnc_scale_factors(b);
        bits = quant_and_count(b);
        if (gz_flag == 0) break;
    }
    for (j = 0; j < 100; j++) {     /* give back if above the maximum */
        if (bits <= b->max_cnt_bits) break;
        dG = (int) (b->dGdB * (bits - b->max_cnt_bits));
        if (dG < 1) dG = 1;
        for (ch = 0; ch < b->nchan; ch++) for (i = 0; i < b->t->nsf[ch]; i++) a->gsf[ch][i] += dG;
        GG = fnc_scale_factors(b);
        bits = quant_and_count(b);
        if (GG >= 100) break;
    }
    for (ch = 0; ch < b->nchan; ch++)
        for (i = 0; i < b->t->nsf[ch]; i++) if (b->ixmax[ch][i] <= 0) a->sf[ch][i] = 0;
    return bits;
}

/* bitallo1.cpp:547-586 */
static void output_sf(a1_t *b, hxo_scalefact sf_out[])
{
    hxo_a1 *a = b->a;
    int i, ch;
    for (ch = 0; ch < b->nchan; ch++) {
        const int nsf = b->t->nsf[ch];
        if (b->scale[ch] == 0) { for (i = 0; i < nsf; i++) a->sf[ch][i] >>= 1; }
        else { for (i = 0; i < nsf; i++) a->sf[ch][i] >>= 2; }
        if (b->preemp[ch]) for (i = 11; i < nsf; i++) a->sf[ch][i] -= pretab[i];
    }
    if (b->is_flag)
        for (i = b->t->nsf[1] - 1; i >= 0; i--) {
            if (b->ixmax[1][i] > 0) break;
            a->sf[1][i] = b->t->ill_is_pos;
        }
    for (ch = 0; ch < b->nchan; ch++) for (i = 0; i < 21; i++) sf_out[ch].l[i] = a->sf[ch][i];
}

/* CBitAllo1::BitAllo (bitallo1.cpp:214-382).  nchan_arg = 2: both channels at once (joint stereo with intensity);
   nchan_arg = 1: channel ch_arg alone (dual channel), whose arrays the caller passes shifted to index 0. */
void hxo_bitallo1(hxo_encoder *e, float xr[][576], hxo_sigmask sm[][36], int ch_arg, int nchan_arg,
                  int min_bits, int target_bits_arg, int max_bits, hxo_scalefact sf_out[], hxo_gr gr[],
                  int ix[][576], unsigned char signx[][576], int ms_flag)
{
    static a1_t B;      /* (large; the oracle is single-threaded by contract) */
    a1_t *b = &B;
    hxo_a1 *a = &e->s.a1;
    const hxo_a1tab *t = &e->a1t;
    int i, j, ch;

    /* per-call members of the reference object; the carried ones live in hxo_a1.  The reference keeps gzero, sf and
       the per-band arrays in the object as well: gzero is recomputed before use, sf persists (hxo_a1). */
    b->p = &e->p; b->t = t; b->a = a;
    b->ms_flag = ms_flag; b->is_flag = e->p.is_flag;
    b->xr = xr; b->ix = ix;
    if (nchan_arg == 1) b->dBG = 0.25f * t->startBand[t->nsf[0]];
    else b->dBG = 0.25f * (t->startBand[t->nsf[0]] + t->startBand[t->nsf[1]]);
    b->dGdB = 1.0f / b->dBG;
    b->nchan = nchan_arg;
    if (nchan_arg == 1) a->bitadjust = a->bitadjust_save[ch_arg];
    b->max_bits = max_bits;
    b->min_bits = min_bits < 0 ? 0 : min_bits;
    b->target_bits = target_bits_arg - (target_bits_arg >> 4);
    if (b->target_bits < b->min_bits) b->target_bits = b->min_bits;
    smr_adj(b, sm, signx);
    compute_x34(b);
    if (b->x34mm < 3.0f) {
        for (i = 0; i < b->nchan; i++) {
            hxo_gr *g = &gr[i];
            g->global_gain = 0; g->window_switching_flag = 0; g->block_type = 0; g->mixed_block_flag = 0;
            g->preflag = 0; g->scalefac_scale = 0;
            g->table_select[0] = g->table_select[1] = g->table_select[2] = 0;
            g->big_values = g->region0_count = g->region1_count = g->count1table_select = 0;
            g->aux_nquads = g->aux_bits = g->aux_not_null = 0;
            g->aux_nreg[0] = g->aux_nreg[1] = g->aux_nreg[2] = 0;
            for (j = 0; j < 21; j++) sf_out[i].l[j] = 0;
        }
        return;
    }
    a->call_count++;
    if (a->call_count <= 20) a->running_a = 1.0f / a->call_count;
    b->max_cnt_bits = b->max_bits;
    if (b->target_bits < b->min_bits) b->target_bits = b->min_bits;
    b->target0_min = b->target_bits >> 1;
    if (b->target0_min < b->min_bits) b->target0_min = b->min_bits;
    b->target0_max = (b->target_bits + b->max_bits) >> 1;
    if (a->bitadjust > (b->target_bits >> 1)) a->bitadjust = b->target_bits >> 1;
    b->target0_bits = b->target_bits - a->bitadjust;
    b->target0_min -= a->bitadjust;
    b->target0_max -= a->bitadjust;
    if (nchan_arg == 1) {
        for (i = 0; i < t->nsf[0]; i++) {
            a->gsf[0][i] = a->gsf_save[ch_arg][i];
            if (a->gsf[0][i] > b->gzero[0][i]) a->gsf[0][i] = b->gzero[0][i];
        }
    } else {
        for (ch = 0; ch < b->nchan; ch++)
            for (i = 0; i < t->nsf[ch]; i++) if (a->gsf[ch][i] > b->gzero[ch][i]) a->gsf[ch][i] = b->gzero[ch][i];
    }
    allo_2(b);
    a->ave_alpha_nmr = a->ave_alpha_nmr + a->running_a * (a->alpha_nmr - a->ave_alpha_nmr);
    output_sf(b, sf_out);
    for (i = 0; i < b->nchan; i++) {
        hxo_gr *g = &gr[i];
        g->global_gain = b->G[i] + (4 * 32 + 14);
        if (g->global_gain > 255) g->global_gain = 255;
        g->window_switching_flag = 0; g->block_type = 0; g->mixed_block_flag = 0;
        g->preflag = b->preemp[i];
        g->scalefac_scale = b->scale[i];
        g->aux_bits = b->huff_bits[i];
        g->aux_not_null = b->huff_bits[i];
        hxo_huffsel_to_gr(&e->p, &b->hs[i], g);
    }
    if (b->is_flag) gr[1].aux_not_null = 1;     /* the right channel's scalefactors carry the intensity positions */
    if (nchan_arg == 1) {
        for (i = 0; i < t->nsf[0]; i++) a->gsf_save[ch_arg][i] = a->gsf[0][i];
        a->bitadjust_save[ch_arg] = a->bitadjust;
    }
}
