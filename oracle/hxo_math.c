/* hxo_math.c - ORACLE (test infrastructure): millibel log/exp, x^(3/4), table generation.
 * Restates l3math.c:228-366, pow34.c:132-154, l3init.c:176-347, amodini2.c:587-940. */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <assert.h>
#include "hxo_int.h"
#include "hxo_iso_data.inc"

static int tab_ready;
static int mblog_tab[256];          /* l3math.c:180-225: round(1000*log10(1+(m+.5)/256)) - 38227 */
static float mbexp_lo[256], mbexp_hi[256];  /* l3math.c:246-338: 10^(x/1000) split into low/high byte */
static float pow34_exp[256];        /* pow34.c:43-108: 2^(0.75*(e-127)) */
int hxo_logsub_tab[84];             /* l3math.c:57-79 */
float hxo_quant_off[32];            /* l3math.c:81-114 rounding offsets minus 0.4375 */

float hxo_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
uint32_t hxo_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

__attribute__((constructor)) void hxo_math_init(void)
{
    static const double q[32] = {
        0.09460, 0.02799, 0.01671, 0.01192, 0.00927, 0.00758, 0.00641, 0.00556, 0.00490, 0.00439, 0.00397,
        0.00362, 0.00333, 0.00309, 0.00287, 0.00269, 0.00253, 0.00238, 0.00225, 0.00214, 0.00203, 0.00194,
        0.00185, 0.00177, 0.00170, 0.00163, 0.00157, 0.00152, 0.00146, 0.00141, 0.00136, 0.00132 };
    int i;
    if (tab_ready) return;
    for (i = 0; i < 256; i++) {
        mblog_tab[i] = (int) floor(1000.0 * log10(1.0 + (i + 0.5) / 256.0) + 0.5) - 38227;
        mbexp_lo[i] = (float) pow(10.0, i / 1000.0);
        mbexp_hi[i] = (float) pow(10.0, ((int) (signed char) i) * 256 / 1000.0);
        pow34_exp[i] = (float) pow(2.0, 0.75 * (i - 127));
    }
    pow34_exp[0] = 0.0f;
    pow34_exp[255] = hxo_bits2f(0x7F800000u);
    for (i = 0; i < 84; i++)
        hxo_logsub_tab[i] = (int) floor(1000.0 * log10(2.0 - pow(10.0, -(16 * i + 8) / 1000.0)));
    for (i = 0; i < 32; i++) hxo_quant_off[i] = (float) q[i] - 0.4375f;
    tab_ready = 1;
}

/* l3math.c:228-242 (IEEE_FLOAT branch): 8 mantissa bits -> table, 301 mB per octave */
int hxo_mblog(float x)
{
    uint32_t u = hxo_f2bits(x);
    return mblog_tab[(u >> 15) & 255] + 301 * (int) (u >> 23);
}

/* l3math.c:342-356 */
float hxo_mbexp(int x)
{
    float t = mbexp_lo[(unsigned) x & 0xff] * mbexp_hi[((unsigned) x & 0xff00) >> 8];
    if (x > 32000) return 1.0E32f;
    if (x < -32000) return 1.0E-32f;
    return t;
}

/* l3math.c:361 */
int hxo_round(float x) { return (int) (x + copysignf(0.5f, x)); }

/* l3math.c:368-382 */
int hxo_logsubber(int n1, int n2)
{
    int k = (n1 - n2) >> 4;
    if (k > 83) k = 83;
    return n1 + hxo_logsub_tab[k];
}

/* l3math.c:142-146 */
float hxo_dblog(float x) { return (float) (10.0 * log10(x)); }

/* pow34.c:132-154 (IEEE_FLOAT branch): piecewise-linear mantissa fit times exponent table */
void hxo_pow34(const float *x, float *y, int n)
{
    int i;
    for (i = 0; i < n; i++) {
        uint32_t u = hxo_f2bits(x[i]);
        float m = hxo_bits2f((u & 0x7FFFFFu) | (127u << 23));
        unsigned seg = (u >> 19) & 15, e = u >> 23;
        y[i] = (m * hxo_bits2f(HX_POW34_B_BITS[seg]) + hxo_bits2f(HX_POW34_A_BITS[seg])) * pow34_exp[e & 255];
    }
}

float hxo_anwin(int n) { return hxo_bits2f(HX_ANWIN_BITS[n]); }

/* ---- ISO Huffman data accessors ---- */
int hxo_huff_dim(int t) { return HX_HUFF_DIM[t]; }
int hxo_huff_linbits(int t) { return HX_HUFF_LINBITS[t]; }
int hxo_huff_code(int t, int x, int y) { return HX_HUFF_CODE[HX_HUFF_OFF[t] + x * HX_HUFF_DIM[t] + y]; }
int hxo_huff_len(int t, int x, int y) { return HX_HUFF_LEN[HX_HUFF_OFF[t] + x * HX_HUFF_DIM[t] + y]; }
int hxo_quada_code(int v) { return HX_QUADA_CODE[v]; }
int hxo_quada_len(int v) { return HX_QUADA_LEN[v]; }

/* ---- scalefactor band edges (l3init.c:56-99): rows 0..2 = ISO 11172-3 Table B.8 (MPEG-1: 44.1, 48,
   32 kHz), rows 3..5 = ISO 13818-3 (MPEG-2 LSF: 22.05, 24, 16 kHz).  Row = sr_index + 3 * (1 - h_id). ---- */
static const short sfb_long[6][23] = {
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576},
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 42, 50, 60, 72, 88, 106, 128, 156, 190, 230, 276, 330, 384, 576},
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 54, 66, 82, 102, 126, 156, 194, 240, 296, 364, 448, 550, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 114, 136, 162, 194, 232, 278, 332, 394, 464, 540, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576}};
static const short sfb_short[6][14] = {
    {0, 4, 8, 12, 16, 22, 30, 40, 52, 66, 84, 106, 136, 192},
    {0, 4, 8, 12, 16, 22, 28, 38, 50, 64, 80, 100, 126, 192},
    {0, 4, 8, 12, 16, 22, 30, 42, 58, 78, 104, 138, 180, 192},
    {0, 4, 8, 12, 18, 24, 32, 42, 56, 74, 100, 132, 174, 192},
    {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 136, 180, 192},
    {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192}};

/* l3init.c:148-172: frequency of the scalefactor band edge nearest to freq */
int hxo_nearest_sf_band_freq(int tix, int samprate, int freq)
{
    int i, f, fout = freq, delta, deltamin = 999999;
    float a = samprate / (2.0f * 576.0f);
    for (i = 0; i < 21; i++) {
        f = (int) (a * sfb_long[tix][i + 1] + 0.5f);
        delta = abs(f - freq);
        if (delta < deltamin) { deltamin = delta; fout = f; }
    }
    return fout;
}
static const int sr_mpeg1[3] = {44100, 48000, 32000};

int hxo_sfb_long_edge(int sr_index, int i) { return sfb_long[sr_index][i]; }
int hxo_sfb_short_edge(int sr_index, int i) { return sfb_short[sr_index][i]; }

/* l3init.c:403-418 / 438-453 */
int hxo_sfbl_limit(int sr_index, int band_limit)
{
    int i;
    for (i = 0; i < 23; i++) if (band_limit <= sfb_long[sr_index][i]) break;
    return i > 21 ? 21 : i;
}
int hxo_sfbs_limit(int sr_index, int band_limit)
{
    int i;
    for (i = 0; i < 14; i++) if (band_limit <= sfb_short[sr_index][i]) break;
    return i > 12 ? 12 : i;
}

/* ---- transform tables (sbt.c:113-131, l3init.c:176-347) ---- */
void hxo_init_transform_tables(hxo_params *p)
{
    static const float Ci[8] = {-0.6f, -0.535f, -0.33f, -0.185f, -0.095f, -0.041f, -0.0142f, -0.0037f};
    double pi = 4.0 * atan(1.0), t;
    int i, j, k, n, q;
    /* 32-point DCT butterfly weights, sizes 16,8,4,2,1 (sbt.c:113-131) */
    for (k = 0, n = 16, i = 0; i < 5; i++, n /= 2)
        for (q = 0; q < n; q++, k++)
            p->dct_coef[k] = (float) (2.0 * cos((pi / (4 * n)) * (2 * q + 1)));
    /* alias butterflies (l3init.c:187-193): Ci*Ci is a float product */
    for (i = 0; i < 8; i++) {
        float c2 = Ci[i] * Ci[i];
        p->csa[0][i] = (float) (1.0 / sqrt(1.0 + c2));
        p->csa[1][i] = (float) (Ci[i] / sqrt(1.0 + c2));
    }
    /* 18-point transform (l3init.c:308-321) */
    t = pi / 72;
    for (q = 0; q < 18; q++) p->m18_w[q] = (float) (2.0 * cos(t * (2 * q + 1)));
    for (q = 0; q < 9; q++) p->m18_w2[q] = (float) (2.0 * cos(2 * t * (2 * q + 1)));
    t = pi / 36;
    for (k = 0; k < 9; k++)
        for (q = 0; q < 4; q++) p->m18_c[k][q] = (float) cos(t * (2 * k) * (2 * q + 1));
    /* 6-point transform (l3init.c:323-347) */
    t = pi / 24;
    for (q = 0; q < 6; q++) p->m6_v[q] = (float) (2.0 * cos(t * (2 * q + 1)));
    for (q = 0; q < 3; q++) p->m6_v2[q] = (float) (2.0 * cos(2 * t * (2 * q + 1)));
    t = pi / 12;
    p->m6_c87 = (float) cos(t * 2 * 1);
    for (q = 0; q < 6; q++) p->m6_v[q] = p->m6_v[q] / 2.0f;
    p->m6_c87 = 2.0f * p->m6_c87;
    /* block-type windows (l3init.c:208-281) */
    {
        float (*w)[36] = p->win;
        for (i = 0; i < 36; i++) w[0][i] = (float) sin(pi / 36 * (i + 0.5));
        for (i = 0; i < 18; i++) w[1][i] = (float) sin(pi / 36 * (i + 0.5));
        for (i = 18; i < 24; i++) w[1][i] = 1.0f;
        for (i = 24; i < 30; i++) w[1][i] = (float) sin(pi / 12 * (i + 0.5 - 18));
        for (i = 30; i < 36; i++) w[1][i] = 0.0f;
        for (i = 0; i < 6; i++) w[3][i] = 0.0f;
        for (i = 6; i < 12; i++) w[3][i] = (float) sin(pi / 12 * (i + 0.5 - 6));
        for (i = 12; i < 18; i++) w[3][i] = 1.0f;
        for (i = 18; i < 36; i++) w[3][i] = (float) sin(pi / 36 * (i + 0.5));
        for (i = 0; i < 12; i++) w[2][i] = (float) sin(pi / 12 * (i + 0.5));
        for (i = 12; i < 36; i++) w[2][i] = 0.0f;
        for (j = 0; j < 4; j++) {
            if (j == 2) continue;
            for (i = 9; i < 36; i++) w[j][i] = -w[j][i];
        }
        for (i = 3; i < 12; i++) w[2][i] = -w[2][i];
        for (j = 0; j < 4; j++) {
            if (j == 2) continue;
            for (i = 0; i < 36; i++) w[j][i] = (1.0f / 9.0f) * w[j][i];
        }
        for (i = 0; i < 36; i++) w[2][i] = (1.0f / 3.0f) * w[2][i];
    }
}

/* ---- psychoacoustic tables (amodini2.c) ---- */
/* amodini2.c:354-364 */
static float f_to_bark(float f)
{
    float t = (1.0f / 1000.0f) * f;
    float tt = (1.0f / 7.5f) * t;
    tt = tt * tt;
    return (float) (13.0 * atan(0.76f * t) + 3.5 * atan(tt));
}

/* amodini2.c:367-385 */
static float interp(const float xy[][2], float x)
{
    int i;
    for (i = 1; i < 100; i++) if (x <= xy[i][0]) break;
    return xy[i - 1][1] + (x - xy[i - 1][0]) * ((xy[i][1] - xy[i - 1][1])) / (xy[i][0] - xy[i - 1][0]);
}

/* amodini2.c:173-196 Painter & Spanias / Schroeder spreading */
static float spread_ps(float bz0, float bz)
{
    double a = 0.2302585093, x, y;
    x = (bz0 - bz) * 1.00;
    x += 0.474;
    y = 15.811389 + 7.5 * x - 17.5 * sqrt(1.0 + x * x);
    if (y <= -60.0) return 0.0f;
    return (float) exp(y * a);
}

/* amodini2.c:203-251 modified P&S used for long blocks */
static float spread_psx(float bz0, float bz)
{
    double a = 0.2302585093, x, y, t1 = 1.2, t2 = 1.2, dt;
    dt = (0.5 / 7.0) * (7.0 - bz0);
    if (dt < 0.0) dt = 0.0;
    t1 = t1 + dt;
    t2 = t2 + dt;
    dt = bz0 - 22.5;
    if (dt < 0.0) dt = 0.0;
    t2 = t2 + dt;
    x = (bz0 - bz);
    if (x > 0.0) x = t1 * x; else x = t2 * x;
    x += 0.474;
    y = 15.811389 + 7.5 * x - 17.5 * sqrt(1.0 + x * x);
    if (y <= -60.0) return 0.0f;
    return (float) exp(y * a);
}

/* one spreading row: evaluate, threshold at 1e-6 keeping the first contiguous run
   (amodini2.c:387-466), then append to w with the row factor */
static int spread_rows(hxo_psytab *pt, float *w, const float *bval, const float *snr_factor,
                       int npart, int is_long)
{
    float s[64], thres = 1.0e-6f;
    int i, j, ntot = 0;
    for (i = 0; i < 64; i++) pt->cnt[i] = pt->off[i] = 0;
    for (i = 0; i < npart; i++) {
        int count = 0, nj;
        float rn;
        for (j = 0; j < 64; j++) s[j] = 0.0f;
        for (j = 0; j < npart; j++) s[j] = is_long ? spread_psx(bval[i], bval[j]) : spread_ps(bval[i], bval[j]);
        /* spread_norm_factor result is unused in both generators except as r_norm = 0.35 (short) */
        rn = 0.35f;
        for (j = 0; j < npart; j++) { if (s[j] > thres) break; s[j] = 0.0f; }
        for (; j < npart; j++) { if (s[j] <= thres) break; }
        for (; j < npart; j++) s[j] = 0.0f;
        for (j = 0; j < npart; j++) if (s[j] != 0.0f) break;
        nj = j;
        if (nj >= npart) break;
        for (; j < npart; j++) {
            if (s[j] == 0.0) break;
            count++; ntot++;
            if (is_long) *w++ = snr_factor[i] * s[j];
            else *w++ = rn * snr_factor[i] * s[j];
        }
        pt->cnt[i] = count;
        pt->off[i] = nj;
    }
    pt->npart = i;
    return ntot;
}

/* amodini2.c:743-940 */
void hxo_init_psy_long(hxo_params *p)
{
    static const float dbsnr[][2] = {
        {0, 0.0f}, {38, 0.0f}, {115, 0.0f}, {191, 0.0f}, {268, 0.0f}, {345, 0.0f}, {421, 0.0f}, {498, 0.0f},
        {574, 0.0f}, {651, 1.0f}, {727, 1.0f}, {804, 2.5f}, {880, 2.5f}, {976, 1.5f}, {1091, 1.5f}, {1206, 2.0f},
        {1321, 2.0f}, {1455, 2.0f}, {1608, 3.0f}, {1761, 3.0f}, {1914, 3.0f}, {2086, 3.0f}, {2278, 3.0f},
        {2488, 1.0f}, {2718, 1.0f}, {2986, 0.0f}, {3292, 0.0f}, {3637, 0.0f}, {4020, 0.0f}, {4441, 0.0f},
        {4900, 0.0f}, {5398, 0.0f}, {5934, 0.0f}, {6527, 0.0f}, {7178, 0.0f}, {7905, 0.0f}, {8709, 0.0f},
        {9589, 0.0f}, {10546, 0.0f}, {11542, 0.0f}, {12575, 0.0f}, {13820, -2.0f}, {15274, -2.0f}, {99999, 0.0f}};
    static const float absthres[][2] = {
        {0.0f, 5.0f}, {350.0f, 0.03f}, {2584.0f, 0.01f}, {5857.0f, 0.01f}, {9302.0f, 0.03f},
        {13092.0f, 0.5f}, {15500.0f, 5.0f}, {99999.0f, 100.0f}};
    hxo_psytab *pt = &p->psyL;
    int part[64], i, t, m, nbin, npart, ntot;
    float snr_factor[64], bval[64], athres[64], x, freq;
    memset(pt, 0, sizeof(*pt));
    memset(athres, 0, sizeof(athres));
    for (i = 0; i < 64; i++) part[i] = 576;
    for (t = 0, i = 0; i < 22; i++) {       /* partitions = each sfb split in two */
        int nb = p->nBand_l_iso[i];
        part[2 * i] = t; m = nb / 2; t += m;
        part[2 * i + 1] = t; m = nb - m; t += m;
    }
    nbin = 18 * p->nsb_limit;
    for (i = 0; i < 64; i++) if (part[i] >= nbin) break;
    npart = i;
    if (npart > 2 * 21) npart = 2 * 21;
    x = 0.5f * p->samprate / 576;
    for (i = 0; i < 63; i++) {
        freq = x * 0.5f * (part[i] + part[i + 1]);
        snr_factor[i] = (float) pow(10.0, -0.1 * interp(dbsnr, freq));
        bval[i] = f_to_bark(freq);
        athres[i] = interp(absthres, freq) * (part[i + 1] - part[i]);
    }
    snr_factor[i] = 1.0f;
    bval[i] = bval[i - 1];
    ntot = spread_rows(pt, pt->w + 128, bval, snr_factor, npart, 1);
    assert(ntot <= 2200 - 128);
    for (i = 128; i < ntot + 128; i++)      /* power-law (alpha = 0.3) summation */
        if (pt->w[i] > 0.0f) pt->w[i] = (float) pow(pt->w[i], 0.30);
    for (i = 0; i < 64; i++) pt->w[i] = athres[i];
    for (i = 0; i < npart; i++) pt->nsum[i] = part[i + 1] - part[i];
    pt->npart_e = npart;
}

/* amodini2.c:587-739 */
void hxo_init_psy_short(hxo_params *p)
{
    static const float dbsnr[][2] = {
        {0.0f, 12.0f}, {861.0f, 10.0f}, {2584.0f, 8.0f}, {5857.0f, 7.0f}, {9302.0f, 5.0f},
        {13092.0f, 4.0f}, {15500.0f, 3.0f}, {99999.0f, -2.0f}};
    hxo_psytab *pt = &p->psyS;
    int part[32], i, t, m, nbin, npart, ntot;
    float snr_factor[32], bval[32], x, freq;
    memset(pt, 0, sizeof(*pt));
    for (i = 0; i < 32; i++) part[i] = 192;
    for (t = 0, i = 0; i < 14; i++) {
        int nb = (i < 13) ? p->nBand_s[i] : 0;
        part[2 * i] = t; m = nb / 2; t += m;
        part[2 * i + 1] = t; m = nb - m; t += m;
    }
    nbin = 6 * p->nsb_limit;
    for (i = 0; i < 32; i++) if (part[i] >= nbin) break;
    npart = i;
    if (npart > 2 * 12) npart = 2 * 12;
    x = 0.5f * p->samprate / 192;
    for (i = 0; i < 31; i++) {
        freq = x * 0.5f * (part[i] + part[i + 1]);
        snr_factor[i] = (float) ((p->h_id ? 0.7 : 2.8) * pow(10.0, -0.1 * interp(dbsnr, freq)));      /* amodini2.c:678-690 */
        bval[i] = f_to_bark(freq);
    }
    snr_factor[i] = 1.0f;
    bval[i] = bval[i - 1];
    ntot = spread_rows(pt, pt->w, bval, snr_factor, npart, 0);
    assert(ntot <= 1000);
    for (i = 0; i < npart; i++) pt->nsum[i] = part[i + 1] - part[i];
    pt->npart_e = npart;
}

int hxo_samprate_mpeg1(int sr_index) { return sr_mpeg1[sr_index]; }
