// hx_host.cpp - host side of the MI355X batched MP3 encoder: resolves an E_CONTROL into the
// per-class parameter/table block the kernels consume.  Mirrors the behaviour of
// CMp3Enc::L3_audio_encode_init (reference mp3enc.cpp:220-870), setup_header (setup.c:189),
// CBitAllo3::BitAlloInit (bitallo3.cpp:288), L3table_init (l3init.c:176) and amod_initLong
// (amodini2.c:743).  Table values are generated with the same double-precision libm
// expressions the reference uses so that the float casts agree bit for bit.
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "hx_types.h"
#include "hx_host.h"
#include "iso_data.inc"

#define MX(a, b) ((a) > (b) ? (a) : (b))
#define MN(a, b) ((a) < (b) ? (a) : (b))

static float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Scalefactor band edges.  Rows 0..2: ISO 11172-3 Table B.8, MPEG-1 (44.1, 48, 32 kHz); rows 3..5:
// ISO 13818-3, MPEG-2 LSF (22.05, 24, 16 kHz) as the reference carries them (l3init.c:56-99).
// Row = HxParams::tix = sr_index + 3 * (1 - h_id).
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
static const int br_mpeg1_l3[16] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, -1};
static const int br_mpeg2_l3[16] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, -1};

// frequency of the scalefactor band edge nearest to freq (l3init.c:148-172)
static int nearest_sf_band_freq(int tix, int samprate, int freq)
{
    int fout = freq, deltamin = 999999;
    float a = samprate / (2.0f * 576.0f);
    for (int i = 0; i < 21; i++) {
        int f = (int) (a * sfb_long[tix][i + 1] + 0.5f), delta = abs(f - freq);
        if (delta < deltamin) { deltamin = delta; fout = f; }
    }
    return fout;
}

void hx_host_default_control(HxControl *ec)
{
    memset(ec, 0, sizeof(*ec));
    ec->mode = 1; ec->bitrate = -1; ec->samprate = 44100; ec->nsbstereo = -1; ec->filter_select = -1;
    ec->freq_limit = 24000; ec->nsb_limit = -1; ec->layer = 3; ec->cr_bit = 1; ec->original = 1;
    ec->vbr_flag = 1; ec->vbr_mnr = 50; ec->vbr_br_limit = 160;
    ec->chan_add_f0 = ec->chan_add_f1 = 24000; ec->sparse_scale = -1;
    ec->quick = -1; ec->test1 = -1; ec->short_block_threshold = 700;
}

void hx_global_tabs(HxGlobalTabs *g)
{
    static const double q[32] = {
        0.09460, 0.02799, 0.01671, 0.01192, 0.00927, 0.00758, 0.00641, 0.00556, 0.00490, 0.00439, 0.00397,
        0.00362, 0.00333, 0.00309, 0.00287, 0.00269, 0.00253, 0.00238, 0.00225, 0.00214, 0.00203, 0.00194,
        0.00185, 0.00177, 0.00170, 0.00163, 0.00157, 0.00152, 0.00146, 0.00141, 0.00136, 0.00132};
    memset(g, 0, sizeof(*g));
    for (int i = 0; i < HX_POW43_N; i++) g->pow43[i] = pow((double) i, (4.0 / 3.0));
    for (int i = 0; i < 512; i++) g->anwin[i] = bits2f(HX_ANWIN_BITS[i]);
    // K1 folds the 512 taps to 32 sums; output k adds taps A + 64 j and B + 64 j (j = 0..7)
    for (int k = 0; k < 32; k++) {
        const int A = (k == 0) ? 16 : ((k <= 16) ? 16 + k : 80 - k), B = (k == 0) ? 16 : ((k <= 16) ? 16 - k : 16 + k);
        for (int j = 0; j < 8; j++) { g->anwin_r[16 * k + 2 * j] = g->anwin[A + 64 * j]; g->anwin_r[16 * k + 2 * j + 1] = g->anwin[B + 64 * j]; }
    }
    for (int i = 0; i < 256; i++) {
        g->mblog[i] = (int) floor(1000.0 * log10(1.0 + (i + 0.5) / 256.0) + 0.5) - 38227;
        g->mbexp_lo[i] = (float) pow(10.0, i / 1000.0);
        g->mbexp_hi[i] = (float) pow(10.0, ((int) (signed char) i) * 256 / 1000.0);
        g->pow34_exp[i] = (float) pow(2.0, 0.75 * (i - 127));
    }
    g->pow34_exp[0] = 0.0f;
    g->pow34_exp[255] = bits2f(0x7F800000u);
    for (int i = 0; i < 16; i++) { g->pow34_a[i] = bits2f(HX_POW34_A_BITS[i]); g->pow34_b[i] = bits2f(HX_POW34_B_BITS[i]); }
    for (int i = 0; i < 32; i++) g->quant_off[i] = (float) q[i] - 0.4375f;
    for (int i = 0; i < 84; i++) g->logsub[i] = (int) floor(1000.0 * log10(2.0 - pow(10.0, -(16 * i + 8) / 1000.0)));
    int n = (int) (sizeof(HX_HUFF_CODE) / sizeof(HX_HUFF_CODE[0]));
    for (int i = 0; i < n; i++) { g->huff_code[i] = HX_HUFF_CODE[i]; g->huff_len[i] = HX_HUFF_LEN[i]; }
    for (int i = 0; i < 32; i++) { g->huff_off[i] = HX_HUFF_OFF[i]; g->huff_dim[i] = (unsigned char) HX_HUFF_DIM[i]; g->huff_lin[i] = HX_HUFF_LINBITS[i]; }
    for (int i = 0; i < 16; i++) { g->quada_code[i] = HX_QUADA_CODE[i]; g->quada_len[i] = HX_QUADA_LEN[i]; }
}

static int sfbl_limit(int sr, int band_limit)
{
    int i;
    for (i = 0; i < 23; i++) if (band_limit <= sfb_long[sr][i]) break;
    return i > 21 ? 21 : i;
}

// ---- transform constants ---------------------------------------------------------------------------
// Each value is formed in double from its closed form and rounded to float once; products of angles are
// taken left to right as written, so the float tables equal the reference encoder's bit for bit
// (tests/test_host_and_abi.py compares them with the oracle's).
static const double kPi = 4.0 * atan(1.0);

// ISO 11172-3 2.4.3.4.10.3 block window of type bt (0 normal, 1 start, 2 short, 3 stop) at n = 0..35
static float iso_block_window(int bt, int n)
{
    const double lng = sin(kPi / 36 * (n + 0.5));
    switch (bt) {
    case 0: return (float) lng;
    case 1: return n < 18 ? (float) lng : (n < 24 ? 1.0f : (n < 30 ? (float) sin(kPi / 12 * (n + 0.5 - 18)) : 0.0f));
    case 3: return n < 6 ? 0.0f : (n < 12 ? (float) sin(kPi / 12 * (n + 0.5 - 6)) : (n < 18 ? 1.0f : (float) lng));
    default: return n < 12 ? (float) sin(kPi / 12 * (n + 0.5)) : 0.0f;
    }
}

// input twiddle 2 cos(pi (2q+1) / 4N) of an N-point MDCT kernel, and the odd half's 2 cos(pi (2q+1) / 2N)
static float mdct_twiddle(int N, int q, int doubled)
{
    const double t = kPi / (4 * N);
    return (float) (2.0 * cos((doubled ? 2 * t : t) * (2 * q + 1)));
}
// cos(pi k (2q+1) / N): entry (k, q) of the N/2-point cosine transform, k even numbered as 2k' in the kernel
static float dct_entry(int N, int k2, int q)
{
    const double t = kPi / (2 * N);
    return (float) cos(t * k2 * (2 * q + 1));
}

static void transform_tables(HxParams *p)
{
    // analysis filterbank: twiddles of the 32-point DCT's five recursion levels
    p->dct_tw[0] = 0.0f;
    for (int N = 2; N <= 32; N *= 2)
        for (int j = 0; j < N / 2; j++) p->dct_tw[N / 2 + j] = (float) (2.0 * cos((kPi / (2 * N)) * (2 * j + 1)));
    // alias reduction: cs = 1 / sqrt(1 + c^2), ca = c / sqrt(1 + c^2) for the eight ISO butterfly constants
    static const float c_iso[8] = {-0.6f, -0.535f, -0.33f, -0.185f, -0.095f, -0.041f, -0.0142f, -0.0037f};
    for (int i = 0; i < 8; i++) {
        const float c = c_iso[i], cc = c * c;
        p->csa[0][i] = (float) (1.0 / sqrt(1.0 + cc));
        p->csa[1][i] = (float) (c / sqrt(1.0 + cc));
    }
    // block windows with the sign pattern of the 36 -> 18 (12 -> 6) fold and the 1/9 (1/3) gain of the kernel folded in
    for (int bt = 0; bt < 4; bt++)
        for (int n = 0; n < 36; n++) {
            const float w = iso_block_window(bt, n);
            const bool neg = (bt == 2) ? (n >= 3 && n < 12) : (n >= 9);
            p->win[bt][n] = (bt == 2 ? 1.0f / 3.0f : 1.0f / 9.0f) * (neg ? -w : w);
        }
    // MDCT kernels
    for (int q = 0; q < 18; q++) p->mdct_pre18[q] = mdct_twiddle(18, q, 0);
    for (int q = 0; q < 9; q++) p->mdct_odd18[q] = mdct_twiddle(18, q, 1);
    static const int even_rows[3] = {2, 4, 8}, odd_rows[3] = {1, 5, 7};
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 4; q++) {
            p->dct9_even[r][q] = dct_entry(18, 2 * even_rows[r], q);
            p->dct9_odd[r][q] = dct_entry(18, 2 * odd_rows[r], q);
        }
    p->dct9_k3 = dct_entry(18, 2 * 3, 0);
    for (int q = 0; q < 6; q++) p->mdct_pre6[q] = mdct_twiddle(6, q, 0) / 2.0f;     // the short kernel carries half the gain here ...
    for (int q = 0; q < 3; q++) p->mdct_odd6[q] = mdct_twiddle(6, q, 1);
    p->dct3_k = 2.0f * dct_entry(6, 2, 0);                                           // ... and twice on its middle output
}

static float f_to_bark(float f)
{
    float t = (1.0f / 1000.0f) * f, tt = (1.0f / 7.5f) * t;
    tt = tt * tt;
    // double-precision atan on the promoted float arguments, as the C reference does
    // (in C++ a bare atan(float) would select atanf)
    return (float) (13.0 * atan((double) (0.76f * t)) + 3.5 * atan((double) tt));
}

static float interp(const float xy[][2], float x)
{
    int i;
    for (i = 1; i < 100; i++) if (x <= xy[i][0]) break;
    return xy[i - 1][1] + (x - xy[i - 1][0]) * ((xy[i][1] - xy[i - 1][1])) / (xy[i][0] - xy[i - 1][0]);
}

// modified Schroeder / Painter-Spanias spreading for long blocks (amodini2.c:203-251)
static float spread_long(float bz0, float bz)
{
    double a = 0.2302585093, x, y, t1 = 1.2, t2 = 1.2, dt;
    dt = (0.5 / 7.0) * (7.0 - bz0);
    if (dt < 0.0) dt = 0.0;
    t1 = t1 + dt; t2 = t2 + dt;
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

static void psy_long_tables(HxParams *p)
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
    HxPsyTab *pt = &p->psyL;
    int part[64], t = 0, npart, ntot = 0;
    float snr_factor[64], bval[64], athres[64], s[64];
    memset(pt, 0, sizeof(*pt));
    memset(athres, 0, sizeof(athres));
    for (int i = 0; i < 64; i++) part[i] = 576;
    for (int i = 0; i < 22; i++) {      // two partitions per scalefactor band
        int nb = p->nBand_l_iso[i], m = nb / 2;
        part[2 * i] = t; t += m;
        part[2 * i + 1] = t; t += nb - m;
    }
    int nbin = 18 * p->nsb_limit;
    for (npart = 0; npart < 64; npart++) if (part[npart] >= nbin) break;
    if (npart > 42) npart = 42;
    float x = 0.5f * p->samprate / 576;
    int i;
    for (i = 0; i < 63; i++) {
        float freq = x * 0.5f * (part[i] + part[i + 1]);
        snr_factor[i] = (float) pow(10.0, -0.1 * interp(dbsnr, freq));
        bval[i] = f_to_bark(freq);
        athres[i] = interp(absthres, freq) * (part[i + 1] - part[i]);
    }
    snr_factor[i] = 1.0f;
    bval[i] = bval[i - 1];
    float *w = pt->w + 128;
    for (i = 0; i < npart; i++) {
        int j, count = 0, nj;
        for (j = 0; j < 64; j++) s[j] = 0.0f;
        for (j = 0; j < npart; j++) s[j] = spread_long(bval[i], bval[j]);
        for (j = 0; j < npart; j++) { if (s[j] > 1.0e-6f) break; s[j] = 0.0f; }
        for (; j < npart; j++) if (s[j] <= 1.0e-6f) break;
        for (; j < npart; j++) s[j] = 0.0f;
        for (j = 0; j < npart; j++) if (s[j] != 0.0f) break;
        nj = j;
        if (nj >= npart) break;
        pt->row[i] = 128 + ntot;
        for (; j < npart; j++) {
            if (s[j] == 0.0f) break;
            count++; ntot++;
            *w++ = snr_factor[i] * s[j];
        }
        pt->cnt[i] = count;
        pt->off[i] = nj;
    }
    pt->npart = i;
    for (i = 128; i < ntot + 128; i++) if (pt->w[i] > 0.0f) pt->w[i] = (float) pow((double) pt->w[i], 0.30);
    for (i = 0; i < 64; i++) pt->w[i] = athres[i];
    for (i = 0; i < npart; i++) pt->nsum[i] = part[i + 1] - part[i];
    pt->npart_e = npart;
    for (i = 0, t = 0; i < 64; i++) { pt->pstart[i] = t; t += pt->nsum[i]; }
    pt->pstart[64] = t;
}

// Painter & Spanias / Schroeder spreading, short blocks (amodini2.c:173-196)
static float spread_short(float bz0, float bz)
{
    double a = 0.2302585093, x, y;
    x = (bz0 - bz) * 1.00;
    x += 0.474;
    y = 15.811389 + 7.5 * x - 17.5 * sqrt(1.0 + x * x);
    if (y <= -60.0) return 0.0f;
    return (float) exp(y * a);
}

// amod_initShort (amodini2.c:587-739): 2 partitions per short sfb on the 192-line window
static void psy_short_tables(HxParams *p)
{
    static const float dbsnr[][2] = {
        {0.0f, 12.0f}, {861.0f, 10.0f}, {2584.0f, 8.0f}, {5857.0f, 7.0f}, {9302.0f, 5.0f},
        {13092.0f, 4.0f}, {15500.0f, 3.0f}, {99999.0f, -2.0f}};
    HxPsyTab *pt = &p->psyS;
    int part[32], t = 0, npart, ntot = 0, i;
    float snr_factor[32], bval[32], s[64];
    memset(pt, 0, sizeof(*pt));
    for (i = 0; i < 32; i++) part[i] = 192;
    for (i = 0; i < 14; i++) {
        int nb = (i < 13) ? p->nBand_s[i] : 0, m = nb / 2;
        part[2 * i] = t; t += m;
        part[2 * i + 1] = t; t += nb - m;
    }
    int nbin = 6 * p->nsb_limit;
    for (npart = 0; npart < 32; npart++) if (part[npart] >= nbin) break;
    if (npart > 24) npart = 24;
    float x = 0.5f * p->samprate / 192;
    for (i = 0; i < 31; i++) {
        float freq = x * 0.5f * (part[i] + part[i + 1]);
        snr_factor[i] = (float) ((p->h_id ? 0.7 : 2.8) * pow(10.0, -0.1 * interp(dbsnr, freq)));     // amodini2.c:678-690
        bval[i] = f_to_bark(freq);
    }
    snr_factor[i] = 1.0f;
    bval[i] = bval[i - 1];
    float *w = pt->w;
    for (i = 0; i < npart; i++) {
        int j, count = 0, nj;
        for (j = 0; j < 64; j++) s[j] = 0.0f;
        for (j = 0; j < npart; j++) s[j] = spread_short(bval[i], bval[j]);
        for (j = 0; j < npart; j++) { if (s[j] > 1.0e-6f) break; s[j] = 0.0f; }
        for (; j < npart; j++) if (s[j] <= 1.0e-6f) break;
        for (; j < npart; j++) s[j] = 0.0f;
        for (j = 0; j < npart; j++) if (s[j] != 0.0f) break;
        nj = j;
        if (nj >= npart) break;
        pt->row[i] = ntot;
        for (; j < npart; j++) {
            if (s[j] == 0.0f) break;
            count++; ntot++;
            *w++ = 0.35f * snr_factor[i] * s[j];
        }
        pt->cnt[i] = count;
        pt->off[i] = nj;
    }
    pt->npart = i;
    for (i = 0; i < npart; i++) pt->nsum[i] = part[i + 1] - part[i];
    pt->npart_e = npart;
    for (i = 0, t = 0; i < 64; i++) { pt->pstart[i] = t; t += pt->nsum[i]; }
    pt->pstart[64] = t;
}

// Returns 0 when the configuration is outside what the MI355X path implements (the reference
// would run dual-channel or intensity stereo there) or when the reference itself
// rejects it (mp3enc.cpp:346-351,388); 9216 (bytes of float PCM per frame) otherwise.
// Tables of the first-generation allocator (reference bitallo1.cpp:107-211 BitAlloInit, :444-543): expected
// quantisation noise as a function of the quantised value (closed form of the x^(4/3) quantiser's cell), a bit
// estimate per line by the band's largest value, the intensity position by the channels' energy ratio.
static void alloc1_tables(HxParams *p)
{
    for (int i = 0; i < 21; i++) p->a1_log_cbw[i] = (float) (10.0 * log10((double) p->nBand_l_iso[i]));
    for (int big = 0; big < 2; big++) {
        double cum = 0.0;
        for (int i = 0; i < 256; i++) {
            const int ix = big ? 32 * i + 16 : i;
            double t = ix + 0.5;
            const double xh = t * pow(t, 1.0 / 3.0);
            t = ix;
            const double x0 = t * pow(t, 1.0 / 3.0);
            t = ix - 0.5;
            const double xl = t * pow(fabs(t), 1.0 / 3.0);
            const double dh = xh - x0, dl = xl - x0;
            const double eps = (dh * dh * dh - dl * dl * dl) / (3.0 * (xh - xl));
            cum += eps;
            (big ? p->a1_f_big_ix : p->a1_f_ix)[i] = (float) eps;
            (big ? p->a1_f_big_ixmax : p->a1_f_ixmax)[i] = (float) (10.0 * log10(cum / (i + 1)));
        }
    }
    p->a1_bits[0] = 0;
    for (int m = 1; m < 256; m++) p->a1_bits[m] = (int) (16 * (1.4427 * log((double) (m + 1)) + (m - 0.6) / m));
    p->a1_gz1 = (float) (16.0 / (3.0 * log(2.0)));
    p->a1_gz2 = (float) (1 - (16.0 / (3.0 * log(2.0))) * log(.5946) + 8);
    p->a1_gz0 = (float) (exp((0.99 - p->a1_gz2) / p->a1_gz1));
    if (p->h_id) {
        const double pi = 4.0 * atan(1.0);
        for (int i = 0; i < 34; i++) p->a1_is_pos[i] = (int) (((12.0 / pi) * atan(sqrt(i / 32.0))) + .25);
    } else
        for (int i = 0; i < 34; i++) {
            const int k = (int) (-log((i + .0001) / 32.0) / log(2.0) + 0.5);
            p->a1_is_pos[i] = 2 * MN(MX(k, 0), 3);
        }
    p->a1_ill_is_pos = p->h_id ? 7 : 999;
    p->a1_c707 = (float) (1.0 / sqrt(2.0));
    static const float sp1[21] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f,
                                  0.20f, 0.30f, 0.40f, 0.50f, 0.60f, 0.70f, 0.80f, 0.90f, 1.0f, 1.5f};
    static const float sp2[21] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f, 0.10f,
                                  0.20f, 0.30f, 0.40f, 0.50f, 0.50f, 0.60f, 0.70f, 0.80f, 0.90f};
    memcpy(p->a1_sparse, p->h_id ? sp1 : sp2, sizeof(p->a1_sparse));
}

// Blocked line runs for the stream walk's certified band sums (hx_alloc.hip: noise_sweep, inverse_sf2).
// The bands 0 .. nb-1 are cut into runs of at most W consecutive lines, one run per lane, a band's runs in neighbouring
// lanes; W = the smallest even width with which the runs fit the 64 lanes and no band takes more than maxl <= 16 of them (the
// segmented scan over a band's lanes crosses at most one 16-lane row boundary).  Returns W, 0 if there is none in wmin .. wmax.
static int cut_runs(const int *width, const int *first, int nb, int wmin, int wmax, int maxl, int lane0, int nlanes, unsigned short *run, unsigned char *last)
{
    for (int W = wmin; W <= wmax; W += 2) {
        int lanes = 0, ok = 1;
        for (int b = 0; b < nb; b++) {
            const int c = (width[b] + W - 1) / W;
            if (c > maxl) ok = 0;
            lanes += c;
            // The kernels read a lane's operands as W / 2 pairs from the run's first line whatever its length (sweep_load,
            // sweep_run_stored, isf2_run, msmetric_unit drop what lies past the run): the last run of a band may start up to
            // W - 2 lines before the band's end, so every run must satisfy start + W <= 576 to stay inside the channel's array.
            if (c > 0 && first[b] + (c - 1) * W + W > 576) ok = 0;
        }
        if (!ok || lanes > nlanes) continue;
        int l = lane0;
        for (int b = 0; b < nb; b++) {
            const int c = (width[b] + W - 1) / W;
            for (int k = 0; k < c; k++, l++) {
                const int st = first[b] + k * W, n = MN(W, width[b] - k * W);
                run[l] = (unsigned short) ((st >> 1) | ((n >> 1) << 9) | (k << 12));
            }
            last[b] = (unsigned char) (l - 1);
        }
        return W;
    }
    return 0;
}

// Why the last hx_resolve of this thread returned 0 although the reference would accept the configuration: a limit of this
// library's kernel layout, not of the format ("" = the reference rejects it too).  hx_batch_create / hx_enc_* put it into
// hx_last_error().
static thread_local const char *g_resolve_why = "";
const char *hx_resolve_error(void) { return g_resolve_why; }

static int band_runs(HxParams *p)
{
    memset(p->lane_run, 0, sizeof(p->lane_run));
    memset(p->band_last_lane, 0, sizeof(p->band_last_lane));
    // (every long band starts on an even line and has an even width: ISO Table B.8; the packed form relies on it)
    for (int b = 0; b < 22; b++) if ((p->startBand_l[b] | p->nBand_l[b]) & 1) {
        g_resolve_why = "kernel layout limit: a long scalefactor band starts on an odd line or has an odd width (line runs are stored as pairs)";
        return 0;
    }
    const int nb = MX(p->nsf[0], p->nchan == 2 ? p->nsf[1] : 0);
    p->run_w = cut_runs(p->nBand_l, p->startBand_l, nb, 2, 10, 16, 0, 64, p->lane_run, p->band_last_lane);
    if (nb > 0 && !p->run_w) {
        // (32 kHz with all 21 bands measured takes exactly 64 of the 64 lanes at W = 10: tests/test_cert_sums.py pins it)
        g_resolve_why = "kernel layout limit: the measured scalefactor bands do not fit 64 lanes in runs of at most 10 lines with at most 16 lanes per band "
                        "(hx_host.cpp band_runs; the certified band sums of the stream walk need one run per lane)";
        return 0;
    }
    return 1;
}

int hx_resolve(const HxControl *ec_arg, HxParams *p)
{
    static const int sr_all[8] = {22050, 24000, 16000, 1, 44100, 48000, 32000, 1};
    static const int mnrGOLD[22] = {-5, 0, 0, 0, 0, 0, 0, 0, 0, 3, 5, 5, 5, 5, 3, 0, 0, 0, -1, -8, -10, 0};
    HxControl ec = *ec_arg;
    memset(p, 0, sizeof(*p));
    g_resolve_why = "";
    if (ec.mode < 0) ec.mode = 1;
    if (ec.mode > 3) ec.mode = 3;
    if (ec.bitrate < 0) { ec.bitrate = 64; if (ec.samprate < 32000) ec.bitrate = 32; }
    if (ec.mode == 2) ec.vbr_flag = 0;
    ec.vbr_mnr = MN(MX(ec.vbr_mnr, 0), 150);
    if (ec.mode != 1) ec.nsbstereo = 0;
    if (ec.vbr_flag) ec.nsbstereo = 0;
    if (ec.mode == 2) ec.hf_flag = 0;
    if (ec.vbr_flag == 0) { if (ec.bitrate < 96) ec.hf_flag = 0; }
    else { if (ec.vbr_mnr < 80) ec.hf_flag = 0; }
    if (ec.samprate < 44100) ec.hf_flag = 0;
    if (ec.filter_select < 0) ec.filter_select = 0;
    if ((ec.vbr_flag == 0) && (ec.samprate > 24000) && (ec.bitrate < 48)) return 0;
    ec.cr_bit &= 1;
    ec.original &= 1;
    if (ec.samprate > 32000) { if (ec.bitrate < 24) ec.bitrate = 24; }
    else if (ec.samprate > 24000) { if (ec.bitrate < 16) ec.bitrate = 16; }
    else if (ec.samprate > 16000) { if (ec.bitrate < 12) ec.bitrate = 12; }
    else { if (ec.bitrate < 8) ec.bitrate = 8; }
    p->short_block_threshold = ec.short_block_threshold;
    if (ec.layer != 3) return 0;

    int k = 0, dmin = 99999;
    for (int i = 0; i < 8; i++) { int d = abs(ec.samprate - sr_all[i]); if (d < dmin) { dmin = d; k = i; } }
    if (sr_all[k] == 1) return 0;
    const int h_id = k >> 2;                        // 1 = MPEG-1, 0 = MPEG-2 LSF (16 / 22.05 / 24 kHz)
    p->h_id = h_id;
    p->sr_index = k & 3;
    p->tix = p->sr_index + 3 * (1 - h_id);
    const int *br_tab = h_id ? br_mpeg1_l3 : br_mpeg2_l3;
    p->h_mode = ec.mode;
    p->nchan = (ec.mode == 3) ? 1 : 2;
    int mode_ext = (p->h_mode == 1) ? ec.nsbstereo / 4 - 1 : 0;
    if (mode_ext < 0) mode_ext = h_id ? 0 : 1;      // setup.c:234-239 (every frame rewrites these two bits)
    mode_ext = MN(mode_ext, 3);
    int bitrate = MX(ec.bitrate, 8) * p->nchan;     // per-channel request; a mono frame carries one channel (setup.c:254-258)
    if (bitrate > (h_id ? 320 : 160)) bitrate = h_id ? 320 : 160;
    int br_index = 0;
    for (int i = 1; br_tab[i] >= 0; i++) if (br_tab[i] == bitrate) br_index = i;
    p->totbitrate = bitrate;
    HxMpegHead *h = &p->head_info;
    h->sync = 1; h->id = h_id; h->option = 1; h->prot = 1; h->br_index = br_index; h->sr_index = p->sr_index;
    h->mode = p->h_mode; h->mode_ext = mode_ext; h->cr = ec.cr_bit; h->original = ec.original;
    p->head[0] = 0xFF;
    p->head[1] = (unsigned char) (0xF3 | (h_id << 3));
    p->head[2] = (unsigned char) ((br_index << 4) | (p->sr_index << 2));
    p->head[3] = (unsigned char) ((p->h_mode << 6) | (mode_ext << 4) | (ec.cr_bit << 3) | (ec.original << 2));

    p->nband = sfb_long[p->tix][21];
    p->nsb = (p->nband + 17) / 18;
    int nsbstereo = h_id ? 12 * p->totbitrate / 32 - 20 : 7 * p->totbitrate / 16 - 7;     // mp3enc.cpp:406-422
    nsbstereo = MX(MN(nsbstereo, 32), 3);
    if (p->totbitrate >= (h_id ? 96 : 48)) nsbstereo = 32;
    if (ec.vbr_flag) nsbstereo = 32;
    if (ec.nsbstereo > 0) nsbstereo = MN(MX(ec.nsbstereo, 3), 32);
    if (nsbstereo > p->nsb) nsbstereo = p->nsb;
    p->samprate = sr_all[4 * h_id + p->sr_index];
    p->divisor = p->samprate;
    // an MPEG-2 frame is one granule: half the bytes, and the side info shrinks to 17 / 9 (mp3enc.cpp:447-481)
    const int fcoef = h_id ? 144000 : 72000;
    p->framebytes = fcoef * p->totbitrate / p->divisor;
    p->remainder = (fcoef * p->totbitrate) % p->divisor;
    p->side_bytes = h_id ? ((p->h_mode == 3) ? 17 : 32) : ((p->h_mode == 3) ? 9 : 17);
    p->main_framebytes = p->framebytes - 4 - p->side_bytes;
    p->sf_bit_max = 3 * (6 * 4 + 6 * 3);
    p->AveTargetBits = 8 * p->main_framebytes / (h_id ? 2 : 1);     // bits per granule ...
    if (p->h_mode != 3) p->AveTargetBits >>= 1;             // ... and per channel
    p->AveTargetBits -= p->sf_bit_max;

    int nsb_user_flag = 0, u1 = 32, u2 = 32, freq_limit;
    if (ec.nsb_limit > 0) {
        u1 = MN(ec.nsb_limit, 32);
        u1 = MX(ec.nsb_limit, (64 * 1000 + p->samprate / 2) / p->samprate);
        nsb_user_flag = 1;
    }
    if (ec.freq_limit < 24000) {
        u2 = (64 * MX(ec.freq_limit, 1000) + p->samprate / 2) / p->samprate;
        nsb_user_flag = 1;
    }
    int nsb_limit_user = MN(u1, u2);
    if (ec.vbr_flag) {
        freq_limit = h_id ? 12000 + 80 * ec.vbr_mnr : 7500 + 50 * ec.vbr_mnr;      // mp3enc.cpp:505-533
        if (ec.vbr_mnr <= 5) freq_limit = h_id ? 12000 : 7500;
        freq_limit = MN(freq_limit, ((int) ((0.96f * 0.5f) * p->samprate)));
    } else {
        static const float factor[4] = {1.1f, 1.333f, 1.0f, 1.0f};
        float chan_bitrate = (float) p->totbitrate;
        if (p->h_mode != 3) chan_bitrate = (float) (0.5 * chan_bitrate);
        chan_bitrate = factor[p->h_mode] * chan_bitrate;
        if (p->samprate < 32000) {                  // calc_freq_limit_L3, low rates (mp3enc.cpp:899-935)
            if (chan_bitrate <= 32.0f) freq_limit = (int) (752.0 + 203.0 * chan_bitrate);
            else if (chan_bitrate <= 42.7f) freq_limit = (int) (-2967.0 + 327.0 * chan_bitrate);
            else freq_limit = 11000;
        } else freq_limit = (int) (187.97 * chan_bitrate);
    }
    if (h_id == 0) {                                // snap to a scalefactor band edge, then to a subband (mp3enc.cpp:526-545)
        freq_limit = nearest_sf_band_freq(p->tix, p->samprate, freq_limit);
        int tmp = (64 * freq_limit + (p->samprate / 2)) / p->samprate;
        freq_limit = (tmp * p->samprate) / 64;
    }
    p->nsb_limit = nsb_user_flag ? nsb_limit_user : (64 * MX(freq_limit, 1000) + p->samprate / 2) / p->samprate;
    p->nsb_limit = MN(p->nsb, p->nsb_limit);
    p->nsb_ms0 = p->nsb_ms1 = p->nsb_limit;
    if (p->nsb_limit < p->nsb) ec.hf_flag = 0;
    if (ec.hf_flag) { p->nsb_ms0 = 29; if (nsb_user_flag) p->nsb_ms0 = MN(nsb_limit_user, 29); }
    if (ec.hf_flag & 2) { p->nsb_ms1 = 29; if (nsb_user_flag) p->nsb_ms1 = MN(nsb_limit_user, 29); }
    p->band_limit = MN(18 * p->nsb_limit, p->nband);
    int nsbstereo_limit = MN(nsbstereo, p->nsb_limit);
    p->band_limit_stereo = (p->h_mode == 1) ? 18 * nsbstereo_limit : p->band_limit;
    if (p->band_limit_stereo > p->band_limit) p->band_limit_stereo = p->band_limit;

    p->filter_alpha = (float) (0.001 * 44100.0 / p->samprate);
    p->filter_dc = ec.filter_select > 1 ? 1 : ec.filter_select;

    for (int i = 0; i < 22; i++) p->nBand_l_iso[i] = p->nBand_l[i] = sfb_long[p->tix][i + 1] - sfb_long[p->tix][i];
    for (int i = 0; i < 13; i++) p->nBand_s[i] = sfb_short[p->tix][i + 1] - sfb_short[p->tix][i];
    transform_tables(p);
    psy_long_tables(p);
    psy_short_tables(p);

    int is_flag = 0;
    p->ms_flag = 0;
    if (p->h_mode == 1) { if (nsbstereo_limit < p->nsb_limit) is_flag = 1; p->ms_flag = 1; }
    // the reference's first-generation allocator codes these (mp3enc.cpp:696-766): long blocks only
    p->is_flag = is_flag;
    p->alloc1 = is_flag || p->h_mode == 2;
    if (is_flag) ec.vbr_flag = 0;
    if (p->alloc1) p->short_block_threshold = 0x7fffffff;       // its drivers never look at the transient detector
    p->vbr_flag = ec.vbr_flag;
    if (ec.vbr_flag) {                              // gen_vbr_table (mp3enc.cpp:964-1041)
        for (int i = 1; i < 15; i++) {
            int mb = fcoef * br_tab[i] / p->samprate;
            p->vbr_framebytes[i] = mb;
            p->vbr_main_framebytes[i] = mb - 4 - p->side_bytes;
        }
        p->vbr_framebytes[15] = p->vbr_main_framebytes[15] = 9999999;
        const int pool_cap = h_id ? 511 : 255;      // main_data_begin is 9 bits in MPEG-1, 8 in MPEG-2
        p->vbr_pool_target = (pool_cap + 1) >> 1;
        int i;
        for (i = 14; i >= 2; i--) {
            if (p->nchan * ec.vbr_br_limit >= br_tab[i]) break;
            p->vbr_pool_target = (p->vbr_pool_target + pool_cap) >> 1;
        }
        p->ivbr_max = i;
        p->ivbr_min = 1;
        p->AveTargetBits = (8 * p->vbr_main_framebytes[p->ivbr_max] / ((h_id ? 2 : 1) * p->nchan)) - p->sf_bit_max;
        p->initialMNR = MN(MX(10 * ec.vbr_mnr, 210), 1500);
    } else {
        const int tmp = p->totbitrate / p->nchan;
        p->initialMNR = MN(MX(h_id ? 125 * (tmp - 32) / 8 : 10 * ((30 * tmp) / 8 - 70), 0), 1000);
    }
    ec.vbr_delta_mnr = MX(MN(ec.vbr_delta_mnr, 50), -40);
    for (int i = 0; i < 21; i++) ec.mnr_adjust[i] = MX(MN(ec.mnr_adjust[i], 200), -200);
    p->hf_flag = ec.hf_flag;
    p->test1 = ec.test1 < 0 ? 6 : ec.test1;

    p->nsf3[0] = p->nsf2[0] = p->nsf[0] = sfbl_limit(p->tix, p->band_limit);
    p->nsf3[1] = p->nsf2[1] = p->nsf[1] = sfbl_limit(p->tix, p->band_limit_stereo);
    if (p->hf_flag) { p->nsf2[0] = 22; p->nBand_l[21] = 100; }
    if (p->hf_flag & 2) { p->nsf3[0] = 22; p->nsf3[1] = 22; }
    k = 0;
    for (int i = 0; i < 22; i++) { p->startBand_l[i] = k; k += p->nBand_l[i]; }
    p->startBand_l[22] = k;
    p->startBand_l[23] = 576;
    k = 0;
    for (int i = 0; i < 13; i++) { p->startBand_s[i] = k; k += p->nBand_s[i]; }
    p->startBand_s[13] = k;
    {   // CBitAlloShort::BitAlloInit (bitallos.cpp:128-200): limits arrive in long-block lines
        int bl = p->band_limit / 3 - 10, i;
        for (i = 0; i < 14; i++) if (bl <= sfb_short[p->tix][i]) break;
        p->nsfs = i > 12 ? 12 : i;
        p->nbmax_s = p->startBand_s[p->nsfs];
        for (i = 0; i < 12; i++) p->look_log_cbwmb_s[i] = (int) (100.0f * (float) (10.0 * log10((double) (float) p->nBand_s[i])));
        for (i = 0; i < 192; i++) {
            int b = 12;
            for (int j = 0; j < 13; j++) if (i < p->startBand_s[j + 1]) { b = j; break; }
            p->sband_of_line[i] = (unsigned char) b;
        }
    }
    for (int j = 0; j < 2; j++) p->nbmax3[j] = p->nbmax2[j] = p->nbmax[j] = p->startBand_l[p->nsf[j]];
    if (p->hf_flag) p->nbmax2[0] = p->startBand_l[p->nsf2[0]];
    if (p->hf_flag & 2) { p->nbmax3[0] = p->startBand_l[p->nsf3[0]]; p->nbmax3[1] = p->startBand_l[p->nsf3[1]]; }
    for (int i = 0; i < 576; i++) {
        int b = 21;
        for (int j = 0; j < 22; j++) if (i < p->startBand_l[j + 1]) { b = j; break; }
        p->band_of_line[i] = (unsigned char) b;
    }
    for (int i = 0; i < 128; i++) {
        p->look_gain[i] = (float) (pow(2.0, 0.25 * (i - 8)));
        p->look_34igain[i] = (float) (1.0 / pow((double) p->look_gain[i], (double) (3.0 / 4.0)));
    }
    for (int i = 0; i < 256; i++) p->look_ix43[i] = (float) (i * pow((double) i, (double) (1.0 / 3.0)));
    for (int i = 0; i < 21; i++) p->look_log_cbwmb[i] = (int) (100.0f * (float) (10.0 * log10((double) (float) p->nBand_l[i])));
    if (!ec.quick) {                                 // taper for the legacy FFT model (bitallo3.cpp:410-432)
        for (int i = 11; i < 22; i++) p->taperNT[i] = 100 + MN(150, 20 * (i - 11));
        if (p->vbr_flag) for (int i = 11; i < 22; i++) p->taperNT[i] = MN(p->taperNT[i], p->initialMNR);
        for (int i = 0; i < 21; i++) p->taperNT[i] -= 10 * mnrGOLD[i];
    }
    p->initialMNR += 10 * ec.vbr_delta_mnr - (h_id ? 0 : 300);         // bitallo3.cpp:446-453
    for (int i = 0; i < 22; i++) if (p->nBand_l[i] != 0) p->rnBand_l[i] = (1.0f / p->nBand_l[i]);

    if (p->alloc1) alloc1_tables(p);
    if (!band_runs(p)) return 0;
    p->ec = ec;
    p->ec.mode = p->h_mode;
    p->ec.bitrate = p->totbitrate / p->nchan;
    p->ec.samprate = p->samprate;
    p->ec.nsbstereo = is_flag ? nsbstereo : 32;
    p->ec.freq_limit = ec.hf_flag ? ec.freq_limit : p->nsb_limit * (p->samprate / 64);
    p->ec.nsb_limit = p->nsb_limit;
    p->ec.layer = 3;
    return p->nchan * 4 * 1152;
}

// Initial per-stream state (mp3enc.cpp:278-287,614-621,788-837; bitallo3.cpp:300-316)
void hx_stream_reset(const HxParams *p, int cls, HxStream *s)
{
    memset(s, 0, sizeof(*s));
    s->cls = cls;
    for (int i = 0; i < 32; i++) s->attack_hist[0][i] = s->attack_hist[1][i] = 9000;
    for (int c = 0; c < 2; c++) for (int i = 0; i < 64; i++) s->thr_prev[c][i] = 1.0e20f;
    s->MNR = p->initialMNR;
    s->PoolFraction = p->vbr_flag ? 614 : 0;
    s->padcount = p->divisor;
    if (p->alloc1) {        // bitallo1.cpp:163-196
        for (int c = 0; c < 2; c++) for (int j = 0; j < p->nsf[c]; j++) s->a1_gsf[c][j] = 35;
        s->a1_bitadjust[0] = s->a1_bitadjust[1] = -100;
        s->a1_running_a = (1.0f / 20.0f);
        s->a1_ave_alpha = 40.0f;
    }
}

// The low-footprint stream walk (hx_alloc.hip, HX_SLIM) keeps 4 + 16 constants in place of the two 128-entry gain tables and
// the 256-entry exponent table of x^(3/4): 2^((i - 8) / 4) has the mantissa of entry 8 + ((i - 8) & 3), 2^(-3 (i - 8) / 16) that
// of entry 8 + ((i - 8) & 15), and scaling a float by a power of two is exact.  The tables come out of the host's pow(), so the
// identity is checked on the tables a batch would use, bit for bit, before that kernel may be chosen.
int hx_slim_tables_ok(const HxParams *p, const HxGlobalTabs *g)
{
    for (int i = 0; i < 128; i++) {
        const int k = i - 8;
        const float gn = ldexpf(p->look_gain[8 + (k & 3)], k >> 2), ig = ldexpf(p->look_34igain[8 + (k & 15)], -3 * (k >> 4));
        if (memcmp(&gn, &p->look_gain[i], 4) != 0 || memcmp(&ig, &p->look_34igain[i], 4) != 0) return 0;
    }
    for (int e = 1; e < 255; e++) {
        const int k = 3 * (e - 127);
        const float ex = ldexpf(p->look_gain[8 + (k & 3)], k >> 2);
        if (memcmp(&ex, &g->pow34_exp[e], 4) != 0) return 0;
    }
    if (g->pow34_exp[0] != 0.0f || !(g->pow34_exp[255] > 3.0e38f)) return 0;
    for (int i = 0; i < 256; i++) if (g->mblog[i] + 38227 < 0 || g->mblog[i] + 38227 > 65535) return 0;
    for (int i = 0; i < 84; i++) if (g->logsub[i] < -32768 || g->logsub[i] > 32767) return 0;
    return 1;
}

// ---- which libm the host has, against the one the first-generation allocator's kernels restate (hx_libm32.h) ----
// The reference's bitallo1.cpp calls libm's logf / log10f, which are not correctly rounded: its output depends on the libm it
// is linked with, and the kernels agree with glibc 2.35's.  The oracle on a box calls that box's libm; where the two differ a
// parity test of those streams would compare two different references.  hx_host_libc_version: gnu_get_libc_version();
// hx_libm32_spot_check: how many of n arguments spread over the positive normal floats give another logf or log10f here.
#include <gnu/libc-version.h>
#define HX_HD static inline
#include "hx_libm32.h"
const char *hx_host_libc_version(void) { return gnu_get_libc_version(); }
int hx_libm32_spot_check(int n)
{
    int bad = 0;
    if (n < 1) n = 1;
    const unsigned long long span = 0x7f800000ull - 0x00800000ull;
    for (int i = 0; i < n; i++) {
        // a stride co-prime with the span walks exponents and mantissas alike
        const uint32_t u = (uint32_t) (0x00800000ull + ((unsigned long long) i * 2654435761ull + 12345ull) % span);
        const float x = hx_u2f(u);
        volatile float a = logf(x), c = log10f(x);
        if (hx_f2u(a) != hx_f2u(hx_logf(x)) || hx_f2u(c) != hx_f2u(hx_log10f(x))) bad++;
    }
    return bad;
}
