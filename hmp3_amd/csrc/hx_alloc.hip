// hx_alloc.hip - K6: per-stream rate loop of the batched MP3 encoder for MI355X (gfx950).
// One wavefront (64 lanes) owns one stream and walks its frames in order, because the
// allocator state (long-term MNR, per-band gain estimators, bit reservoir, scfsi memory)
// is carried frame to frame (reference bitallo3.cpp:484-3149, mp3enc.cpp:1492-1597,
// :2106-2333, l3pack.c:107-1187, bitalloc.cpp:470-811).
//
// Inside a granule the work is spread over the lanes three ways:
//   line-parallel   quantise / x^(3/4) / M-S butterflies / noise terms / Huffman lengths
//                   (lane j handles lines j, j+64, ...);
//   band-parallel   every scalefactor band owns lane 32*ch + sfb for its integer control
//                   (noise targets, gain search state machine, scalefactors);
//   wave collective integer reductions and scans (bit counts, code offsets, maxima).
// Floating-point sums that the reference accumulates line by line are formed from
// line-parallel terms that the owning band lane then adds in the reference's order, so every
// decision (they are integer compares on millibel values) is bit-identical to the oracle.
// The gain search of all 42 bands advances in the same sweep.
#include "hx_dev.h"

#define GMIN_OFFSET 70
#define PART23 4021
#define NB 22

struct AllocLds {
    float xr[2][576];
    float x34[2][576];
    float term[2][576];
    int ix[2][576];
    unsigned char signx[2][576];
    unsigned char band_of_line[576];
    // tables staged from global memory
    float look_ix43[256], look_gain[128], look_34igain[128];
    int mblog[256];
    float pow34_exp[256], pow34_a[16], pow34_b[16], quant_off[32];
    int logsub[84];
    unsigned char huff_len[1408];       // codes stay in global memory (read once per pair when packing)
    unsigned char sband_of_line[192];
    int nBand_s[16], startBand_s[16], logcbw_s[16];
    unsigned short huff_off[32];
    unsigned char huff_dim[32], huff_lin[32], quada_code[16], quada_len[16];
    int nBand[NB], startBand[24], logcbw[NB], taper[NB];
    int NTadjust[2][NB];                // long-block gain estimator feedback (persists)
    union {
        struct {    // long blocks: per band working set, [channel][sfb]
            int snr[2][NB], Noise0[2][NB], Noise[2][NB], NT[2][NB];
            int gzero[2][NB], gmin[2][NB], gsf[2][NB], sf[2][NB], active[2][NB];
            int ixmax[2][NB], ix10xmax[2][NB], up[2][NB], lo[2][NB], geval[2][NB], maskmb[2][NB];
            float xsxx[2][NB], xsxxms[2][NB], x34max[2][NB];
        };
        struct {    // short blocks: [channel][window][sfb]
            int s_snr[2][3][16], s_Noise0[2][3][16], s_Noise[2][3][16], s_NT[2][3][16];
            int s_gzero[2][3][16], s_gmin[2][3][16], s_gsf[2][3][16], s_sf[2][3][16], s_active[2][3][16];
            int s_ixmax[2][3][16], s_geval[3][16], s_tmpn[3][16], s_maskmb[2][3][16];
            float s_xsxx[2][3][16], s_x34max[2][3][16];
            int s_G[2][3], s_GG[2], s_subgain[2][3];
        };
    };
    // per channel
    int G[2], preemp[2], scale[2], huff_bits[2];
    int hs_table[2][4], hs_cbreg[2][3], hs_nbig[2], hs_nquads[2], hs_bits[2];
    // stream scalars (persist across frames)
    int MNR, PoolFraction, call_count, ms_memory;
    int hf_quant, hf_quant_stereo[2], gsf_hf, gsf_hf_stereo[2];
    int sf_save[2][21];
    int scfsi[2];
    // call scalars
    int nchan, block_type, maxBits, maxTargetBits, minTargetBits, PoolBits, TargetBits, deltaMNR, activeBands;
    int tmp[8];
    int tmpn[2][NB];
    // bit staging for one frame's main data (MSB-first 32-bit words)
    unsigned int bitw[640];
    HxGr gr[2][2];
    int sfout[2][2][NB];
    int sfs[2][3][12];                  // short-block scalefactors of the current granule
};

#define LANE ((int) threadIdx.x)
#define SYNC() __syncthreads()

// ---------------------------------------------------------------------------------------
// bit staging: OR an n-bit field (n <= 32) at absolute bit position pos
__device__ __forceinline__ void put_bits(AllocLds &L, int pos, unsigned val, int n)
{
    if (n <= 0) return;
    int w = pos >> 5, o = pos & 31;
    unsigned long long v = ((unsigned long long) val) << (64 - n - o);  // field left-aligned in 64 bits at offset o
    unsigned hi = (unsigned) (v >> 32), lo = (unsigned) v;
    if (hi) atomicOr(&L.bitw[w], hi);
    if (lo) atomicOr(&L.bitw[w + 1], lo);
}
__device__ __forceinline__ void put_bits64(AllocLds &L, int pos, unsigned long long val, int n)
{
    if (n > 32) { put_bits(L, pos, (unsigned) (val >> 32), n - 32); put_bits(L, pos + n - 32, (unsigned) val, 32); }
    else put_bits(L, pos, (unsigned) val, n);
}

// sequential (reference-order) sum of term[ch][start .. start+n)
__device__ __forceinline__ float band_sum(const float *t, int n, float acc)
{
    for (int j = 0; j < n; j++) acc += t[j];
    return acc;
}

// x^(3/4): piecewise-linear mantissa fit x exponent table (reference pow34.c:132-154)
__device__ __forceinline__ float pow34(const AllocLds &L, float x)
{
    unsigned u = hx_f2bits(x);
    float m = hx_bits2f((u & 0x7FFFFFu) | (127u << 23));
    unsigned seg = (u >> 19) & 15, e = (u >> 23) & 255;
    return (m * L.pow34_b[seg] + L.pow34_a[seg]) * L.pow34_exp[e];
}

__device__ __forceinline__ int logsubber(const AllocLds &L, int n1, int n2)
{
    int k = (n1 - n2) >> 4;
    if (k > 83) k = 83;
    return n1 + L.logsub[k];
}

__device__ __forceinline__ int drop_guard(int noise0, int nt)
{
    int tsnr = noise0 - nt;
    if (tsnr < 300) { tsnr = 187 + ((3 * tsnr) >> 3) - tsnr; nt -= tsnr; }
    return nt;
}

// scalefactor limits by [scalefac_scale][preflag] (reference bitallo3.cpp:87-162)
__device__ __forceinline__ int pretab(int i)
{
    return (i < 11 || i > 20) ? 0 : (i < 15 ? 1 : (i < 17 ? 2 : (i < 20 ? 3 : 2)));
}
__device__ __forceinline__ int sf_limit_hi(int scale, int pre, int i)
{
    int base = (i < 11) ? (scale ? 62 : 31) : (scale ? 30 : 15);
    if (pre) base += (scale ? 4 : 2) * pretab(i);
    return base;
}
__device__ __forceinline__ int sf_limit_lo(int scale, int pre, int i) { return pre ? (scale ? 4 : 2) * pretab(i) : 0; }
__device__ __forceinline__ int sf_upper(int scale, int pre, int i)
{
    int base = (i < 11) ? (scale ? 60 : 30) : (scale ? 28 : 14);
    if (pre) base += (scale ? 4 : 2) * pretab(i);
    return base;
}

// ---------------------------------------------------------------------------------------
// Noise sweep: every band with geval >= 0 gets its quantisation noise (mB) measured at gain
// step geval (reference l3math.c:512-541).  Result left in L.tmp-free per-band array `out`.
__device__ void noise_sweep(AllocLds &L, const HxParams *p, int nlines0, int nlines1, int out[2][NB])
{
    for (int ch = 0; ch < 2; ch++) {
        int nl = ch ? nlines1 : nlines0;
        for (int j = LANE; j < nl; j += 64) {
            int g = L.geval[ch][L.band_of_line[j]];
            if (g >= 0) {
                float igain = L.look_34igain[g], gain = L.look_gain[g];
                float tmp = (igain * L.x34[ch][j] + (0.0f - 0.0946f));
                int qx = (int) (tmp + copysignf(0.5f, tmp));
                float xhat;
                if (qx >= 0 && qx < 256) xhat = gain * L.look_ix43[qx];
                else xhat = (float) (gain * pow((double) qx, (4.0 / 3.0)));
                tmp = L.xr[ch][j] - xhat;
                L.term[ch][j] = tmp * tmp;
            }
        }
    }
    SYNC();
    {
        int ch = LANE >> 5, i = LANE & 31;
        if (i < NB && L.geval[ch][i] >= 0) {
            float sxx = band_sum(&L.term[ch][L.startBand[i]], L.nBand[i], 0.0f);
            out[ch][i] = hx_mblog(L.mblog, 1.0e-12f + sxx) - L.logcbw[i];
        }
    }
    SYNC();
}

// ---------------------------------------------------------------------------------------
// reference bitallo3.cpp:1069-1126
__device__ void adjust_nt(AllocLds &L, const HxParams *p)
{
    const int f = p->test1;
    if (f == 0) return;
    const int ch = LANE >> 5, i = LANE & 31;
    const int nsf = p->nsf[ch];
    const int sth = (i < 14) ? 0 : (i < 17 ? 100 : (i == 17 ? 200 : 300));
    const bool sel = (i < nsf) && (L.snr[ch][i] > sth);
    // integer sums within each half-wave (channel)
    int na = sel ? 1 : 0, ab = sel ? L.nBand[i] * L.NT[ch][i] : 0, nab = sel ? L.nBand[i] : 0;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        na += __shfl_xor(na, m, 64);
        ab += __shfl_xor(ab, m, 64);
        nab += __shfl_xor(nab, m, 64);
    }
    na += 1; nab += 1;
    ab = ab / nab;
    if (na >= 5 && sel) {
        int dmax = max(L.snr[ch][i] - 400, 0);
        int d = (f * (ab - L.NT[ch][i])) >> 4;
        d = min(d, dmax);
        L.NT[ch][i] = L.NT[ch][i] + d;
    }
    SYNC();
}

// x^(3/4) of the first nl lines, band maxima, gzero / gmin (reference bitallo3.cpp:878-896)
__device__ void pow34_gzero(AllocLds &L, const HxParams *p, int nl0, int nl1, int nb0, int nb1)
{
    for (int ch = 0; ch < 2; ch++) {
        int nl = ch ? nl1 : nl0;
        for (int j = LANE; j < nl; j += 64) L.x34[ch][j] = pow34(L, L.xr[ch][j]);
    }
    SYNC();
    const int ch = LANE >> 5, i = LANE & 31;
    if (i < (ch ? nb1 : nb0)) {
        const float *y = &L.x34[ch][L.startBand[i]];
        float m = 0.0f;
        for (int j = 0; j < L.nBand[i]; j++) if (y[j] > m) m = y[j];
        L.x34max[ch][i] = m;
        int gz = max(0, hx_round((0.017716950f * hx_mblog(L.mblog, m) + (104.585000f - 100.0f + 8.0f))));
        L.gzero[ch][i] = gz;
        L.gmin[ch][i] = max(0, gz - GMIN_OFFSET);
    }
    SYNC();
}

// reference bitallo3.cpp:816-898
__device__ void startup_lr(AllocLds &L, const HxParams *p)
{
    const int mnr = L.MNR + 100;
    for (int ch = 0; ch < 2; ch++)
        for (int j = LANE; j < p->nbmax3[ch]; j += 64) {
            float x = L.xr[ch][j];
            unsigned char sg = 0;
            if (!(x >= 0.0f)) { sg = 1; x = -x; }
            L.signx[ch][j] = sg;
            L.xr[ch][j] = x;
            L.term[ch][j] = x * x;
        }
    SYNC();
    const int ch = LANE >> 5, i = LANE & 31;
    int act = 0;
    if (i < p->nsf3[ch]) L.xsxx[ch][i] = band_sum(&L.term[ch][L.startBand[i]], L.nBand[i], 0.0f);
    if (i < p->nsf[ch]) {
        int n0 = hx_mblog(L.mblog, L.xsxx[ch][i]) - L.logcbw[i], nt;
        if (n0 < -2000) nt = n0 + 1000;
        else {
            act = L.nBand[i];
            nt = drop_guard(n0, L.maskmb[ch][i] - L.logcbw[i] - mnr + L.taper[i]);
        }
        L.Noise0[ch][i] = n0;
        L.NT[ch][i] = nt;
        L.snr[ch][i] = n0 - nt;
    }
    act = hx_wave_sum(act);
    if (LANE == 0) L.activeBands = act;
    SYNC();
    adjust_nt(L, p);
    pow34_gzero(L, p, p->nbmax3[0], p->nbmax3[1], p->nsf3[0], p->nsf3[1]);
}

// reference bitallo3.cpp:902-1066
__device__ void startup_ms(AllocLds &L, const HxParams *p)
{
    if (LANE == 0 && p->vbr_flag == 0 && L.call_count > 10 && (L.TargetBits - L.minTargetBits) < 100)
        L.MNR = min(L.MNR + 50, 2050);
    SYNC();
    const int mnr = L.MNR;
    const int nl = p->hf_flag ? L.startBand[22] : p->nbmax[0];     // lines that get the M/S butterfly
    for (int j = LANE; j < nl; j += 64) {
        float l = L.xr[0][j], r = L.xr[1][j];
        L.term[0][j] = l * l;
        L.term[1][j] = r * r;
    }
    SYNC();
    const int ch = LANE >> 5, i = LANE & 31;
    const bool band = i < p->nsf[0];
    if (band) L.xsxx[ch][i] = band_sum(&L.term[ch][L.startBand[i]], L.nBand[i], 0.0f);
    SYNC();
    for (int j = LANE; j < nl; j += 64) {       // reference l3math.c:905-930, no 1/sqrt(2)
        float l = L.xr[0][j], r = L.xr[1][j];
        float x0 = (l + r), x1 = (l - r);
        unsigned char s0 = 0, s1 = 0;
        if (x0 < 0.0f) { s0 = 1; x0 = -x0; }
        if (x1 < 0.0f) { s1 = 1; x1 = -x1; }
        L.signx[0][j] = s0; L.signx[1][j] = s1;
        L.xr[0][j] = x0; L.xr[1][j] = x1;
        L.term[0][j] = x0 * x0;
        L.term[1][j] = x1 * x1;
    }
    SYNC();
    if (band) L.xsxxms[ch][i] = band_sum(&L.term[ch][L.startBand[i]], L.nBand[i], 0.0f);
    SYNC();
    int act = 0;
    if (band) {     // lane (ch, i): left (ch 0) / right (ch 1) noise target
        int cbw = L.logcbw[i];
        int n0 = hx_mblog(L.mblog, L.xsxx[ch][i]) - cbw, nt;
        if (n0 < -2000) nt = 10000;
        else { nt = drop_guard(n0, (L.maskmb[ch][i] - cbw) - mnr + L.taper[i]); act = L.nBand[i]; }
        L.NT[ch][i] = nt;
        L.snr[ch][i] = n0 - nt;
        L.Noise0[ch][i] = hx_mblog(L.mblog, L.xsxxms[ch][i]) - cbw;
    }
    act = hx_wave_sum(act);
    if (LANE == 0) L.activeBands = act;
    SYNC();
    adjust_nt(L, p);
    if (LANE < p->nsf[0]) {
        int b = LANE;
        int NTL = L.NT[0][b], NTR = L.NT[1][b], Nsum = L.Noise0[0][b], Ndiff = L.Noise0[1][b];
        int xNT = min(NTL, NTR) + 300, nt0, nt1;
        nt0 = nt1 = xNT;
        if (Ndiff < xNT) { nt0 = logsubber(L, xNT, Ndiff); if (b < 16) nt0 -= 200; }
        if (Nsum < xNT) nt1 = logsubber(L, xNT, Nsum);
        L.NT[0][b] = nt0; L.NT[1][b] = nt1;
        L.snr[0][b] = Nsum - nt0;
        L.snr[1][b] = Ndiff - nt1;
    }
    SYNC();
    pow34_gzero(L, p, p->nbmax2[0], p->nbmax2[1], p->nsf2[0], p->nsf2[1]);
}

// reference bitallo3.cpp:1130-1160
__device__ void seek_initial(AllocLds &L, const HxParams *p)
{
    const int ch = LANE >> 5, i = LANE & 31;
    if (i < p->nsf[ch]) {
        int na = L.NTadjust[ch][i];
        na = max(na, -400);
        na = min(na, 400);
        L.NTadjust[ch][i] = na;
        float g4 = 0.017716950f * hx_mblog(L.mblog, L.x34max[ch][i]) + (88.411238f - 100.0f + 8.0f);
        float d = (1.00f / 110.5f) * (1800 - 8 * i - (L.Noise0[ch][i] - L.NT[ch][i] + na));
        float g = g4 + d;
        int gs = hx_round(g);
        gs = min(gs, L.gzero[ch][i]);
        gs = max(gs, L.gmin[ch][i]);
        L.gsf[ch][i] = gs;
    }
    SYNC();
}

// reference bitallo3.cpp:1164-1296: all bands walk their gain step concurrently
__device__ void seek_actual(AllocLds &L, const HxParams *p)
{
    const int ch = LANE >> 5, i = LANE & 31;
    const bool band = i < p->nsf[ch];
    // per-lane state machine: mode 0 = idle/done, 1 = first measurement, 2 = walking down, 3 = walking up
    int mode = 0, s = 0, t = 0, NTarget = 0, absmin = 0, tnmin = 0, smin = 0, iter = 0, niter = 0;
    if (band) {
        NTarget = L.NT[ch][i];
        s = L.gsf[ch][i];
        if (L.Noise0[ch][i] > NTarget) mode = 1;
        else { L.gsf[ch][i] = L.gzero[ch][i] + 5; L.Noise[ch][i] = L.Noise0[ch][i]; }
    }
    if (i < NB) L.geval[ch][i] = (mode == 1) ? s : -1;
    SYNC();
    while (__any(mode != 0)) {
        noise_sweep(L, p, p->nbmax[0], p->nbmax[1], L.tmpn);
        if (mode == 1) {
            int noise = L.tmpn[ch][i], dn = noise - NTarget;
            L.NTadjust[ch][i] = L.NTadjust[ch][i] + (dn >> 3);
            absmin = abs(dn); tnmin = noise; smin = s; iter = 0;
            if (dn > 100) { t = s - 1; niter = min(t, 20); mode = (niter > 0) ? 2 : 0; }
            else if (dn < -100) { t = s + 1; niter = 20; mode = 3; }
            else mode = 0;
            if (mode == 0) { L.gsf[ch][i] = smin; L.Noise[ch][i] = tnmin; }
        } else if (mode == 2 || mode == 3) {
            int tn = L.tmpn[ch][i], ad = abs(tn - NTarget);
            if (ad < absmin) { absmin = ad; tnmin = tn; smin = t; }
            iter++;
            bool stop = (mode == 2) ? (tn <= NTarget) : (tn >= NTarget);
            if (stop || iter >= niter) mode = 0;
            else t += (mode == 2) ? -1 : 1;
            if (mode == 0) { L.gsf[ch][i] = smin; L.Noise[ch][i] = tnmin; }
        }
        if (i < NB) L.geval[ch][i] = (mode != 0) ? t : -1;
        SYNC();
    }
}

// ---------------------------------------------------------------------------------------
// Global gain / scalefactors (reference bitallo3.cpp:1793-1862, 1892-2019 L/R, 2022-2170 M/S).
// Band-parallel: channel ch lives in lanes 32*ch .. 32*ch+21; maxima / ORs are half-wave
// reductions.  The M/S variant carries the running maximum from a silent channel 0 into
// channel 1 (the reference only resets it on the non-silent path).
__device__ __forceinline__ int half_max(int v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) { int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int half_or(int v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v |= __shfl_xor(v, m, 64);
    return v;
}

__device__ int scale_factors(AllocLds &L, const HxParams *p, int ms)
{
    const int ch = LANE >> 5, i = LANE & 31;
    const bool band = i < p->nsf[ch];
    int gsf = 0, gz = 0, act = 0;
    if (band) {
        gsf = max(L.gsf[ch][i], L.gmin[ch][i]);
        gz = L.gzero[ch][i];
        act = (gsf < gz) ? -1 : 0;
    }
    int gact = half_max((band && act) ? gsf : -1);          // max over active bands, -1 if none
    int gzmax = half_max(band ? gz : -1);
    int g0init, g1init;
    if (ms) {
        g0init = L.hf_quant ? L.gsf_hf : -1;
        int G0 = max(g0init, __shfl(gact, 0, 64));
        // channel 0 silent -> its Gtmp (max of gzero, and of the initial value) leaks into channel 1
        int leak = max(G0, __shfl(gzmax, 0, 64));
        g1init = (G0 < 0) ? leak : -1;
    } else {
        g0init = L.gsf_hf_stereo[0];
        g1init = L.gsf_hf_stereo[1];
    }
    int Gtmp = max(ch ? g1init : g0init, gact);
    const bool silent = Gtmp < 0;
    int sf = 0, pre = 0, scale = 0, dsf = 2;
    if (silent) {
        Gtmp = max(Gtmp, gzmax);
        if (band) { sf = 0; gsf = gz; }
    } else {
        if (band) sf = (Gtmp - gsf) & act;
        int sp0 = 0, sp1 = 0, sp2 = 0, sp3 = 0;
        if (band && act) {
            sp0 = (sf_limit_hi(0, 0, i) - sf);
            sp1 = (sf_limit_hi(0, 1, i) - sf) | (sf - sf_limit_lo(0, 1, i));
            sp2 = (sf_limit_hi(1, 0, i) - sf);
            sp3 = (sf_limit_hi(1, 1, i) - sf) | (sf - sf_limit_lo(1, 1, i));
        }
        sp0 = half_or(sp0); sp1 = half_or(sp1); sp2 = half_or(sp2); sp3 = half_or(sp3);
        if (sp0 >= 0) { scale = 0; pre = 0; }
        else if (sp1 >= 0) { scale = 0; pre = 1; }
        else if (sp2 >= 0) { scale = 1; pre = 0; }
        else if (sp3 >= 0) { scale = 1; pre = 1; }
        else { scale = 1; pre = 0; }
        if (band) {
            int noise = L.Noise[ch][i], nt = L.NT[ch][i];
            if (scale == 0) {
                dsf = 2;
                if (ms) {
                    if (act) {
                        if ((gz - gsf) < 5) sf++;
                        else if ((i < 11) && (noise > nt)) sf++;
                        sf &= (~1);
                    }
                } else {
                    if ((i < 11) && (noise > nt)) sf++;
                    sf &= (~1);
                }
            } else {
                dsf = 4;
                if (!ms || act) {
                    int s = sf & (~3), d = sf - s;
                    int dN = noise - nt + 150 * d;
                    int thr = (i < 15) ? 250 : (i == 15 ? 300 : (i < 18 ? 400 : (i < 20 ? 500 : 600)));
                    if (dN > thr) s = s + 4;
                    else if (ms && (gz - gsf - d) < 5) s = s + 4;
                    sf = ms ? s : (s & act);
                }
            }
        }
    }
    if (i < NB) {
        int up = sf_upper(silent ? 0 : scale, silent ? 0 : pre, i), lo = sf_limit_lo(silent ? 0 : scale, silent ? 0 : pre, i);
        L.up[ch][i] = up;
        L.lo[ch][i] = lo;
        if (band && !silent) {
            if (sf > up) sf = up; else if (sf < lo) sf = lo;
            if (act) {
                gsf = Gtmp - sf;
                if (gsf < 0) { gsf += dsf; sf -= dsf; }
                if (gsf >= gz) { gsf = gz + 5; sf = lo; }
            }
        }
        if (band) { L.gsf[ch][i] = gsf; L.sf[ch][i] = sf; L.active[ch][i] = silent ? 0 : act; }
    }
    if (i == 0) { L.G[ch] = Gtmp; L.preemp[ch] = pre; L.scale[ch] = scale; }
    SYNC();
    return 0;
}

// reference bitallo3.cpp:1348-1396
__device__ void big_lucky_noise(AllocLds &L, const HxParams *p)
{
    const int ch = LANE >> 5, i = LANE & 31;
    const int m = min(13, p->nsf[ch]);
    int mode = 0, s = 0, s0 = 0, g0 = 0, GG = 0, sdelta = 2, smin = 0, nt = 0;
    if (i < m && L.active[ch][i] && (L.gsf[ch][i] < (L.gzero[ch][i] - 5))) {
        sdelta = 2 * (1 + L.scale[ch]);
        GG = L.G[ch];
        smin = L.sf[ch][i];
        g0 = L.gzero[ch][i] - 4;
        s = min(L.sf[ch][i] - sdelta, L.up[ch][i]);
        s0 = L.lo[ch][i];
        nt = L.NT[ch][i];
        mode = 1;
        if (!(s >= s0) || (GG - s) >= g0) mode = 2;        // loop body never runs
    }
    if (i < NB) L.geval[ch][i] = (mode == 1) ? GG - s : -1;
    SYNC();
    while (__any(mode == 1)) {
        noise_sweep(L, p, L.startBand[13], L.startBand[13], L.tmpn);
        if (mode == 1) {
            int noise = L.tmpn[ch][i];
            if (noise <= nt) { L.Noise[ch][i] = noise; smin = s; }
            s -= sdelta;
            if (!(s >= s0) || (GG - s) >= g0) mode = 2;
        }
        if (i < NB) L.geval[ch][i] = (mode == 1) ? GG - s : -1;
        SYNC();
    }
    if (mode == 2) {
        L.sf[ch][i] = smin;
        L.gsf[ch][i] = max(GG - smin, 0);
    }
    SYNC();
}

// reference bitallo3.cpp:1540-1585 with l3math.c:656-694: quantise every coded band
__device__ void do_quant(AllocLds &L, const HxParams *p, int opt)
{
    const int ch = LANE >> 5, i = LANE & 31;
    if (i < p->nsf[ch]) L.ixmax[ch][i] = 0;
    SYNC();
    for (int c = 0; c < 2; c++)
        for (int j = LANE; j < p->nbmax[c]; j += 64) {
            int b = L.band_of_line[j];
            float igain = L.look_34igain[L.gsf[c][b]];
            int q;
            if (opt) {
                float t = igain * L.x34[c][j] + (0.5f - 0.4375f);
                int iq = (int) t;
                if (iq > 31) iq = 31;
                q = (int) (t - L.quant_off[iq]);
            } else {
                q = (int) (igain * L.x34[c][j] + (0.5f - 0.0946f));
            }
            L.ix[c][j] = q;
            if (q > 0) atomicMax(&L.ixmax[c][b], q);
        }
    SYNC();
}

// ---------------------------------------------------------------------------------------
// Huffman region split, table choice and bit count for one channel
// (reference bitalloc.cpp:310-420, 470-756; cnt.c:96-325).
struct Cand { int n; int t[4]; int tmax; };

__device__ __forceinline__ Cand candidates(int rmax)
{
    Cand c;
    c.t[0] = c.t[1] = c.t[2] = c.t[3] = 0;
    if (rmax <= 0) { c.n = 0; c.tmax = 0; }
    else if (rmax == 1) { c.n = 2; c.t[0] = 1; c.t[1] = 3; c.tmax = 1; }
    else if (rmax == 2) { c.n = 2; c.t[0] = 2; c.t[1] = 3; c.tmax = 2; }
    else if (rmax == 3) { c.n = 2; c.t[0] = 5; c.t[1] = 6; c.tmax = 3; }
    else if (rmax <= 5) { c.n = 4; c.t[0] = 7; c.t[1] = 8; c.t[2] = 9; c.t[3] = 12; c.tmax = 5; }
    else if (rmax <= 7) { c.n = 4; c.t[0] = 10; c.t[1] = 11; c.t[2] = 12; c.t[3] = 15; c.tmax = 7; }
    else if (rmax <= 15) { c.n = 2; c.t[0] = 13; c.t[1] = 15; c.tmax = 15; }
    else if (rmax == 16) { c.n = 2; c.t[0] = 16; c.t[1] = 24; c.tmax = 16; }
    else if (rmax <= 18) { c.n = 2; c.t[0] = 17; c.t[1] = 24; c.tmax = 18; }
    else if (rmax <= 22) { c.n = 2; c.t[0] = 18; c.t[1] = 24; c.tmax = 22; }
    else if (rmax <= 30) { c.n = 2; c.t[0] = 19; c.t[1] = 24; c.tmax = 30; }
    else if (rmax <= 46) { c.n = 2; c.t[0] = 25; c.t[1] = 20; c.tmax = 46; }
    else if (rmax <= 78) { c.n = 2; c.t[0] = 20; c.t[1] = 26; c.tmax = 78; }
    else if (rmax <= 142) { c.n = 2; c.t[0] = 27; c.t[1] = 21; c.tmax = 142; }
    else if (rmax <= 270) { c.n = 2; c.t[0] = 21; c.t[1] = 28; c.tmax = 270; }
    else if (rmax <= 526) { c.n = 2; c.t[0] = 29; c.t[1] = 22; c.tmax = 526; }
    else if (rmax <= 1038) { c.n = 2; c.t[0] = 22; c.t[1] = 30; c.tmax = 1038; }
    else if (rmax <= 2062) { c.n = 2; c.t[0] = 30; c.t[1] = 23; c.tmax = 2062; }
    else { c.n = 2; c.t[0] = 31; c.t[1] = 23; c.tmax = 8206; }
    return c;
}

// coded length of one pair in table t: Huffman length + sign bits + linbits
__device__ __forceinline__ int pair_len(const AllocLds &L, int t, int x, int y)
{
    int n;
    if (t >= 16) {
        int cx = x > 15 ? 15 : x, cy = y > 15 ? 15 : y, lin = L.huff_lin[t];
        n = L.huff_len[L.huff_off[t] + cx * 16 + cy];
        if (x >= 15) n += lin;
        if (y >= 15) n += lin;
    } else {
        n = L.huff_len[L.huff_off[t] + x * L.huff_dim[t] + y];
    }
    return n + (x != 0) + (y != 0);
}

__device__ __forceinline__ int region_max(const int *ixmax, int a, int b)
{
    int m = 0;
    for (int i = a; i < b; i++) if (m < ixmax[i]) m = ixmax[i];
    return m;
}

__device__ int count_bits_ch(AllocLds &L, const HxParams *p, int ch, int ncb)
{
    const int *ixmax = L.ixmax[ch];
    const int *ix = L.ix[ch];
    const int bt = L.block_type;
    int cb0, cb1, cb2, cb3, i;
    // region boundaries: cheap, computed redundantly by every lane (uniform)
    for (i = ncb - 1; i >= 0; i--) if (ixmax[i] > 0) break;
    cb3 = i + 1;
    for (; i >= 0; i--) if (ixmax[i] > 1) break;
    cb2 = i + 1;
    cb0 = cb1 = 0;
    if (bt == 0) { if (cb2 < 2) { cb2 = 2; if (cb3 < cb2) cb3 = cb2; } }
    else { cb0 = 8; cb2 = max(cb2, 8); cb3 = max(cb3, cb2); cb1 = cb0; }
    // topmost line > 1 in the last "big" band, topmost line > 0 in the last count1 band
    int lo2 = L.startBand[cb2 - 1], hi2 = L.startBand[cb2], lo3 = L.startBand[cb3 - 1], hi3 = L.startBand[cb3];
    int j2 = lo2, j3 = lo3;
    for (int j = lo2 + LANE; j < hi2; j += 64) if (ix[j] > 1) j2 = j;
    for (int j = lo3 + LANE; j < hi3; j += 64) if (ix[j] > 0) j3 = j;
    j2 = hx_wave_max(j2);
    j3 = hx_wave_max(j3);
    int nbig = (j2 + 2) & (~1);
    if (bt == 0) { if (nbig < L.startBand[2]) nbig = L.startBand[2]; }
    else { if (nbig < L.startBand[8]) nbig = L.startBand[8]; }
    int nquads = (j3 + 4 - nbig) >> 2;
    if (bt != 0) nquads = max(nquads, 0);
    Cand c0, c1, c2;
    if (bt == 0) {
        const int c = cb2;      // region_table (reference bitalloc.cpp:124-197)
        int r0 = (c < 6) ? 1 : (c < 9 ? 2 : (c < 12 ? 3 : (c < 15 ? 4 : (c < 18 ? 5 : (c < 21 ? 6 : 7)))));
        int r1 = (c < 5) ? 1 : (c < 8 ? 2 : (c < 10 ? 3 : (c == 10 ? 4 : (c < 15 ? 5 : (c < 17 ? 6 : (c < 20 ? 7 : 8))))));
        cb0 = r0;
        cb1 = r0 + r1;
        if (cb0 < 1) cb0 = 1;
        if (cb1 <= cb0) cb1 = cb0 + 1;
        if (cb1 > cb0 + 8) cb1 = cb0 + 8;
        c0 = candidates(region_max(ixmax, 0, cb0));
        c1 = candidates(region_max(ixmax, cb0, cb1));
        c2 = candidates(region_max(ixmax, cb1, cb2));
        if (c2.tmax < c1.tmax) {
            int j;
            for (j = cb1 - 1; j > cb0; j--) if (ixmax[j] > c2.tmax) break;
            cb1 = j + 1;
        }
        if (c1.tmax < c0.tmax) {
            int n = cb1 - 8, j;
            if (n < 1) n = 1;
            for (j = cb0 - 1; j > n; j--) if (ixmax[j] > c1.tmax) break;
            cb0 = j + 1;
        }
    } else {
        c0 = candidates(region_max(ixmax, 0, cb0));
        c1 = candidates(0);
        c2 = candidates(region_max(ixmax, cb0, cb2));
    }
    const int n0 = L.startBand[cb0], n1 = L.startBand[cb1];
    // pair lengths: region 0 = [0,n0), region 1 = [n0,n1), region 2 = [n1,nbig)
    int b[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const int end = (nbig > n1) ? nbig : n1;      // region 1 is counted to n1 even beyond nbig
    for (int j = 2 * LANE; j < end; j += 128) {
        int x = ix[j], y = ix[j + 1];
        if (j < n0) { for (int k = 0; k < c0.n; k++) b[0][k] += pair_len(L, c0.t[k], x, y); }
        else if (j < n1) { if (bt == 0) for (int k = 0; k < c1.n; k++) b[1][k] += pair_len(L, c1.t[k], x, y); }
        else { for (int k = 0; k < c2.n; k++) b[2][k] += pair_len(L, c2.t[k], x, y); }
    }
    int qa = 0, qb = 0;
    for (int q = LANE; q < nquads; q += 64) {
        const int *v = ix + nbig + 4 * q;
        int pop = v[0] + v[1] + v[2] + v[3];
        qa += L.quada_len[((v[0] << 3) + (v[1] << 2) + (v[2] << 1) + v[3]) & 15] + pop;
        qb += 4 + pop;
    }
    int bits = 0, tab[4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const Cand &c = (r == 0) ? c0 : (r == 1 ? c1 : c2);
        int b0 = hx_wave_sum(b[r][0]) & 0xFFFF, b1 = hx_wave_sum(b[r][1]) & 0xFFFF;
        int b2 = hx_wave_sum(b[r][2]) & 0xFFFF, b3 = hx_wave_sum(b[r][3]) & 0xFFFF;
        int len = (r == 0) ? n0 : (r == 1 ? n1 - n0 : nbig - n1);
        int best = 0, idx = 0;
        if (c.n != 0 && len > 0 && !(r == 1 && bt != 0)) {
            if (b0 < b1) { best = b0; idx = 0; } else { best = b1; idx = 1; }
            if (c.n == 4) {
                if (b2 <= best) { best = b2; idx = 2; }
                if (b3 <= best) { best = b3; idx = 3; }
            }
        }
        bits += best;
        tab[r] = c.t[idx];
    }
    if (bt != 0) tab[1] = tab[2];
    qa = hx_wave_sum(qa);
    qb = hx_wave_sum(qb);
    int qidx = 0;
    if (nquads > 0) { if (qa < qb) { bits += qa; qidx = 0; } else { bits += qb; qidx = 1; } }
    tab[3] = qidx;
    if (LANE == 0) {
        L.hs_table[ch][0] = tab[0]; L.hs_table[ch][1] = tab[1]; L.hs_table[ch][2] = tab[2]; L.hs_table[ch][3] = tab[3];
        L.hs_cbreg[ch][0] = cb0; L.hs_cbreg[ch][1] = cb1; L.hs_cbreg[ch][2] = cb2;
        L.hs_nbig[ch] = nbig; L.hs_nquads[ch] = nquads; L.hs_bits[ch] = bits;
        L.huff_bits[ch] = bits;
    }
    return bits;
}

__device__ int count_bits(AllocLds &L, const HxParams *p, const int *ncb)
{
    int bits = count_bits_ch(L, p, 0, ncb[0]);
    bits += count_bits_ch(L, p, 1, ncb[1]);
    SYNC();
    return bits;
}

#include "hx_alloc2.inc"
#include "hx_alloc_short.inc"
#include "hx_alloc3.inc"
