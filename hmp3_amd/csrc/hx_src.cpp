// hx_src.cpp - sample-format and sample-rate conversion in front of the encoder (host side), the
// part of the reference's Csrc that CMp3Enc::MP3_audio_encode runs before every frame
// (reference srcc.cpp:83-214 filter plan, :224-357 factors, :333-397 + :501-611
// filter bank, :730-792 init, :795-908 convert; srccf.cpp the fifteen filter loops).
//
// One call turns the caller's PCM into the 1152 samples per channel the encoder takes and reports
// how many input bytes it used.  Five cases by the rate ratio:
//   0  same rate: copy (or down-mix)           1  exactly 1:2 up: insert midpoints
//   2  other up-sampling: linear interpolation with a table of fractions
//   3  down-sampling, small filter bank: polyphase FIR, one filter per output phase
//   4  down-sampling, bank too large: linear interpolation up to an intermediate rate, then case 3
// times three channel layouts (mono, stereo, stereo summed to mono).  Results are bit-identical to
// the reference: same operand types (several expressions there are evaluated in double), same
// summation order, same state carried from call to call.
#include "hx_src.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

int taps_for(int source, int target)
{
    if (source <= target) return 1;
    int n = (12 * source + target / 2) / target;
    if (n > 48) n = 48;
    if (n < 1) n = 1;
    return (n & ~1) | 1;
}

int common_factor(int s, int t)
{
    int cf = 1;
    for (int i = 2; i <= t; i++) {
        if (s % i || t % i) continue;
        cf *= i; s /= i; t /= i;
        i = 1;
    }
    return cf;
}

// intermediate rate of the two-stage plan: source * (i + 1) / i for the first i >= 7 that divides
// the reduced ratio and leaves a second stage whose filter bank fits (0 = no plan)
int intermediate_rate(int source, int target)
{
    if (source <= target) return source;
    const int cf = common_factor(source, target), s = source / cf, t = target / cf;
    int up = 0, down = 0;
    for (int i = 7; i < t; i++) {
        if (s % i || t % (i + 1)) continue;
        down = i; up = i + 1;
        const int rate2 = up * source / down;
        if (taps_for(rate2, target) * (t / up) <= 780) break;
    }
    return down ? up * source / down : 0;
}

// one low-pass filter of N taps centred at (N - 1) / 2 + alpha: cosine series with n terms, raised
// cosine window, unit DC gain
void design_filter(float *b, int N, int n, float alpha)
{
    const double pi = 4.0 * atan(1.0), t = pi / (2 * N), scale = 1.0 / N;
    const double x = (N - 1) / 2.0 + alpha;
    for (int p = 0; p < N; p++) {
        const double wp = p == 0 ? 1.0 : 2.0;
        b[p] = 0.0f;
        for (int k = 0; k < n; k++) b[p] += (float) (scale * wp * cos(t * x * (2 * k + 1)) * cos(t * p * (2 * k + 1)));
    }
    const double tw = 2.0 * pi / N;
    for (int i = 0; i < N; i++) {
        double w = 0.5 * (1.0 - cos((i + 0.5) * tw));
        w = .5 + .5 * w;
        b[i] = (float) (w * b[i]);
    }
    float sum = 0.0f;
    for (int i = 0; i < N; i++) sum += b[i];
    for (int i = 0; i < N; i++) b[i] = b[i] / sum;
}

// the bank: nfilters filters (or interpolation fractions when ntaps == 1) in the order the output
// phases use them
void design_bank(float *a, int ntaps, int ncutoff, int nfilters, int m)
{
    int am = 0;
    for (int i = 0; i < nfilters; i++, a += ntaps) {
        float alpha = ((float) am) / nfilters;
        if (ntaps == 1) a[0] = alpha;
        else if (ntaps == 2) { a[0] = 1.0f - alpha; a[1] = alpha; }
        else {
            alpha = alpha + 0.5f / nfilters - 0.5f;
            design_filter(a, ntaps, ncutoff, alpha);
        }
        am += m;
        if (am >= nfilters) am -= nfilters;
    }
}

}  // namespace

struct hx_src {
    int ncase = 0, layout = 0;          // layout: 0 mono, 1 stereo, 2 stereo summed to mono
    int channels = 1, bits = 16, is_float = 0, convert_frames = 1152;
    int out_bytes = 0;
    // stage 1 (case 4 only) and main stage: input step k + m/n per output, bank position
    int k1 = 0, m1 = 0, n1 = 0, ntaps1 = 0, totcoef1 = 0, am1 = 0, ic1 = 0;
    int k = 0, m = 0, n = 0, ntaps = 0, totcoef = 0, am = 0, ic = 0;
    int nbuf = 0, kbuf = 0;
    float coef1[21];
    float coef[1280];
    float buf[128 + 64], buf2[128 + 64];
    float *staged = nullptr;            // the call's input as float at int16 scale

    // advance the bank and the fractional input position after one output sample; true = one more input sample
    bool step() { am -= m; if (am <= 0) { am += n; return true; } return false; }
    bool step1() { am1 -= m1; if (am1 <= 0) { am1 += n1; return true; } return false; }

    int plan(int source0, int target);
    int run_mono(const float *x, float *y);
    int run_stereo(const float (*x)[2], float (*y)[2]);
    int run_downmix(const float (*x)[2], float *y);
    int refill_mono(const float *x);
    int refill_stereo(const float (*x)[2]);
    int refill_downmix(const float (*x)[2]);
    void make_room();
};

int hx_src::plan(int source0, int target)
{
    int source = source0, target1 = source0;
    const int mem = target / common_factor(source, target) * taps_for(source, target);
    if (source == target) ncase = 0;
    else if (2 * source == target) ncase = 1;
    else if (source < target) ncase = 2;
    else ncase = mem <= 780 ? 3 : 4;
    if (ncase == 4) {
        source = intermediate_rate(source, target);
        if (source <= 0) return 0;
        target1 = source;
    }
    ntaps1 = taps_for(source0, target1);
    n1 = target1 / common_factor(source0, target1);
    k1 = source0 / target1;
    m1 = (n1 * source0 - target1 * n1 * k1) / target1;
    totcoef1 = ntaps1 * n1;
    int ncutoff1 = (int) (0.90 * ntaps1 * target1 / source0 + 0.50);
    if (ncutoff1 > ntaps1) ncutoff1 = ntaps1;
    ntaps = taps_for(source, target);
    n = target / common_factor(source, target);
    k = source / target;
    m = (n * source - target * n * k) / target;
    totcoef = ntaps * n;
    int ncutoff = (int) (0.90 * ntaps * target / source + 0.50);
    if (ncutoff > ntaps) ncutoff = ntaps;
    am = n; ic = 0;
    int minbuf = (int) (1152.0 * source0 / target + (ntaps - 1) + 1);
    if (ncase == 4) minbuf += (128 + 4);
    am1 = n1; ic1 = 0; nbuf = 0;
    if (totcoef1 > (int) (sizeof(coef1) / sizeof(float)) || totcoef > (int) (sizeof(coef) / sizeof(float))) return 0;
    design_bank(coef1, ntaps1, ncutoff1, n1, m1);
    design_bank(coef, ntaps, ncutoff, n, m);
    return minbuf;
}

void hx_src::make_room()
{
    nbuf -= kbuf;
    if (nbuf > 0) {
        memmove(buf, buf + kbuf, sizeof(float) * nbuf);
        if (layout == 1) memmove(buf2, buf2 + kbuf, sizeof(float) * nbuf);
    }
    kbuf = 0;
}

// ---- stage 1 of case 4: 128 more samples at the intermediate rate; returns input samples used ----
int hx_src::refill_mono(const float *x)
{
    make_room();
    int j = 0;
    for (int i = 0; i < 128; i++) {
        buf[nbuf++] = (float) x[j] + coef1[ic1] * ((float) x[j + 1] - (float) x[j]);
        if (++ic1 >= totcoef1) ic1 = 0;
        if (step1()) j++;
    }
    return j;
}

int hx_src::refill_stereo(const float (*x)[2])
{
    make_room();
    int j = 0;
    for (int i = 0; i < 128; i++) {
        buf[nbuf] = (float) x[j][0] + coef1[ic1] * ((float) x[j + 1][0] - (float) x[j][0]);
        buf2[nbuf++] = (float) x[j][1] + coef1[ic1] * ((float) x[j + 1][1] - (float) x[j][1]);
        if (++ic1 >= totcoef1) ic1 = 0;
        if (step1()) j++;
    }
    return j;
}

int hx_src::refill_downmix(const float (*x)[2])
{
    make_room();
    int j = 0;
    float a = (x[0][0] + x[0][1]) * 0.5, b = (x[1][0] + x[1][1]) * 0.5;
    for (int i = 0; i < 128; i++) {
        buf[nbuf++] = a + coef1[ic1] * (b - a);
        if (++ic1 >= totcoef1) ic1 = 0;
        if (step1()) { j++; a = b; b = (x[j + 1][0] + x[j + 1][1]) * 0.5; }
    }
    return j;
}

// ---- the three layouts; each returns the number of input sample frames consumed ----
int hx_src::run_mono(const float *x, float *y)
{
    int used = 0;
    switch (ncase) {
    case 0:
        memmove(y, x, sizeof(float) * 1152);
        return 1152;
    case 1: {       // the reference takes this path through integers
        int a = x[0], b, o = 0;
        for (int i = 0; i < 576; i += 2, o += 4) {
            b = x[i + 1];
            y[o] = (float) (a);
            y[o + 1] = (float) ((a + b) >> 1);
            a = x[i + 2];
            y[o + 2] = (float) (b);
            y[o + 3] = (float) ((a + b) >> 1);
        }
        return 576;
    }
    case 2:
        for (int i = 0; i < 1152; i++) {
            y[i] = (float) ((float) x[used] + coef[ic] * ((float) x[used + 1] - (float) x[used]));
            if (++ic >= totcoef) ic = 0;
            if (step()) used++;
        }
        return used;
    case 3:
        for (int i = 0; i < 1152; i++) {
            float u = 0.0f;
            for (int j = 0; j < ntaps; j++) u += coef[ic++] * x[used + j];
            y[i] = u;
            if (ic >= totcoef) ic = 0;
            used += k;
            if (step()) used++;
        }
        return used;
    default: {
        int thres = nbuf - ntaps;
        for (int i = 0; i < 1152; i++) {
            if (kbuf > thres) { used += refill_mono(x + used); thres = nbuf - ntaps; }
            float u = 0.0f;
            for (int j = 0; j < ntaps; j++) u += coef[ic++] * buf[kbuf + j];
            y[i] = u;
            if (ic >= totcoef) ic = 0;
            kbuf += k;
            if (step()) kbuf++;
        }
        return used;
    }
    }
}

int hx_src::run_stereo(const float (*x)[2], float (*y)[2])
{
    int used = 0;
    switch (ncase) {
    case 0:
        memmove(y, x, sizeof(float) * 2 * 1152);
        return 1152;
    case 1:
        for (int i = 0, o = 0; i < 576; i++, o += 2) {
            y[o][0] = x[i][0];
            y[o + 1][0] = (float) ((x[i][0] + x[i + 1][0]) * 0.5);
            y[o][1] = x[i][1];
            y[o + 1][1] = (float) ((x[i][1] + x[i + 1][1]) * 0.5);
        }
        return 576;
    case 2:
        for (int i = 0; i < 1152; i++) {
            y[i][0] = (float) ((float) x[used][0] + coef[ic] * ((float) x[used + 1][0] - (float) x[used][0]));
            y[i][1] = (float) ((float) x[used][1] + coef[ic] * ((float) x[used + 1][1] - (float) x[used][1]));
            if (++ic >= totcoef) ic = 0;
            if (step()) used++;
        }
        return used;
    case 3:
        for (int i = 0; i < 1152; i++) {
            float u = 0.0f, v = 0.0f;
            for (int j = 0; j < ntaps; j++) { u += coef[ic] * x[used + j][0]; v += coef[ic++] * x[used + j][1]; }
            y[i][0] = u; y[i][1] = v;
            if (ic >= totcoef) ic = 0;
            used += k;
            if (step()) used++;
        }
        return used;
    default: {
        int thres = nbuf - ntaps;
        for (int i = 0; i < 1152; i++) {
            if (kbuf > thres) { used += refill_stereo(x + used); thres = nbuf - ntaps; }
            float u = 0.0f, v = 0.0f;
            for (int j = 0; j < ntaps; j++) { u += coef[ic] * buf[kbuf + j]; v += coef[ic++] * buf2[kbuf + j]; }
            y[i][0] = u; y[i][1] = v;
            if (ic >= totcoef) ic = 0;
            kbuf += k;
            if (step()) kbuf++;
        }
        return used;
    }
    }
}

int hx_src::run_downmix(const float (*x)[2], float *y)
{
    int used = 0;
    switch (ncase) {
    case 0:
        for (int i = 0; i < 1152; i++) y[i] = (float) ((x[i][0] + x[i][1]) * 0.5);
        return 1152;
    case 1: {
        float a = x[0][0] + x[0][1], b;
        for (int i = 0, o = 0; i < 576; i += 2, o += 4) {
            b = x[i + 1][0] + x[i + 1][1];
            y[o + 1] = (float) ((a + b) * 0.25);
            y[o] = (float) (a * 0.5);
            a = x[i + 2][0] + x[i + 2][1];
            y[o + 3] = (float) ((a + b) * 0.25);
            y[o + 2] = (float) (b * 0.5);
        }
        return 576;
    }
    case 2: {
        float a = (x[0][0] + x[0][1]) * 0.5;
        float b = ((x[1][0] + x[1][1]) * 0.5) - a;
        for (int i = 0; i < 1152; i++) {
            y[i] = (float) (a + coef[ic] * b);
            if (++ic >= totcoef) ic = 0;
            if (step()) { used++; a = a + b; b = ((x[used + 1][0] + x[used + 1][1]) * 0.5) - a; }
        }
        return used;
    }
    case 3:
        for (int i = 0; i < 1152; i++) {
            float u = 0.0f;
            for (int j = 0; j < ntaps; j++) u += coef[ic++] * ((x[used + j][0] + x[used + j][1]) * 0.5);
            y[i] = u;
            if (ic >= totcoef) ic = 0;
            used += k;
            if (step()) used++;
        }
        return used;
    default: {
        int thres = nbuf - ntaps;
        for (int i = 0; i < 1152; i++) {
            if (kbuf > thres) { used += refill_downmix(x + used); thres = nbuf - ntaps; }
            float u = 0.0f;
            for (int j = 0; j < ntaps; j++) u += coef[ic++] * buf[kbuf + j];
            y[i] = u;
            if (ic >= totcoef) ic = 0;
            kbuf += k;
            if (step()) kbuf++;
        }
        return used;
    }
    }
}

extern "C" hx_src *hx_src_create(void) { return new hx_src; }

extern "C" void hx_src_destroy(hx_src *s)
{
    if (!s) return;
    delete[] s->staged;
    delete s;
}

// Csrc::sr_convert_init: returns the bytes the caller must hold before each convert call (0 = cannot convert)
extern "C" int hx_src_init(hx_src *s, int source, int channels, int bits, int is_float, int target, int target_channels,
                           int *encode_cutoff_freq)
{
    delete[] s->staged;
    *s = hx_src();
    if (is_float && bits != 32) return 0;
    if (bits != 32 && bits != 24 && bits != 16 && bits != 8) return 0;
    if (channels < 1 || channels > 2 || source < 8000 || source > 48000 || target < 5000 || target > 50400) return 0;
    if (target_channels < 1) target_channels = 1;
    if (target_channels > channels) target_channels = channels;
    s->layout = (channels == 2) ? (target_channels == 2 ? 1 : 2) : 0;
    const int min_samps = s->plan(source, target);
    if (min_samps <= 0) return 0;
    s->out_bytes = (int) sizeof(float) * target_channels * 1152;
    *encode_cutoff_freq = (int) (0.90f * (target < source ? target : source) / 2);
    s->channels = channels; s->bits = bits; s->is_float = is_float;
    s->convert_frames = 1152;
    if (source > target) s->convert_frames *= ((source / target) + 1);
    s->staged = new float[(size_t) s->convert_frames * 2];
    return min_samps * channels * bits / 8;
}

// Csrc::sr_convert: xin -> 1152 samples per output channel in yout; returns input bytes used
extern "C" int hx_src_convert(hx_src *s, const unsigned char *xin, float *yout, int *out_bytes)
{
    const int ns = s->convert_frames * s->channels;
    float *dst = s->staged;
    if (s->bits == 32 && s->is_float) { const float *p = (const float *) xin; for (int i = 0; i < ns; i++) dst[i] = (float) p[i] * 32768.0f; }
    else if (s->bits == 32) { const int *p = (const int *) xin; for (int i = 0; i < ns; i++) dst[i] = (float) (p[i] / 65536.0f); }
    else if (s->bits == 24) {
        for (int i = 0; i < ns; i++) {
            const unsigned char *b = xin + 3 * i;
            const int v = (int) (((unsigned) b[2] << 24) | ((unsigned) b[1] << 16) | ((unsigned) b[0] << 8)) >> 8;
            dst[i] = (float) ((float) v / 256.0f);
        }
    } else if (s->bits == 16) { const short *p = (const short *) xin; for (int i = 0; i < ns; i++) dst[i] = (float) p[i]; }
    else { for (int i = 0; i < ns; i++) dst[i] = (((float) xin[i]) - 128.0f) * (256.0f); }
    typedef float pair[2];
    int frames;
    if (s->layout == 0) frames = s->run_mono(dst, yout);
    else if (s->layout == 1) frames = s->run_stereo((const pair *) dst, (pair *) yout);
    else frames = s->run_downmix((const pair *) dst, yout);
    int in_bytes = (int) sizeof(float) * frames * s->channels;          // as float bytes, then scaled to the source width
    if (s->bits == 8) in_bytes /= 4;
    if (s->bits == 16) in_bytes /= 2;
    else if (s->bits == 24) in_bytes = in_bytes * 3 / 4;
    if (out_bytes) *out_bytes = s->out_bytes;
    return in_bytes;
}
