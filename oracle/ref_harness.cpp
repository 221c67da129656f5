/*
 * ref_harness.cpp - C-callable shim around the REAL reference encoder, compiled only by
 * `make -C oracle ref` against the sources where they lie under /root/reference (nothing of
 * the reference is copied into this repo).  Output: oracle/_ref/libhmp3ref.so (git-ignored).
 * TEST INFRASTRUCTURE: used to pin the oracle restatement and as bench.py's cpu_baseline
 * ("kind": "reference").  Never loaded by the product.
 *
 * The reference keeps process-global scratch (bit writer, sbt buffers ...), so callers must
 * drive one encoder at a time from one thread; interleaving frame-by-frame is fine.
 */
#include <vector>
#include <cstring>
#include <cstdio>
#include <malloc.h>
#define private public
#define protected public
#include "mp3enc.h"
#include "mp3low.h"
#include "srcc.h"
#undef private
#undef protected

extern "C" {

/* the reference's sample-format / sample-rate converter on its own (pins hmp3_amd/csrc/hx_src.cpp) */
void *ref_src_new(void) { return new Csrc; }
void ref_src_free(void *h) { delete (Csrc *) h; }
int ref_src_init(void *h, int source, int channels, int bits, int is_float, int target, int target_channels, int *cutoff)
{
    return ((Csrc *) h)->sr_convert_init(source, channels, bits, is_float, target, target_channels, cutoff);
}
int ref_src_convert(void *h, unsigned char *xin, float *yout, int *out_bytes)
{
    IN_OUT x = ((Csrc *) h)->sr_convert(xin, yout);
    *out_bytes = x.out_bytes;
    return x.in_bytes;
}
/* CMp3Enc::MP3_audio_encode_init / MP3_audio_encode with every argument */
struct ZeroHeap;
int ref_init_mp3(void *h, E_CONTROL *ec, int bits, int is_float, int mpeg_select, int mono_convert)
{
    mallopt(M_PERTURB, 0xFF);
    struct Restore { ~Restore() { mallopt(M_PERTURB, 0); } } restore;
    return ((CMp3Enc *) h)->MP3_audio_encode_init(ec, bits, is_float, mpeg_select, mono_convert);
}
int ref_encode_mp3(void *h, unsigned char *pcm, unsigned char *out, int *in_bytes)
{
    IN_OUT x = ((CMp3Enc *) h)->MP3_audio_encode(pcm, out);
    *in_bytes = x.in_bytes;
    return x.out_bytes;
}

/* The reference reads members that its constructors and init functions never set (found by tools/fuzz_oracle_vs_ref.py
 * --a1: the first-generation allocator's output for impulses on silence depended on what earlier encoders of the
 * process had left on the heap).  A process that creates one encoder - the reference's own command line - gets them
 * from fresh, zero-filled pages; this harness pins that behaviour for every encoder it creates: while the encoder
 * object and the objects its init function allocates are created, glibc's allocator hands out zero-filled blocks
 * (M_PERTURB with 0xFF fills an allocated block with the complement, 0x00). */
struct ZeroHeap { ZeroHeap() { mallopt(M_PERTURB, 0xFF); } ~ZeroHeap() { mallopt(M_PERTURB, 0); } };
void *ref_new(void) { ZeroHeap z; return new CMp3Enc; }
void ref_free(void *h) { delete (CMp3Enc *) h; }
int ref_init(void *h, E_CONTROL *ec) { ZeroHeap z; return ((CMp3Enc *) h)->L3_audio_encode_init(ec); }
int ref_init_s16(void *h, E_CONTROL *ec) { ZeroHeap z; return ((CMp3Enc *) h)->MP3_audio_encode_init(ec, 16, 0, 0, 0); }
int ref_encode(void *h, float *pcm, unsigned char *out) { return ((CMp3Enc *) h)->L3_audio_encode(pcm, out).out_bytes; }
int ref_encode_packet(void *h, float *pcm, unsigned char *out, unsigned char *packet, int *nbytes)
{
    int nb[2] = {0, 0};
    int n = ((CMp3Enc *) h)->L3_audio_encode_Packet(pcm, out, packet, nb).out_bytes;
    nbytes[0] = nb[0]; nbytes[1] = nb[1];
    return n;
}
int ref_encode_s16(void *h, short *pcm, unsigned char *out) { return ((CMp3Enc *) h)->MP3_audio_encode((unsigned char *) pcm, out).out_bytes; }
unsigned ref_frames(void *h) { return ((CMp3Enc *) h)->L3_audio_encode_get_frames(); }
void ref_info_ec(void *h, E_CONTROL *ec) { ((CMp3Enc *) h)->L3_audio_encode_info_ec(ec); }

/* encode nframes of int16 stereo PCM in one call (timing loop for bench.py's cpu_baseline) */
long ref_encode_stream_s16(void *h, short *pcm, int nframes, unsigned char *out, long out_cap)
{
    CMp3Enc *e = (CMp3Enc *) h;
    long n = 0;
    for (int f = 0; f < nframes; f++) {
        if (out_cap - n < 8192) n = 0;     /* timing use: wrap instead of overflowing */
        n += e->MP3_audio_encode((unsigned char *) (pcm + 2304 * f), out + n).out_bytes;
    }
    return n;
}

/* snapshot of the members the oracle mirrors (pub/mp3enc.h:241-291, pub/bitallo3.h) */
typedef struct {
    float sample[2][4][576];
    float xr[2][2][576];
    float sig_mask[2][36][2];
    int ix[2][576];
    unsigned char signx[2][576];
    int sf_l[2][2][23];
    int gr[2][2][26];
    int scfsi[2];
    int attack_buf[2][32];
    float ecsave[2][64];
    int igrx, byte_pool, MNR, PoolFraction, call_count, ms_correlation_memory;
    int NTadjust[2][22];
    int nsb_limit, nsb_limitMS[2], band_limit, AveTargetBits, main_framebytes, initialMNR, nsf[2];
} ref_dump_t;

void ref_dump(void *h, ref_dump_t *d)
{
    CMp3Enc *e = (CMp3Enc *) h;
    CBitAllo3 *b = (CBitAllo3 *) e->BitAllo;
    memcpy(d->sample, e->sample, sizeof(d->sample));
    memcpy(d->xr, e->xr, sizeof(d->xr));
    memcpy(d->sig_mask, e->sig_mask, sizeof(d->sig_mask));
    memcpy(d->ix, e->ix, sizeof(d->ix));
    memcpy(d->signx, e->signx, sizeof(d->signx));
    for (int g = 0; g < 2; g++)
        for (int c = 0; c < 2; c++) {
            memcpy(d->sf_l[g][c], e->sf[g][c].l, sizeof(int) * 23);
            memcpy(d->gr[g][c], &e->side_info.gr[g][c], sizeof(int) * 26);
        }
    d->scfsi[0] = e->side_info.scfsi[0];
    d->scfsi[1] = e->side_info.scfsi[1];
    memcpy(d->attack_buf, e->attack_buf, sizeof(d->attack_buf));
    memcpy(d->ecsave[0], e->ecsave[0][0], sizeof(float) * 64);
    memcpy(d->ecsave[1], e->ecsave[1][0], sizeof(float) * 64);
    d->igrx = e->igrx;
    d->byte_pool = e->byte_pool;
    d->MNR = b->MNR;
    d->PoolFraction = b->PoolFraction;
    d->call_count = b->call_count;
    d->ms_correlation_memory = b->ms_correlation_memory;
    memcpy(d->NTadjust, b->NTadjust, sizeof(d->NTadjust));
    d->nsb_limit = e->nsb_limit;
    d->nsb_limitMS[0] = e->nsb_limitMS[0];
    d->nsb_limitMS[1] = e->nsb_limitMS[1];
    d->band_limit = e->band_limit;
    d->AveTargetBits = e->AveTargetBits;
    d->main_framebytes = e->main_framebytes;
    d->initialMNR = b->initialMNR;
    d->nsf[0] = b->nsf[0];
    d->nsf[1] = b->nsf[1];
}

/* psy tables as generated by amod_initLong/Short for the handle's configuration */
void ref_psy_tables(void *h, int *nsumL, int *cntoffL, float *wL, int *nsumS, int *cntoffS, float *wS)
{
    CMp3Enc *e = (CMp3Enc *) h;
    memcpy(nsumL, e->nsum, sizeof(int) * 68);
    memcpy(cntoffL, e->spd_cntl, sizeof(int) * 130);
    memcpy(wL, e->w_spd, sizeof(float) * 2200);
    memcpy(nsumS, e->nsumShort, sizeof(int) * 68);
    memcpy(cntoffS, e->spd_cntlShort, sizeof(int) * 130);
    memcpy(wS, e->w_spdShort, sizeof(float) * 1000);
}

/* stage functions of the reference, callable on their own (mp3low.h, pub/balow.h) */
int ref_mbLogC(float x) { return mbLogC(x); }
float ref_mbExp(int x) { return mbExp(x); }
void ref_fpow34(float *x, float *y, int n) { vect_fpow34(x, y, n); }
void ref_sbt_L3(float *vbuf, float *samp) { sbt_init(); sbt_L3(vbuf, samp); }
void ref_tables_init(int sr_index, int band_limit) { L3table_init(sr_index, 1, band_limit); }
void ref_FreqInvert(float *y, int nsb) { FreqInvert(y, nsb); }
void ref_hybridLong(float *x1, float *x2, float *y, int bt, int n, int clear) { hybridLong(x1, x2, y, bt, n, clear); }
void ref_hybridShort(float *x1, float *x2, float *y, int n) { hybridShort(x1, x2, y, n); }
void ref_antialias(float *x, int n) { antialias(x, n); }
int ref_attack(float *samp, int *eng, int prev) { return attack_detectSBT_igr(samp, eng, prev); }

}   /* extern "C" */
