/*
 * hxo - CPU ORACLE for the MI355X batched MP3 encoder.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, single-stream, plain-C restatement of the MPEG-1 Layer III stereo encode
 * path of the Helix encoder (reference: /root/reference/hmp3/src, cited per function
 * as file:line).  It evaluates every floating-point expression in the same order as
 * the reference so that it is bit-exact against it; it is pinned by
 *   (a) whole-stream byte equality with the reference built under oracle/_ref
 *       (tests/test_oracle_vs_ref.py, live when oracle/_ref exists) and
 *   (b) the golden vectors under tests/golden/ (generated from oracle/_ref by
 *       tools/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library.  The product (hmp3_amd/) never includes, links or calls anything here.
 *
 * Scope: MPEG-1 (32/44.1/48 kHz) and MPEG-2 LSF (16/22.05/24 kHz); stereo, joint stereo (with
 * intensity coding at the low bitrates), dual channel and mono; long and short blocks, CBR and
 * VBR, -HF high-frequency extension.
 */
#ifndef HXO_H
#define HXO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order and meaning as the reference's E_CONTROL (pub/encapp.h:42-72). */
typedef struct {
    int mode, bitrate, samprate, nsbstereo, filter_select, freq_limit, nsb_limit;
    int layer, cr_bit, original, hf_flag, vbr_flag, vbr_mnr, vbr_br_limit, vbr_delta_mnr;
    int chan_add_f0, chan_add_f1, sparse_scale;
    int mnr_adjust[21];
    int cpu_select, quick, test1, test2, test3, short_block_threshold;
} hxo_control;

/* per granule/channel side information (pub/l3e.h:72-96) */
typedef struct {
    int part2_3_length, big_values, global_gain, scalefac_compress;
    int window_switching_flag, block_type, mixed_block_flag;
    int table_select[3], subblock_gain[3];
    int region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    int aux_nquads, aux_bits, aux_not_null, aux_nreg[3];
    int block_type_prev, short_flag_current, short_flag_next;
} hxo_gr;

typedef struct { int l[23]; int s[3][13]; } hxo_scalefact;
typedef struct { float sig, mask; } hxo_sigmask;

/* psy-model tables for one block class (pub/mp3enc.h:214-225) */
typedef struct {
    int npart;              /* number of spreading rows (spd_cntl[64].count) */
    int npart_e;            /* number of energy partitions (nsum[66]) */
    int nsum[64];           /* lines per partition */
    int cnt[64], off[64];   /* spreading rows */
    float w[2200];          /* long: [0..63] abs threshold, [128..] rows; short: rows from 0 */
} hxo_psytab;

/* everything resolved at init (mp3enc.cpp:220-870, bitallo3.cpp:288-480) */
typedef struct {
    hxo_control ec;                 /* as echoed by L3_audio_encode_info_ec */
    int h_mode, h_sr_index, h_br_index, h_cr, h_original;
    unsigned char head[4];
    int totbitrate, samprate;
    int nband, nsb, nsb_limit, nsb_limitMS[2], band_limit, band_limit_stereo;
    int framebytes, remainder, divisor, main_framebytes, side_bytes, sf_bit_max, AveTargetBits;
    int ms_flag, hf_flag, vbr_flag, short_block_threshold, filter_dc;
    int nchan;                  /* 1 = mono (mode 3), 2 = stereo / joint stereo / dual channel */
    int is_flag;                /* joint stereo with an intensity part (nsf[1] < nsf[0]) */
    int alloc1;                 /* 1: the first-generation allocator (CBitAllo1) codes this stream: intensity stereo or dual channel */
    int h_id;                   /* 1 = MPEG-1 (32 / 44.1 / 48 kHz), 0 = MPEG-2 LSF (16 / 22.05 / 24 kHz): one granule per frame */
    int tix;                    /* row of the band tables: h_sr_index + 3 * (1 - h_id) */
    float filter_alpha;
    int ivbr_min, ivbr_max, vbr_main_framebytes[16], vbr_framebytes[16], vbr_pool_target;
    int initialMNR, test1, taperNT[22];
    /* band tables */
    int nBand_l_iso[22];            /* ISO widths (psy model) */
    int nBand_l[22], startBand_l[24], nBand_s[13], startBand_s[14];  /* allocator's copy ([21]=100 with -HF) */
    int nsf[2], nsf2[2], nsf3[2], nbmax[2], nbmax2[2], nbmax3[2];
    int look_log_cbwmb[22];
    float rnBand_l[22];
    /* transforms */
    float dct_coef[31];             /* 32-pt DCT butterflies 16+8+4+2+1 */
    float win[4][36], csa[2][8];
    float m18_w[18], m18_w2[9], m18_c[9][4];
    float m6_v[6], m6_v2[3], m6_c87;
    /* psy */
    hxo_psytab psyL, psyS;
    /* quantiser lookups (bitallo3.cpp:366-375) */
    float look_gain[128], look_34igain[128], look_ix43[256];
    /* short-block allocator init (bitallos.cpp:128-200) */
    int nsfs, nbmax_s, look_log_cbwmb_s[16];
} hxo_params;

/* carried state of the first-generation allocator (pub/bitallo1.h:80-140) */
typedef struct {
    int call_count, bitadjust, bitadjust_save[2];
    int gsf[2][21], gsf_save[2][21], sf[2][21];
    float running_a, ave_alpha_nmr, alpha_nmr;
} hxo_a1;
/* its tables (bitallo1.cpp:107-211, 444-543) */
typedef struct {
    int nsf[2], nBand[21], startBand[22], ill_is_pos;
    float look_log_cbw[21], look_f_ixmax[256], look_f_ix[256], look_f_big_ixmax[256], look_f_big_ix[256];
    int look_bits[256], look_is_pos[34];
    float gz_con0, gz_con1, gz_con2, con707, Ssb[21];
} hxo_a1tab;

typedef struct {
    /* input filter + polyphase history (filter2.c, pub/mp3enc.h:241-243) */
    float dc[2];
    float buf[2][2192 + 1152];
    float sample[2][4][576];
    int igrx;
    int attack_buf[2][32];
    /* per granule slot (igr = 0/1) block switching flags; ch 0 is authoritative */
    int block_type[2], block_type_prev[2], short_flag_current[2], short_flag_next[2];
    float xr[2][2][576];            /* [gr][ch] */
    int xr_clear_flag[2][2];
    float ecsave[2][64];
    hxo_sigmask sig_mask[2][36];
    int ix[2][576];
    unsigned char signx[2][576];
    hxo_scalefact sf[2][2];         /* [gr][ch] */
    hxo_gr gr[2][2];                /* [gr][ch] */
    int scfsi[2];
    int sf_save[2][21];
    /* long-block allocator carried state (pub/bitallo3.h) */
    int MNR, PoolFraction, call_count, ms_correlation_memory, NTadjust[2][22];
    int huff_bits[2], ixmax[2][22];
    int hf_quant, hf_quant_stereo[2], gsf_hf, gsf_hf_stereo[2];
    /* short-block allocator carried state */
    int s_call_count;
    hxo_a1 a1;
    /* bit reservoir / frame assembly (mp3enc.cpp:2230-2333) */
    int padcount;
    unsigned main_tot, main_sent, mf_tot;
    int main_bytes, main_p0, main_p1;
    unsigned side_p0, side_p1;
    unsigned frame_main_pos[32];
    int frame_mf_bytes[32];
    unsigned char mode_ext_buf[32], br_index_buf[32], side_buf[32][32];
    unsigned char main_buf[16384 + 512 + 1440 + 256];
    unsigned tot_frames_out, tot_bytes_out;
    int ave_tot_bytes_out;
    int byte_pool, byte_min, byte_max;
    /* last frame's decision (for tests) */
    int last_ms, last_attack[2][2], last_ms_metric[2];
} hxo_state;

/* per-frame taps for stage-wise tests (filled when hxo_set_debug() has been called) */
typedef struct {
    float sample_new[2][2][576];    /* [gr][ch] subband granule produced by this call */
    float xr_pre[2][2][576];        /* [gr][ch] spectrum before the allocator mutates it */
    float etab[2][2][64], thr[2][2][64];    /* psy: partition energy+ATH, unclamped threshold */
    float mask[2][2][22];
    int block_type[2], attack[2][2];
    int ms, ms_metric[2], byte_pool, MNR_after;
    hxo_gr gr[2][2];
    int sf[2][2][22];
    int ix[2][2][576];
    unsigned char signx[2][2][576];
    int scfsi[2];
    int main_bytes;
} hxo_frame_debug;

typedef struct hxo_encoder {
    hxo_params p;
    hxo_state s;
    hxo_a1tab a1t;
    hxo_frame_debug *dbg;
    unsigned char *packet;      /* set for the duration of hxo_encode_frame_packet */
    int packet_bytes, packet_bytes2[2];    /* MPEG-2: two packets per call, back to back */
} hxo_encoder;
void hxo_set_debug(hxo_encoder *e, hxo_frame_debug *d);
int hxo_sizeof_frame_debug(void);

/* --- API --- */
hxo_encoder *hxo_new(void);
void hxo_free(hxo_encoder *e);
/* returns bytes of float PCM consumed per call (9216) or 0 on failure (mp3enc.cpp:220) */
int hxo_init(hxo_encoder *e, const hxo_control *ec);
/* one frame: pcm = 2304 floats interleaved L/R at int16 scale, oldest first.
   returns bytes written to out (0, or one or more whole frames) (mp3enc.cpp:2031) */
int hxo_encode_frame(hxo_encoder *e, const float *pcm, unsigned char *out);
/* 16-bit entry (MP3_audio_encode with source_bits=16, mp3enc.cpp:2812; srcc.cpp:824-828) */
int hxo_encode_frame_s16(hxo_encoder *e, const int16_t *pcm, unsigned char *out);
int hxo_encode_frame_packet(hxo_encoder *e, const float *pcm, unsigned char *out, unsigned char *packet, int nbytes[2]);
unsigned hxo_frames_out(const hxo_encoder *e);
unsigned hxo_bytes_out(const hxo_encoder *e);
void hxo_default_control(hxo_control *ec);      /* test/tomp3.cpp:357-384 */
int hxo_sizeof_encoder(void);

/* --- stage functions exposed for stage-wise (teacher-forced) tests --- */
int   hxo_mblog(float x);                       /* l3math.c:228 */
float hxo_mbexp(int mb);                        /* l3math.c:342 */
void  hxo_pow34(const float *x, float *y, int n);   /* pow34.c:132 */
void  hxo_polyphase_granule(const hxo_params *p, const float *vbuf, float *samp);   /* sbt.c:293 */
void  hxo_freq_invert(float *samp, int nsb);    /* hwin.c:282 */
void  hxo_hybrid_long(const hxo_params *p, const float *prev, const float *cur, float *xr, int btype, int nsb, int clear); /* hwin.c:147 */
void  hxo_hybrid_short(const hxo_params *p, const float *prev, const float *cur, float *xr, int nsb);   /* hwin.c:228 */
void  hxo_antialias(const hxo_params *p, float *xr, int nsb);   /* hwin.c:298 */
int   hxo_attack_detect(const float *samp, int eng[32], int short_flag_prev);   /* detect.c:53 */
int   hxo_attack_detect_lsf(const float *samp, int eng[32], int short_flag_prev);       /* detect.c:147 */
void  hxo_psy_long(const hxo_params *p, const float *xr, float *ecsave, hxo_sigmask *sm, int block_type);   /* emap.c:96 + spdsmr.c:188 */
void  hxo_psy_short(const hxo_params *p, const float *xr, float *ecsave, hxo_sigmask *sm, int block_type_prev);   /* emap.c:61 + spdsmr.c:64 */
int   hxo_ms_metric_long(hxo_encoder *e, const float x[2][576]);    /* bitallo3.cpp:682 */
/* the long-block allocator for one granule (both channels); mutates xr (bitallo3.cpp:484) */
void  hxo_bitallo_long(hxo_encoder *e, float xr[2][576], hxo_sigmask sm[2][36],
                       int min_bits, int target_bits, int max_bits, int bit_pool,
                       hxo_scalefact sf_out[2], hxo_gr gr[2], int ms_flag);
/* Huffman bit count + table/region choice for one channel (bitalloc.cpp:470) */
typedef struct { int table[4]; int cbreg[3]; int nbig, nquads, bits; } hxo_huffsel;
int   hxo_count_bits(const hxo_params *p, const int *ixmax, const int *ix, int ncb, int opti, int block_type, hxo_huffsel *out);
void  hxo_huffsel_to_gr(const hxo_params *p, const hxo_huffsel *s, hxo_gr *g);  /* bitalloc.cpp:758 */

/* bit writer (l3pack.c:107-152) */
typedef struct { unsigned char *buf, *buf0; int room; int bitbuf; int bit_pos_start; } hxo_bitw;
void hxo_bw_init(hxo_bitw *w, unsigned char *out);
void hxo_bw_put(hxo_bitw *w, unsigned x, int n);
int  hxo_bw_flush(hxo_bitw *w);
int  hxo_pack_sf_long_scfsi(hxo_bitw *w, int sf_save[21], const hxo_scalefact *sf, int igr, int *scfsi, int not_null); /* l3pack.c:421 */
int  hxo_pack_sf_long(hxo_bitw *w, const hxo_scalefact *sf);      /* l3pack.c:157 */
int  hxo_pack_sf_short(hxo_bitw *w, const hxo_scalefact *sf);     /* l3pack.c:218 */
int  hxo_pack_huff(hxo_bitw *w, const hxo_gr *g, const int *ix, const unsigned char *sign);  /* l3pack.c:946 */
void hxo_pack_side(unsigned char out[32], int mode, const int scfsi[2], hxo_gr gr[2][2], int nchan);      /* l3pack.c:1123 */
void hxo_pack_side_lsf(unsigned char out[32], int mode, hxo_gr gr[2], int nchan);      /* l3pack.c:1189 */
int  hxo_pack_sf_lsf(hxo_bitw *w, const hxo_scalefact *sf, int block_type);           /* l3pack.c:561,732 */

#ifdef __cplusplus
}
#endif
#endif
