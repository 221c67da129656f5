/* hxo_enc.c - ORACLE (test infrastructure): parameter resolution, per-frame driver,
 * bit reservoir and frame assembly.  Restates mp3enc.cpp:220-870 (init), :1045-1114
 * (transform), :1398-1440 (block type), :1492-1597 (encode_jointB), :2106-2333 (CBR/VBR
 * frame drivers), setup.c:189-292, filter2.c:62-153. */
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <assert.h>
#include "hxo_int.h"

hxo_encoder *hxo_new(void) { return (hxo_encoder *) calloc(1, sizeof(hxo_encoder)); }
void hxo_free(hxo_encoder *e) { free(e); }
int hxo_sizeof_encoder(void) { return (int) sizeof(hxo_encoder); }
int hxo_sizeof_frame_debug(void) { return (int) sizeof(hxo_frame_debug); }
void hxo_set_debug(hxo_encoder *e, hxo_frame_debug *d) { e->dbg = d; }
extern float *hxo_tap_etab, *hxo_tap_thr;
unsigned hxo_frames_out(const hxo_encoder *e) { return e->s.tot_frames_out; }
unsigned hxo_bytes_out(const hxo_encoder *e) { return e->s.tot_bytes_out; }

/* test/tomp3.cpp:357-384 */
void hxo_default_control(hxo_control *ec)
{
    memset(ec, 0, sizeof(*ec));
    ec->mode = 1; ec->bitrate = -1; ec->samprate = 44100; ec->nsbstereo = -1; ec->filter_select = -1;
    ec->freq_limit = 24000; ec->nsb_limit = -1; ec->layer = 3; ec->cr_bit = 1; ec->original = 1;
    ec->hf_flag = 0; ec->vbr_flag = 1; ec->vbr_mnr = 50; ec->vbr_br_limit = 160; ec->vbr_delta_mnr = 0;
    ec->chan_add_f0 = ec->chan_add_f1 = 24000; ec->sparse_scale = -1; ec->cpu_select = 0;
    ec->quick = -1; ec->test1 = -1; ec->test2 = ec->test3 = 0; ec->short_block_threshold = 700;
}

static const int br_mpeg1_l3[16] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, -1};
static const int br_mpeg2_l3[16] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, -1};

/* mp3enc.cpp:964-1041 (MPEG-1 half) */
static void gen_vbr_table(hxo_params *p, int max_tot_bitrate)
{
    int i;
    if (p->h_id == 0) {     /* mp3enc.cpp:1006-1040 */
        for (i = 1; i < 15; i++) {
            int mb = 72000 * br_mpeg2_l3[i] / p->samprate;
            p->vbr_framebytes[i] = mb;
            p->vbr_main_framebytes[i] = mb - 4 - p->side_bytes;
        }
        p->vbr_framebytes[15] = p->vbr_main_framebytes[15] = 9999999;
        p->vbr_pool_target = 128;
        for (i = 14; i >= 2; i--) {
            if (max_tot_bitrate >= br_mpeg2_l3[i]) break;
            p->vbr_pool_target = (p->vbr_pool_target + 255) >> 1;
        }
        p->ivbr_max = i;
        p->ivbr_min = 1;
        p->AveTargetBits = (8 * p->vbr_main_framebytes[p->ivbr_max] / p->nchan) - p->sf_bit_max;
        return;
    }
    for (i = 1; i < 15; i++) {
        int mb = 144000 * br_mpeg1_l3[i] / p->samprate;
        p->vbr_framebytes[i] = mb;
        p->vbr_main_framebytes[i] = mb - 4 - p->side_bytes;
    }
    p->vbr_framebytes[15] = p->vbr_main_framebytes[15] = 9999999;
    p->vbr_pool_target = 256;
    for (i = 14; i >= 2; i--) {
        if (max_tot_bitrate >= br_mpeg1_l3[i]) break;
        p->vbr_pool_target = (p->vbr_pool_target + 511) >> 1;
    }
    p->ivbr_max = i;
    p->ivbr_min = 1;
    p->AveTargetBits = (8 * p->vbr_main_framebytes[p->ivbr_max] / (2 * p->nchan)) - p->sf_bit_max;
}

/* mp3enc.cpp:220-870 + setup.c:189-292 + bitallo3.cpp:288-480 */
int hxo_init(hxo_encoder *e, const hxo_control *ec_arg)
{
    static const int sr_all[8] = {22050, 24000, 16000, 1, 44100, 48000, 32000, 1};
    static const int mnrGOLD[22] = {-5, 0, 0, 0, 0, 0, 0, 0, 0, 3, 5, 5, 5, 5, 3, 0, 0, 0, -1, -8, -10, 0};
    hxo_params *p = &e->p;
    hxo_state *s = &e->s;
    hxo_control ec = *ec_arg;
    int i, j, k, d, dmin, bitrate, h_id, mode_ext, nsbstereo, nsbstereo_limit, is_flag;
    int nsb_limit_user1, nsb_limit_user2, nsb_limit_user, nsb_user_flag, freq_limit, tmp, disable_taper, MNRbias;

    hxo_math_init();
    memset(e, 0, sizeof(*e));

    if (ec.mode < 0) ec.mode = 1;
    if (ec.mode > 3) ec.mode = 3;
    if (ec.bitrate < 0) { ec.bitrate = 64; if (ec.samprate < 32000) ec.bitrate = 32; }
    if (ec.mode == 2) ec.vbr_flag = 0;
    if (ec.vbr_mnr < 0) ec.vbr_mnr = 0;
    if (ec.vbr_mnr > 150) ec.vbr_mnr = 150;
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

    /* setup_header (setup.c:189-292) */
    if (ec.layer != 3) return 0;
    for (k = 0, dmin = 99999, i = 0; i < 8; i++) {
        d = abs(ec.samprate - sr_all[i]);
        if (d < dmin) { dmin = d; k = i; }
    }
    h_id = k >> 2;
    p->h_sr_index = k & 3;
    if (sr_all[k] == 1) return 0;
    p->h_id = h_id;
    p->tix = p->h_sr_index + 3 * (1 - h_id);
    p->h_mode = ec.mode;
    p->nchan = (ec.mode == 3) ? 1 : 2;
    mode_ext = 0;
    if (p->h_mode == 1) mode_ext = ec.nsbstereo / 4 - 1;
    if (mode_ext < 0) mode_ext = h_id ? 0 : 1;     /* setup.c:234-239 */
    if (mode_ext > 3) mode_ext = 3;
    bitrate = ec.bitrate;
    if (bitrate < 8) bitrate = 8;
    if (ec.mode != 3) bitrate = 2 * bitrate;
    if (bitrate > (h_id ? 320 : 160)) bitrate = h_id ? 320 : 160;
    p->h_br_index = 0;
    for (i = 1; (h_id ? br_mpeg1_l3 : br_mpeg2_l3)[i] >= 0; i++) if ((h_id ? br_mpeg1_l3 : br_mpeg2_l3)[i] == bitrate) p->h_br_index = i;
    p->totbitrate = bitrate;
    p->h_cr = ec.cr_bit;
    p->h_original = ec.original;
    /* pack_head_local (mp3enc.cpp:874-895): sync, id=1, layer III, no CRC */
    p->head[0] = 0xFF;
    p->head[1] = (unsigned char) (0xF0 | (h_id << 3) | (1 << 1) | 1);
    p->head[2] = (unsigned char) ((p->h_br_index << 4) | (p->h_sr_index << 2));
    p->head[3] = (unsigned char) ((p->h_mode << 6) | (mode_ext << 4) | (p->h_cr << 3) | (p->h_original << 2));

    p->nband = hxo_sfb_long_edge(p->tix, 21);
    p->nsb = (p->nband + 17) / 18;
    if (h_id == 0) {
        nsbstereo = 7 * p->totbitrate / 16 - 7;
        nsbstereo = HXO_MIN(nsbstereo, 32);
        nsbstereo = HXO_MAX(nsbstereo, 3);
        if (p->totbitrate >= 48) nsbstereo = 32;
    } else {
        nsbstereo = 12 * p->totbitrate / 32 - 20;
        nsbstereo = HXO_MIN(nsbstereo, 32);
        nsbstereo = HXO_MAX(nsbstereo, 3);
        if (p->totbitrate >= 96) nsbstereo = 32;
    }
    if (ec.vbr_flag) nsbstereo = 32;
    if (ec.nsbstereo > 0) {
        nsbstereo = ec.nsbstereo;
        if (nsbstereo < 3) nsbstereo = 3;
        if (nsbstereo > 32) nsbstereo = 32;
    }
    if (nsbstereo > p->nsb) nsbstereo = p->nsb;
    p->samprate = sr_all[4 * h_id + p->h_sr_index];
    p->divisor = p->samprate;
    p->sf_bit_max = 3 * (6 * 4 + 6 * 3);
    if (h_id) {
        p->framebytes = 144000 * p->totbitrate / p->divisor;
        p->remainder = (144000 * p->totbitrate) % p->divisor;
        p->side_bytes = (p->h_mode == 3) ? 17 : 32;
        p->main_framebytes = p->framebytes - 4 - p->side_bytes;
        p->AveTargetBits = 8 * p->main_framebytes / 2;
    } else {        /* one granule per frame (mp3enc.cpp:462-481) */
        p->framebytes = (144000 / 2) * p->totbitrate / p->divisor;
        p->remainder = ((144000 / 2) * p->totbitrate) % p->divisor;
        p->side_bytes = (p->h_mode == 3) ? 9 : 17;
        p->main_framebytes = p->framebytes - 4 - p->side_bytes;
        p->AveTargetBits = 8 * p->main_framebytes;
    }
    if (p->h_mode != 3) p->AveTargetBits >>= 1;
    p->AveTargetBits -= p->sf_bit_max;

    nsb_user_flag = 0;
    nsb_limit_user1 = 32;
    if (ec.nsb_limit > 0) {
        nsb_limit_user1 = HXO_MIN(ec.nsb_limit, 32);
        nsb_limit_user1 = HXO_MAX(ec.nsb_limit, (64 * 1000 + p->samprate / 2) / p->samprate);
        nsb_user_flag = 1;
    }
    nsb_limit_user2 = 32;
    if (ec.freq_limit < 24000) {
        nsb_limit_user2 = (64 * HXO_MAX(ec.freq_limit, 1000) + p->samprate / 2) / p->samprate;
        nsb_user_flag = 1;
    }
    nsb_limit_user = HXO_MIN(nsb_limit_user1, nsb_limit_user2);
    if (ec.vbr_flag && h_id == 0) {         /* mp3enc.cpp:518-533 */
        freq_limit = 7500 + 50 * ec.vbr_mnr;
        if (ec.vbr_mnr <= 5) freq_limit = 7500;
        freq_limit = HXO_MIN(freq_limit, ((int) ((0.96f * 0.5f) * p->samprate)));
        freq_limit = hxo_nearest_sf_band_freq(p->tix, p->samprate, freq_limit);
        tmp = (64 * freq_limit + (p->samprate / 2)) / p->samprate;
        freq_limit = (tmp * p->samprate) / 64;
    } else if (ec.vbr_flag) {
        freq_limit = 12000 + 80 * ec.vbr_mnr;
        if (ec.vbr_mnr <= 5) freq_limit = 12000;
        freq_limit = HXO_MIN(freq_limit, ((int) ((0.96f * 0.5f) * p->samprate)));
    } else {
        /* calc_freq_limit_L3 (mp3enc.cpp:899-935) */
        static const float factor[4] = {1.1f, 1.333f, 1.0f, 1.0f};
        float chan_bitrate = (float) p->totbitrate;
        if (p->h_mode != 3) chan_bitrate = (float) (0.5 * chan_bitrate);
        chan_bitrate = factor[p->h_mode] * chan_bitrate;
        if (p->samprate < 32000) {
            if (chan_bitrate <= 32.0f) freq_limit = (int) (752.0 + 203.0 * chan_bitrate);
            else if (chan_bitrate <= 42.7f) freq_limit = (int) (-2967.0 + 327.0 * chan_bitrate);
            else freq_limit = 11000;
        } else freq_limit = (int) (187.97 * chan_bitrate);
        if (h_id == 0) {                    /* mp3enc.cpp:537-545 */
            freq_limit = hxo_nearest_sf_band_freq(p->tix, p->samprate, freq_limit);
            tmp = (64 * freq_limit + (p->samprate / 2)) / p->samprate;
            freq_limit = (tmp * p->samprate) / 64;
        }
    }
    if (nsb_user_flag) p->nsb_limit = nsb_limit_user;
    else p->nsb_limit = (64 * HXO_MAX(freq_limit, 1000) + p->samprate / 2) / p->samprate;
    p->nsb_limit = HXO_MIN(p->nsb, p->nsb_limit);
    p->nsb_limitMS[0] = p->nsb_limitMS[1] = p->nsb_limit;
    if (p->nsb_limit < p->nsb) ec.hf_flag = 0;
    if (ec.hf_flag) {
        p->nsb_limitMS[0] = 29;
        if (nsb_user_flag) p->nsb_limitMS[0] = HXO_MIN(nsb_limit_user, 29);
    }
    if (ec.hf_flag & 2) {
        p->nsb_limitMS[1] = 29;
        if (nsb_user_flag) p->nsb_limitMS[1] = HXO_MIN(nsb_limit_user, 29);
    }
    p->band_limit = 18 * p->nsb_limit;
    if (p->band_limit > p->nband) p->band_limit = p->nband;
    nsbstereo_limit = HXO_MIN(nsbstereo, p->nsb_limit);
    if (p->h_mode == 1) p->band_limit_stereo = 18 * nsbstereo_limit;
    else p->band_limit_stereo = p->band_limit;
    if (p->band_limit_stereo > p->band_limit) p->band_limit_stereo = p->band_limit;

    /* filter2_init (filter2.c:62-76) */
    p->filter_alpha = (float) (0.001 * 44100.0 / p->samprate);
    p->filter_dc = ec.filter_select > 1 ? 1 : ec.filter_select;

    /* band tables */
    for (i = 0; i < 22; i++)
        p->nBand_l_iso[i] = p->nBand_l[i] = hxo_sfb_long_edge(p->tix, i + 1) - hxo_sfb_long_edge(p->tix, i);
    for (i = 0; i < 13; i++)
        p->nBand_s[i] = hxo_sfb_short_edge(p->tix, i + 1) - hxo_sfb_short_edge(p->tix, i);
    hxo_init_transform_tables(p);
    hxo_init_psy_long(p);
    hxo_init_psy_short(p);
    for (k = 0; k < 2; k++) for (i = 0; i < 64; i++) s->ecsave[k][i] = 1.0e20f;

    p->ms_flag = is_flag = 0;
    if (p->h_mode == 1) {
        if (nsbstereo_limit < p->nsb_limit) is_flag = 1;
        p->ms_flag = 1;
    }
    p->is_flag = is_flag;
    p->alloc1 = is_flag || p->h_mode == 2;  /* mp3enc.cpp:696-766: the first-generation allocator codes these */
    if (is_flag) ec.vbr_flag = 0;
    p->vbr_flag = ec.vbr_flag;
    if (ec.vbr_flag) gen_vbr_table(p, (ec.mode == 3 ? 1 : 2) * ec.vbr_br_limit);
    if (ec.vbr_flag) {
        p->initialMNR = 10 * ec.vbr_mnr;
        if (p->initialMNR < 210) p->initialMNR = 210;
        if (p->initialMNR > 1500) p->initialMNR = 1500;
    } else {
        tmp = p->totbitrate / p->nchan;
        if (h_id) p->initialMNR = 125 * (tmp - 32) / 8;
        else p->initialMNR = 10 * ((30 * tmp) / 8 - 70);
        if (p->initialMNR < 0) p->initialMNR = 0;
        if (p->initialMNR > 1000) p->initialMNR = 1000;
    }
    ec.vbr_delta_mnr = HXO_MIN(ec.vbr_delta_mnr, 50);
    ec.vbr_delta_mnr = HXO_MAX(ec.vbr_delta_mnr, -40);
    for (i = 0; i < 21; i++) {
        ec.mnr_adjust[i] = HXO_MIN(ec.mnr_adjust[i], 200);
        ec.mnr_adjust[i] = HXO_MAX(ec.mnr_adjust[i], -200);
    }
    p->hf_flag = ec.hf_flag;
    MNRbias = 10 * ec.vbr_delta_mnr;
    disable_taper = ec.quick;
    p->test1 = ec.test1;
    if (p->test1 < 0) p->test1 = 6;

    /* CBitAllo3::BitAlloInit (bitallo3.cpp:288-480) */
    s->MNR = p->initialMNR;
    s->PoolFraction = p->vbr_flag ? 614 : 0;
    p->nsf3[0] = p->nsf2[0] = p->nsf[0] = hxo_sfbl_limit(p->tix, p->band_limit);
    p->nsf3[1] = p->nsf2[1] = p->nsf[1] = hxo_sfbl_limit(p->tix, p->band_limit_stereo);
    if (p->hf_flag) { p->nsf2[0] = 22; p->nBand_l[21] = 100; }
    if (p->hf_flag & 2) { p->nsf3[0] = 22; p->nsf3[1] = 22; }
    for (k = 0, i = 0; i < 22; i++) { p->startBand_l[i] = k; k += p->nBand_l[i]; }
    p->startBand_l[22] = k;
    p->startBand_l[23] = 576;
    for (k = 0, i = 0; i < 13; i++) { p->startBand_s[i] = k; k += p->nBand_s[i]; }
    p->startBand_s[13] = k;
    for (j = 0; j < 2; j++) p->nbmax3[j] = p->nbmax2[j] = p->nbmax[j] = p->startBand_l[p->nsf[j]];
    if (p->hf_flag) p->nbmax2[0] = p->startBand_l[p->nsf2[0]];
    if (p->hf_flag & 2) { p->nbmax3[0] = p->startBand_l[p->nsf3[0]]; p->nbmax3[1] = p->startBand_l[p->nsf3[1]]; }
    for (i = 0; i < 128; i++) {
        p->look_gain[i] = (float) (pow(2.0, 0.25 * (i - 8)));
        p->look_34igain[i] = (float) (1.0 / pow((double) p->look_gain[i], (double) (3.0 / 4.0)));
    }
    for (i = 0; i < 256; i++) p->look_ix43[i] = (float) (i * pow((double) i, (double) (1.0 / 3.0)));
    for (i = 0; i < 21; i++) p->look_log_cbwmb[i] = (int) (100.0f * hxo_dblog((float) (double) p->nBand_l[i]));
    for (i = 0; i < 22; i++) p->taperNT[i] = 0;
    if (!disable_taper) {
        for (i = 11; i < 22; i++) p->taperNT[i] = 100 + HXO_MIN(150, 20 * (i - 11));
        if (p->vbr_flag) for (i = 11; i < 22; i++) p->taperNT[i] = HXO_MIN(p->taperNT[i], p->initialMNR);
        for (i = 0; i < 21; i++) p->taperNT[i] -= 10 * mnrGOLD[i];
    }
    p->initialMNR = s->MNR = s->MNR + MNRbias - (h_id ? 0 : 300);      /* bitallo3.cpp:446-453 */
    for (i = 0; i < 22; i++) if (p->nBand_l[i] != 0) p->rnBand_l[i] = (1.0f / p->nBand_l[i]);
    s->hf_quant = 0;
    s->gsf_hf = -1;
    s->gsf_hf_stereo[0] = s->gsf_hf_stereo[1] = -1;
    hxo_short_init(e);
    if (p->alloc1) hxo_a1_init(e);

    /* stream state (mp3enc.cpp:278-287,788-837) */
    for (i = 0; i < 32; i++) s->attack_buf[0][i] = s->attack_buf[1][i] = 9000;
    for (i = 0; i < 36; i++)
        s->sig_mask[0][i].sig = s->sig_mask[0][i].mask = s->sig_mask[1][i].sig = s->sig_mask[1][i].mask = 100.0f;
    s->padcount = p->divisor;

    /* echoed control (mp3enc.cpp:839-866) */
    p->ec = ec;
    p->ec.mode = p->h_mode;
    p->ec.bitrate = p->totbitrate;
    if (p->h_mode != 3) p->ec.bitrate /= 2;
    p->ec.samprate = p->samprate;
    p->ec.nsbstereo = is_flag ? nsbstereo : 32;
    p->ec.freq_limit = ec.hf_flag ? ec.freq_limit : p->nsb_limit * (p->samprate / 64);
    p->ec.nsb_limit = p->nsb_limit;
    p->ec.layer = 3;
    return p->nchan * 4 * 1152;
}

/* filter2.c:80-153: shift history, store newest first, optional one-pole DC blocker */
static void input_filter(hxo_encoder *e, const float *pcm)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    float *x = s->buf[0] + 1152, *y = s->buf[1] + 1152, t;
    int i;
    memmove(x, s->buf[0], 2192 * sizeof(float));
    if (p->nchan == 1) {        /* filter2.c:96-121, one channel */
        if (!p->filter_dc) { for (i = 0; i < 1152; i++) *--x = pcm[i]; }
        else {
            float alpha = p->filter_alpha, d = s->dc[0];
            for (i = 0; i < 1152; i++) { t = (float) (pcm[i] - d); d = d + alpha * t; *--x = t; }
            s->dc[0] = d;
        }
        return;
    }
    memmove(y, s->buf[1], 2192 * sizeof(float));
    if (!p->filter_dc) {
        for (i = 0; i < 2304; i += 2) { *--x = pcm[i]; *--y = pcm[i + 1]; }
    } else {
        float alpha = p->filter_alpha, d = s->dc[0], d2 = s->dc[1];
        for (i = 0; i < 2304; i += 2) {
            t = (float) (pcm[i] - d); d = d + alpha * t; *--x = t;
            t = pcm[i + 1] - d2; d2 = d2 + alpha * t; *--y = t;
        }
        s->dc[0] = d; s->dc[1] = d2;
    }
}

/* mp3enc.cpp:1398-1440 + table :82-87 */
static void blocktype_select(hxo_encoder *e, int igr)
{
    static const int bt_sel[4][2][2] = {{{0, 1}, {2, 2}}, {{3, 2}, {2, 2}}, {{3, 2}, {2, 2}}, {{0, 1}, {2, 2}}};
    hxo_state *s = &e->s;
    int prev_gr = igr ^ 1, ahead = (s->igrx + 1) & 3, v1, v2, sf = 0;
    /* the _MPEG2 selectors (mp3enc.cpp:1363-1395,1444-1488) differ only in the detector; the mono ones only use v1 */
    int (*detect)(const float *, int *, int) = e->p.h_id ? hxo_attack_detect : hxo_attack_detect_lsf;
    v1 = detect(s->sample[0][ahead], s->attack_buf[0], s->short_flag_next[prev_gr]);
    v2 = (e->p.nchan == 2) ? detect(s->sample[1][ahead], s->attack_buf[1], s->short_flag_next[prev_gr]) : 0;
    s->last_attack[igr][0] = v1; s->last_attack[igr][1] = v2;
    if (v1 > e->p.short_block_threshold) sf = 1;
    if (v2 > e->p.short_block_threshold) sf = 1;
    s->short_flag_next[igr] = sf;
    s->short_flag_current[igr] = s->short_flag_next[prev_gr];
    s->block_type_prev[igr] = s->block_type[prev_gr];
    s->block_type[igr] = bt_sel[s->block_type_prev[igr]][s->short_flag_current[igr]][s->short_flag_next[igr]];
}

/* mp3enc.cpp:1045-1114 */
static void transform_granule(hxo_encoder *e, int igr)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    int ch, prev = (s->igrx - 1) & 3, ahead = (s->igrx + 2) & 3, bt = s->block_type[igr];
    for (ch = 0; ch < p->nchan; ch++) {
        hxo_freq_invert(s->sample[ch][s->igrx], p->nsb_limitMS[0]);
        if (bt != 2) {
            hxo_hybrid_long(p, s->sample[ch][prev], s->sample[ch][s->igrx], s->xr[igr][ch], bt,
                            p->nsb_limitMS[0], s->xr_clear_flag[igr][ch]);
            s->xr_clear_flag[igr][ch] = 0;
            hxo_antialias(p, s->xr[igr][ch], p->nsb_limitMS[0]);
        } else {
            hxo_hybrid_short(p, s->sample[ch][prev], s->sample[ch][s->igrx], s->xr[igr][ch], p->nsb_limitMS[0]);
            s->xr_clear_flag[igr][ch] = 1;
        }
        /* audio_buf[ch][0] = buf + 576, [ch][1] = buf (mp3enc.cpp:193-196) */
        hxo_polyphase_granule(p, s->buf[ch] + (igr == 0 ? 576 : 0), s->sample[ch][ahead]);
    }
    s->igrx = (s->igrx + 1) & 3;
}

/* mp3enc.cpp:1492-1597 */
static int encode_joint(hxo_encoder *e, hxo_bitw *w)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    int ch, igr, bits, bit_pool, bit_min, bit_max, ba_bit_min, ba_bit_max, ba_min, ba_max;
    int TargetBits, sf_bits, ms = 0, dba_max, shortblock_frame;

    TargetBits = p->AveTargetBits + p->AveTargetBits;
    bit_pool = s->byte_pool << 2;
    bit_max = s->byte_max << 2;
    bit_min = s->byte_min * (1 << 2);        /* (may be negative: a multiplication, not a shift) */
    sf_bits = p->sf_bit_max + p->sf_bit_max;
    ba_bit_max = bit_max - sf_bits;
    ba_bit_min = bit_min - sf_bits;
    ba_min = ba_bit_min;
    ba_max = ba_bit_max;
    dba_max = bit_pool >> 2;
    ba_max = ba_max + dba_max;

    blocktype_select(e, 0);
    transform_granule(e, 0);
    blocktype_select(e, 1);
    transform_granule(e, 1);
    shortblock_frame = (s->block_type[0] == 2) | (s->block_type[1] == 2);

    if (p->ms_flag) {
        int m1, m2;
        if (s->block_type[0] == 2) { s->ms_correlation_memory = 0; m1 = hxo_ms_metric_short(e, (const float (*)[576]) s->xr[0]); }
        else m1 = hxo_ms_metric_long(e, (const float (*)[576]) s->xr[0]);
        if (s->block_type[1] == 2) { s->ms_correlation_memory = 0; m2 = hxo_ms_metric_short(e, (const float (*)[576]) s->xr[1]); }
        else m2 = hxo_ms_metric_long(e, (const float (*)[576]) s->xr[1]);
        s->last_ms_metric[0] = m1; s->last_ms_metric[1] = m2;
        if ((m1 + m2) >= 0) ms = 1;
    }

    if (e->dbg) {
        e->dbg->ms = ms; e->dbg->ms_metric[0] = s->last_ms_metric[0]; e->dbg->ms_metric[1] = s->last_ms_metric[1];
        e->dbg->byte_pool = s->byte_pool;
        memcpy(e->dbg->xr_pre, s->xr, sizeof(s->xr));
        for (igr = 0; igr < 2; igr++) {
            e->dbg->block_type[igr] = s->block_type[igr];
            for (ch = 0; ch < 2; ch++) {
                e->dbg->attack[igr][ch] = s->last_attack[igr][ch];
                /* subband granule computed in this call for (igr, ch): slot written by transform_granule */
                memcpy(e->dbg->sample_new[igr][ch], s->sample[ch][(s->igrx + igr) & 3], 576 * sizeof(float));
            }
        }
    }
    for (igr = 0; igr < 2; igr++) {
        int bt = s->block_type[igr];
        for (ch = 0; ch < 2; ch++) {
            if (e->dbg) { hxo_tap_etab = e->dbg->etab[igr][ch]; hxo_tap_thr = e->dbg->thr[igr][ch]; }
            if (bt != 2) hxo_psy_long(p, s->xr[igr][ch], s->ecsave[ch], s->sig_mask[ch], bt);
            else hxo_psy_short(p, s->xr[igr][ch], s->ecsave[ch], s->sig_mask[ch], s->block_type_prev[igr]);
            hxo_tap_etab = hxo_tap_thr = 0;
            if (e->dbg) { int i; for (i = 0; i < 22; i++) e->dbg->mask[igr][ch][i] = s->sig_mask[ch][i].mask; }
        }
        s->gr[igr][0].block_type = s->gr[igr][1].block_type = bt;
        hxo_bitallo_long(e, s->xr[igr], s->sig_mask, ba_min, TargetBits, ba_max, bit_pool,
                         s->sf[igr], s->gr[igr], ms);
        if (e->dbg) {
            memcpy(e->dbg->ix[igr], s->ix, sizeof(s->ix));
            memcpy(e->dbg->signx[igr], s->signx, sizeof(s->signx));
            for (ch = 0; ch < 2; ch++) { int i; for (i = 0; i < 22; i++) e->dbg->sf[igr][ch][i] = s->sf[igr][ch].l[i]; }
        }
        for (ch = 0; ch < 2; ch++) {
            hxo_gr *g = &s->gr[igr][ch];
            bits = 0;
            g->scalefac_compress = 0;
            if (shortblock_frame) {
                s->scfsi[ch] = 0;
                if (g->aux_not_null)
                    g->scalefac_compress = (bt == 2) ? hxo_pack_sf_short(w, &s->sf[igr][ch]) : hxo_pack_sf_long(w, &s->sf[igr][ch]);
            } else {
                g->scalefac_compress = hxo_pack_sf_long_scfsi(w, s->sf_save[ch], &s->sf[igr][ch], igr, &s->scfsi[ch], g->aux_not_null);
            }
            if (g->aux_not_null) bits = hxo_pack_huff(w, g, s->ix[ch], s->signx[ch]);
            ba_min -= bits;
            ba_max -= bits;
            g->part2_3_length = bits;
        }
        ba_min += ba_bit_min + sf_bits;
        ba_max = ba_max - dba_max;
        ba_max += ba_bit_max + sf_bits;
    }
    return ms;
}

/* mp3enc.cpp:1675-1749 encode_singleB: one channel, no M/S */
static int encode_single(hxo_encoder *e, hxo_bitw *w)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    int igr, bits, bit_pool, bit_min, bit_max, ba_bit_min, ba_bit_max, ba_min, ba_max, shortblock_frame;

    bit_pool = s->byte_pool << 2;
    bit_max = s->byte_max << 2;
    bit_min = s->byte_min * (1 << 2);        /* (may be negative: a multiplication, not a shift) */
    ba_bit_max = bit_max;
    if (ba_bit_max > 4095) ba_bit_max = 4095;
    ba_bit_min = bit_min;
    ba_bit_max -= p->sf_bit_max;
    ba_bit_min -= p->sf_bit_max;
    ba_min = ba_bit_min;
    ba_max = ba_bit_max;

    blocktype_select(e, 0);
    transform_granule(e, 0);
    blocktype_select(e, 1);
    transform_granule(e, 1);
    shortblock_frame = (s->block_type[0] == 2) | (s->block_type[1] == 2);
    if (e->dbg) {
        e->dbg->ms = 0; e->dbg->ms_metric[0] = e->dbg->ms_metric[1] = 0;
        e->dbg->byte_pool = s->byte_pool;
        memcpy(e->dbg->xr_pre, s->xr, sizeof(s->xr));
        for (igr = 0; igr < 2; igr++) e->dbg->block_type[igr] = s->block_type[igr];
    }
    for (igr = 0; igr < 2; igr++) {
        int bt = s->block_type[igr];
        hxo_gr *g = &s->gr[igr][0];
        if (bt != 2) hxo_psy_long(p, s->xr[igr][0], s->ecsave[0], s->sig_mask[0], bt);
        else hxo_psy_short(p, s->xr[igr][0], s->ecsave[0], s->sig_mask[0], s->block_type_prev[igr]);
        g->block_type = bt;
        hxo_bitallo_long(e, s->xr[igr], s->sig_mask, ba_min, p->AveTargetBits, ba_max, bit_pool, s->sf[igr], s->gr[igr], 0);
        if (e->dbg) {
            int i;
            memcpy(e->dbg->ix[igr], s->ix, sizeof(s->ix));
            memcpy(e->dbg->signx[igr], s->signx, sizeof(s->signx));
            for (i = 0; i < 22; i++) e->dbg->sf[igr][0][i] = s->sf[igr][0].l[i];
        }
        bits = 0;
        g->scalefac_compress = 0;
        if (shortblock_frame) {
            s->scfsi[0] = 0;
            if (g->aux_not_null)
                g->scalefac_compress = (bt == 2) ? hxo_pack_sf_short(w, &s->sf[igr][0]) : hxo_pack_sf_long(w, &s->sf[igr][0]);
        } else {
            g->scalefac_compress = hxo_pack_sf_long_scfsi(w, s->sf_save[0], &s->sf[igr][0], igr, &s->scfsi[0], g->aux_not_null);
        }
        if (g->aux_bits) bits = hxo_pack_huff(w, g, s->ix[0], s->signx[0]);
        ba_min += ba_bit_min + p->sf_bit_max - bits;
        ba_max += ba_bit_max + p->sf_bit_max - bits;
        g->part2_3_length = bits;
    }
    return 0;
}

/* ---- streams coded by the first-generation allocator: long blocks only, no block-type decision ---- */
static void psy_long_all(hxo_encoder *e, int igr)
{
    int ch;
    for (ch = 0; ch < e->p.nchan; ch++) hxo_psy_long(&e->p, e->s.xr[igr][ch], e->s.ecsave[ch], e->s.sig_mask[ch], 0);
}

/* encode_jointA (mp3enc.cpp:1236-1318): joint stereo with an intensity part, both channels allocated together */
static int encode_joint_a1(hxo_encoder *e, hxo_bitw *w)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    int ch, igr, bits, ba_bit_min, ba_bit_max, ba_min, ba_max, TargetBits, sf_bits, ms = 0;
    TargetBits = p->AveTargetBits + p->AveTargetBits;
    ba_bit_max = s->byte_max << 2;
    if (ba_bit_max > 4095) ba_bit_max = 4095;
    ba_bit_min = s->byte_min * (1 << 2);        /* (may be negative: a multiplication, not a shift) */
    sf_bits = p->sf_bit_max + p->sf_bit_max;
    ba_bit_max -= sf_bits;
    ba_bit_min -= sf_bits;
    ba_min = ba_bit_min;
    ba_max = ba_bit_max;
    transform_granule(e, 0);
    transform_granule(e, 1);
    if (p->ms_flag) {
        int m1 = hxo_a1_ms_metric(e, (const float (*)[576]) s->xr[0]), m2 = hxo_a1_ms_metric(e, (const float (*)[576]) s->xr[1]);
        s->last_ms_metric[0] = m1; s->last_ms_metric[1] = m2;
        if ((m1 + m2) >= 0) ms = 1;
    }
    if (e->dbg) {
        e->dbg->ms = ms; e->dbg->ms_metric[0] = s->last_ms_metric[0]; e->dbg->ms_metric[1] = s->last_ms_metric[1];
        e->dbg->byte_pool = s->byte_pool;
        memcpy(e->dbg->xr_pre, s->xr, sizeof(s->xr));
        e->dbg->block_type[0] = e->dbg->block_type[1] = 0;
    }
    for (igr = 0; igr < 2; igr++) {
        psy_long_all(e, igr);
        hxo_bitallo1(e, s->xr[igr], s->sig_mask, 0, 2, ba_min, TargetBits, ba_max, s->sf[igr], s->gr[igr], s->ix, s->signx, ms);
        if (e->dbg) {
            int i;
            memcpy(e->dbg->ix[igr], s->ix, sizeof(s->ix));
            memcpy(e->dbg->signx[igr], s->signx, sizeof(s->signx));
            for (ch = 0; ch < 2; ch++) for (i = 0; i < 22; i++) e->dbg->sf[igr][ch][i] = s->sf[igr][ch].l[i];
        }
        for (ch = 0; ch < 2; ch++) {
            hxo_gr *g = &s->gr[igr][ch];
            bits = 0;
            g->scalefac_compress = hxo_pack_sf_long_scfsi(w, s->sf_save[ch], &s->sf[igr][ch], igr, &s->scfsi[ch], g->aux_not_null);
            if (g->aux_not_null) bits = hxo_pack_huff(w, g, s->ix[ch], s->signx[ch]);
            ba_min -= bits;
            ba_max -= bits;
            g->part2_3_length = bits;
        }
        ba_min += ba_bit_min + sf_bits;
        ba_max += ba_bit_max + sf_bits;
    }
    return ms;
}

/* encode_singleA (mp3enc.cpp:1601-1672): dual channel, each channel allocated on its own */
static int encode_single_a1(hxo_encoder *e, hxo_bitw *w)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    int ch, igr, bits, ba_bit_min, ba_bit_max, ba_min, ba_max;
    ba_bit_max = s->byte_max << 1;          /* two channels share the frame */
    ba_bit_min = s->byte_min * (1 << 1);        /* (may be negative: a multiplication, not a shift) */
    if (ba_bit_max > 4095) ba_bit_max = 4095;
    ba_bit_max -= p->sf_bit_max;
    ba_bit_min -= p->sf_bit_max;
    ba_min = ba_bit_min;
    ba_max = ba_bit_max;
    transform_granule(e, 0);
    transform_granule(e, 1);
    if (e->dbg) {
        e->dbg->ms = 0; e->dbg->ms_metric[0] = e->dbg->ms_metric[1] = 0;
        e->dbg->byte_pool = s->byte_pool;
        memcpy(e->dbg->xr_pre, s->xr, sizeof(s->xr));
        e->dbg->block_type[0] = e->dbg->block_type[1] = 0;
    }
    for (igr = 0; igr < 2; igr++) {
        psy_long_all(e, igr);
        for (ch = 0; ch < 2; ch++) {
            hxo_gr *g = &s->gr[igr][ch];
            hxo_bitallo1(e, s->xr[igr] + ch, s->sig_mask + ch, ch, 1, ba_min, p->AveTargetBits, ba_max, &s->sf[igr][ch], g,
                         s->ix + ch, s->signx + ch, p->ms_flag);
            bits = 0;
            g->scalefac_compress = 0;
            if (g->aux_bits) {
                g->scalefac_compress = hxo_pack_sf_long(w, &s->sf[igr][ch]);
                bits = hxo_pack_huff(w, g, s->ix[ch], s->signx[ch]);
            }
            ba_min += ba_bit_min + p->sf_bit_max - bits;
            ba_max += ba_bit_max + p->sf_bit_max - bits;
            g->part2_3_length = bits;
        }
        if (e->dbg) {
            int i;
            memcpy(e->dbg->ix[igr], s->ix, sizeof(s->ix));
            memcpy(e->dbg->signx[igr], s->signx, sizeof(s->signx));
            for (ch = 0; ch < 2; ch++) for (i = 0; i < 22; i++) e->dbg->sf[igr][ch][i] = s->sf[igr][ch].l[i];
        }
    }
    s->scfsi[0] = s->scfsi[1] = 0;
    return 0;
}

/* encode_jointA_MPEG2 (mp3enc.cpp:1753-1829) and encode_singleA_MPEG2 (:1910-1973): one granule = one frame */
static int encode_granule_lsf_a1(hxo_encoder *e, hxo_bitw *w, int igr)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    int ch, bits, bit_min, bit_max, ba_min, ba_max, ms = 0;
    if (p->h_mode == 2) {
        bit_max = s->byte_max << 2;
        bit_min = s->byte_min * (1 << 2);        /* (may be negative: a multiplication, not a shift) */
        ba_max = HXO_MIN(bit_max, 4095) - p->sf_bit_max;
        ba_min = bit_min - p->sf_bit_max;
        transform_granule(e, igr);
        if (e->dbg) { e->dbg->ms = 0; e->dbg->byte_pool = s->byte_pool; e->dbg->block_type[igr] = 0; memcpy(e->dbg->xr_pre[igr], s->xr[igr], sizeof(s->xr[igr])); }
        psy_long_all(e, igr);
        for (ch = 0; ch < 2; ch++) {
            hxo_gr *g = &s->gr[igr][ch];
            const int ba_bit_max = HXO_MIN(bit_max, 4095) - p->sf_bit_max, ba_bit_min = bit_min - p->sf_bit_max;
            hxo_bitallo1(e, s->xr[igr] + ch, s->sig_mask + ch, ch, 1, ba_min, p->AveTargetBits, ba_max, &s->sf[igr][ch], g,
                         s->ix + ch, s->signx + ch, p->ms_flag);
            bits = 0;
            g->scalefac_compress = 0;
            if (g->aux_bits) {
                g->scalefac_compress = hxo_pack_sf_lsf(w, &s->sf[igr][ch], 0);
                bits = hxo_pack_huff(w, g, s->ix[ch], s->signx[ch]);
            }
            ba_min += ba_bit_min + p->sf_bit_max - bits;
            ba_max += ba_bit_max + p->sf_bit_max - bits;
            g->part2_3_length = bits;
        }
        if (e->dbg) { memcpy(e->dbg->ix[igr], s->ix, sizeof(s->ix)); memcpy(e->dbg->signx[igr], s->signx, sizeof(s->signx)); }
        return 0;
    }
    bit_max = s->byte_max << 3;
    bit_min = s->byte_min * (1 << 3);        /* (may be negative: a multiplication, not a shift) */
    if (s->byte_pool > 245) bit_min += 40;
    ba_max = HXO_MIN(bit_max, 4095) - 2 * p->sf_bit_max;
    ba_min = bit_min - 2 * p->sf_bit_max;
    transform_granule(e, igr);
    if (p->ms_flag) {
        int m1 = hxo_a1_ms_metric(e, (const float (*)[576]) s->xr[igr]);
        s->last_ms_metric[igr] = m1;
        if (m1 >= 0) ms = 1;
    }
    if (e->dbg) { e->dbg->ms = ms; e->dbg->byte_pool = s->byte_pool; e->dbg->block_type[igr] = 0; memcpy(e->dbg->xr_pre[igr], s->xr[igr], sizeof(s->xr[igr])); }
    psy_long_all(e, igr);
    hxo_bitallo1(e, s->xr[igr], s->sig_mask, 0, 2, ba_min, p->AveTargetBits + p->AveTargetBits, ba_max, s->sf[igr], s->gr[igr],
                 s->ix, s->signx, ms);
    if (e->dbg) { memcpy(e->dbg->ix[igr], s->ix, sizeof(s->ix)); memcpy(e->dbg->signx[igr], s->signx, sizeof(s->signx)); }
    for (ch = 0; ch < 2; ch++) {
        hxo_gr *g = &s->gr[igr][ch];
        bits = 0;
        g->scalefac_compress = 0;
        if (g->aux_not_null) {
            if (ch & p->is_flag) g->scalefac_compress = hxo_pack_sf_lsf_is(w, &s->sf[igr][ch], e->a1t.nsf[1]);
            else g->scalefac_compress = hxo_pack_sf_lsf(w, &s->sf[igr][ch], 0);
            bits = hxo_pack_huff(w, g, s->ix[ch], s->signx[ch]);
        }
        g->part2_3_length = bits;
    }
    return ms;
}

/* encode_jointB_MPEG2 (mp3enc.cpp:1832-1907) and encode_singleB_MPEG2 (:1977-2027): one granule = one frame */
static int encode_granule_lsf(hxo_encoder *e, hxo_bitw *w, int igr)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    int ch, bits, bit_pool, bit_min, bit_max, ba_min, ba_max, sf_bits, ms = 0, bt;

    bit_pool = s->byte_pool << 3;
    bit_max = s->byte_max << 3;
    bit_min = s->byte_min * (1 << 3);        /* (may be negative: a multiplication, not a shift) */
    if (p->nchan == 2 && s->byte_pool > 245) bit_min += 40;
    ba_max = bit_max;
    if (ba_max > 4095) ba_max = 4095;
    sf_bits = p->nchan * p->sf_bit_max;
    ba_max -= sf_bits;
    ba_min = bit_min - sf_bits;

    blocktype_select(e, igr);
    transform_granule(e, igr);
    bt = s->block_type[igr];
    if (p->ms_flag) {
        int m1;
        if (bt == 2) { s->ms_correlation_memory = 0; m1 = hxo_ms_metric_short(e, (const float (*)[576]) s->xr[igr]); }
        else m1 = hxo_ms_metric_long(e, (const float (*)[576]) s->xr[igr]);
        s->last_ms_metric[igr] = m1;
        if (m1 >= 0) ms = 1;
    }
    if (e->dbg) {
        e->dbg->ms = ms; e->dbg->byte_pool = s->byte_pool;
        e->dbg->block_type[igr] = bt;
        memcpy(e->dbg->xr_pre[igr], s->xr[igr], sizeof(s->xr[igr]));
    }
    for (ch = 0; ch < p->nchan; ch++) {
        if (bt != 2) hxo_psy_long(p, s->xr[igr][ch], s->ecsave[ch], s->sig_mask[ch], bt);
        else hxo_psy_short(p, s->xr[igr][ch], s->ecsave[ch], s->sig_mask[ch], s->block_type_prev[igr]);
    }
    s->gr[igr][0].block_type = s->gr[igr][1].block_type = bt;
    hxo_bitallo_long(e, s->xr[igr], s->sig_mask, ba_min, p->nchan * p->AveTargetBits, ba_max, bit_pool,
                     s->sf[igr], s->gr[igr], ms);
    if (e->dbg) {
        memcpy(e->dbg->ix[igr], s->ix, sizeof(s->ix));
        memcpy(e->dbg->signx[igr], s->signx, sizeof(s->signx));
    }
    for (ch = 0; ch < p->nchan; ch++) {
        hxo_gr *g = &s->gr[igr][ch];
        bits = 0;
        g->scalefac_compress = 0;
        if (p->nchan == 2 ? g->aux_not_null : g->aux_bits) {
            g->scalefac_compress = hxo_pack_sf_lsf(w, &s->sf[igr][ch], bt);
            bits = hxo_pack_huff(w, g, s->ix[ch], s->signx[ch]);
        }
        g->part2_3_length = bits;
    }
    return ms;
}

/* L3_audio_encode_MPEG2 (mp3enc.cpp:2483-2593) and L3_audio_encode_vbr_MPEG2 (:2337-2479): the 1152-sample block
   already shifted in by input_filter becomes two single-granule frames */
static int encode_block_lsf(hxo_encoder *e, unsigned char *out)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    hxo_bitw w;
    unsigned char *out0 = out, *pk = e->packet;
    int igr, bytes, raw_bytes, pad, ms, ibr, mf, bytesout;

    for (igr = 0; igr < 2; igr++) {
        pad = 0; ibr = 0;
        if (!p->vbr_flag) {
            s->padcount -= p->remainder;
            if (s->padcount <= 0) { s->padcount += p->divisor; pad = 1; }
            s->frame_mf_bytes[s->side_p1] = p->main_framebytes + pad;
        }
        s->frame_main_pos[s->side_p1] = s->main_tot;
        s->byte_pool = (int) (s->mf_tot - s->main_tot);
        if (!p->vbr_flag) {
            s->byte_max = p->main_framebytes + pad + s->byte_pool;
            s->byte_min = s->byte_max - 255;
        } else {
            s->byte_max = p->vbr_main_framebytes[p->ivbr_max] + s->byte_pool;
            s->byte_min = p->vbr_main_framebytes[p->ivbr_min] + s->byte_pool - 255;
        }
        hxo_bw_init(&w, s->main_buf + s->main_p1);
        ms = p->alloc1 ? encode_granule_lsf_a1(e, &w, igr) : encode_granule_lsf(e, &w, igr);
        s->last_ms = ms;
        s->mode_ext_buf[s->side_p1] = (unsigned char) (ms + ms + p->is_flag);
        bytes = hxo_bw_flush(&w);
        assert(bytes <= s->byte_max);
        if (e->dbg) {
            memcpy(e->dbg->gr[igr], s->gr[igr], sizeof(s->gr[igr]));
            e->dbg->MNR_after = s->MNR;
            e->dbg->main_bytes = bytes;
        }
        if (p->vbr_flag) {
            int bytes2 = bytes - s->byte_pool, bytes3 = bytes2 + p->vbr_pool_target;
            int side_dp = (s->side_p1 - s->side_p0) & 31;
            for (ibr = p->ivbr_min; ibr <= p->ivbr_max; ibr++) if (bytes2 <= p->vbr_main_framebytes[ibr]) break;
            if (side_dp < 10) {
                for (; ibr <= p->ivbr_max; ibr++) if (bytes3 < p->vbr_main_framebytes[ibr + 1]) break;
            } else if (side_dp > 15) {      /* many frames span the pool: drain it through byte_min padding */
                if (side_dp > 24) s->byte_min = p->vbr_main_framebytes[p->ivbr_min] + s->byte_pool;
                else s->byte_min = p->vbr_main_framebytes[p->ivbr_min] + (s->byte_pool >> 4);
            }
            if (ibr > p->ivbr_max) ibr = p->ivbr_max;
            s->br_index_buf[s->side_p1] = (unsigned char) ibr;
            s->frame_mf_bytes[s->side_p1] = p->vbr_main_framebytes[ibr];
        }
        raw_bytes = bytes;
        if (bytes < s->byte_min) {
            memset(s->main_buf + s->main_p1 + bytes, 0, s->byte_min - bytes);
            bytes = s->byte_min;
        }
        hxo_pack_side_lsf(s->side_buf[s->side_p1], p->h_mode, s->gr[igr], p->nchan);
        if (pk) {       /* mp3enc.cpp:3352-3362, :3196-3206: the two packets back to back */
            pk[0] = p->head[0]; pk[1] = p->head[1]; pk[2] = p->head[2]; pk[3] = p->head[3];
            if (pad) pk[2] |= 2;
            pk[3] = (unsigned char) ((pk[3] & 0xCF) | ((ms + ms + p->is_flag) << 4));
            memcpy(pk + 4, s->side_buf[s->side_p1], p->side_bytes);
            memcpy(pk + 4 + p->side_bytes, s->main_buf + s->main_p1, (size_t) raw_bytes);
            e->packet_bytes2[igr] = 4 + p->side_bytes + raw_bytes;
            pk += e->packet_bytes2[igr];
        }
        s->main_tot += bytes;
        s->main_bytes += bytes;
        s->main_p1 += bytes;
        s->mf_tot += p->vbr_flag ? p->vbr_main_framebytes[ibr] : p->main_framebytes + pad;
        s->side_p1 = (s->side_p1 + 1) & 31;

        while (s->side_p0 != s->side_p1) {
            int main_data_begin;
            mf = s->frame_mf_bytes[s->side_p0];
            if (s->main_bytes < mf) break;
            s->tot_frames_out++;
            main_data_begin = (int) (s->main_sent - s->frame_main_pos[s->side_p0]);
            assert(main_data_begin >= 0 && main_data_begin < 256);
            s->main_sent += mf;
            out[0] = p->head[0]; out[1] = p->head[1]; out[2] = p->head[2]; out[3] = p->head[3];
            if (!p->vbr_flag) { if (mf - p->main_framebytes) out[2] |= 2; }
            else out[2] = (unsigned char) ((out[2] & 0x0F) | (s->br_index_buf[s->side_p0] << 4));
            out[3] = (unsigned char) ((out[3] & 0xCF) | (s->mode_ext_buf[s->side_p0] << 4));
            out += 4;
            s->side_buf[s->side_p0][0] = (unsigned char) main_data_begin;
            memmove(out, s->side_buf[s->side_p0], p->side_bytes);
            out += p->side_bytes;
            memmove(out, s->main_buf + s->main_p0, mf);
            out += mf;
            s->main_bytes -= mf;
            s->main_p0 += mf;
            s->side_p0 = (s->side_p0 + 1) & 31;
        }
        if (s->main_p1 > 16384) {
            s->main_p1 = s->main_p1 - s->main_p0;
            memmove(s->main_buf, s->main_buf + s->main_p0, s->main_p1);
            s->main_p0 = 0;
        }
    }
    e->packet_bytes = e->packet ? e->packet_bytes2[0] + e->packet_bytes2[1] : 0;
    bytesout = (int) (out - out0);
    s->tot_bytes_out += bytesout;
    s->ave_tot_bytes_out = s->ave_tot_bytes_out + ((((bytesout << 8) - s->ave_tot_bytes_out)) >> 6);
    return bytesout;
}

/* mp3enc.cpp:2230-2333 (CBR) and :2106-2226 (VBR) */
int hxo_encode_frame(hxo_encoder *e, const float *pcm, unsigned char *out)
{
    hxo_state *s = &e->s;
    const hxo_params *p = &e->p;
    hxo_bitw w;
    unsigned char *out0 = out;
    int bytes, raw_bytes, pad = 0, ms, ibr = 0, mf, bytesout;

    input_filter(e, pcm);
    if (!p->h_id) return encode_block_lsf(e, out);
    if (!p->vbr_flag) {
        s->padcount -= p->remainder;
        if (s->padcount <= 0) { s->padcount += p->divisor; pad = 1; }
        s->frame_mf_bytes[s->side_p1] = p->main_framebytes + pad;
    }
    s->frame_main_pos[s->side_p1] = s->main_tot;
    s->byte_pool = (int) (s->mf_tot - s->main_tot);
    if (!p->vbr_flag) {
        s->byte_max = p->main_framebytes + pad + s->byte_pool;
        s->byte_min = s->byte_max - 511;
    } else {
        s->byte_max = p->vbr_main_framebytes[p->ivbr_max] + s->byte_pool;
        s->byte_min = p->vbr_main_framebytes[p->ivbr_min] + s->byte_pool - 511;
    }
    hxo_bw_init(&w, s->main_buf + s->main_p1);
    if (p->alloc1) ms = (p->h_mode == 2) ? encode_single_a1(e, &w) : encode_joint_a1(e, &w);
    else ms = (p->nchan == 2) ? encode_joint(e, &w) : encode_single(e, &w);
    s->last_ms = ms;
    s->mode_ext_buf[s->side_p1] = (unsigned char) (ms + ms + p->is_flag);
    bytes = hxo_bw_flush(&w);
    assert(bytes <= s->byte_max);
    if (e->dbg) {
        memcpy(e->dbg->gr, s->gr, sizeof(s->gr));
        e->dbg->scfsi[0] = s->scfsi[0]; e->dbg->scfsi[1] = s->scfsi[1];
        e->dbg->MNR_after = s->MNR;
        e->dbg->main_bytes = bytes;
    }
    if (p->vbr_flag) {
        int bytes2 = bytes - s->byte_pool, bytes3 = bytes2 + p->vbr_pool_target;
        for (ibr = p->ivbr_min; ibr <= p->ivbr_max; ibr++) if (bytes2 <= p->vbr_main_framebytes[ibr]) break;
        for (; ibr <= p->ivbr_max; ibr++) if (bytes3 < p->vbr_main_framebytes[ibr + 1]) break;
        if (ibr > p->ivbr_max) ibr = p->ivbr_max;
        s->br_index_buf[s->side_p1] = (unsigned char) ibr;
        s->frame_mf_bytes[s->side_p1] = p->vbr_main_framebytes[ibr];
    }
    raw_bytes = bytes;
    if (bytes < s->byte_min) {
        memset(s->main_buf + s->main_p1 + bytes, 0, s->byte_min - bytes);
        bytes = s->byte_min;
    }
    hxo_pack_side(s->side_buf[s->side_p1], p->h_mode, s->scfsi, s->gr, p->nchan);
    if (e->packet) {    /* reformatted frame of the *_Packet entry points (mp3enc.cpp:3066-3074, :2944-2951):
                           this frame's header, side info (main_data_begin still 0) and unpadded main data */
        unsigned char *q = e->packet;
        q[0] = p->head[0]; q[1] = p->head[1]; q[2] = p->head[2]; q[3] = p->head[3];
        if (pad) q[2] |= 2;     /* L3_pack_head in both variants: the VBR packet keeps the nominal bitrate index */
        q[3] = (unsigned char) ((q[3] & 0xCF) | ((ms + ms + p->is_flag) << 4));
        memcpy(q + 4, s->side_buf[s->side_p1], p->side_bytes);
        memcpy(q + 4 + p->side_bytes, s->main_buf + s->main_p1, (size_t) raw_bytes);
        e->packet_bytes = 4 + p->side_bytes + raw_bytes;
    }
    s->main_tot += bytes;
    s->main_bytes += bytes;
    s->main_p1 += bytes;
    s->mf_tot += p->vbr_flag ? p->vbr_main_framebytes[ibr] : p->main_framebytes + pad;
    s->side_p1 = (s->side_p1 + 1) & 31;

    while (s->side_p0 != s->side_p1) {
        int main_data_begin;
        mf = s->frame_mf_bytes[s->side_p0];
        if (s->main_bytes < mf) break;
        s->tot_frames_out++;
        main_data_begin = (int) (s->main_sent - s->frame_main_pos[s->side_p0]);
        s->main_sent += mf;
        out[0] = p->head[0]; out[1] = p->head[1]; out[2] = p->head[2]; out[3] = p->head[3];
        if (!p->vbr_flag) { if (mf - p->main_framebytes) out[2] |= 2; }
        else out[2] = (unsigned char) ((out[2] & 0x0F) | (s->br_index_buf[s->side_p0] << 4));
        out[3] = (unsigned char) ((out[3] & 0xCF) | (s->mode_ext_buf[s->side_p0] << 4));
        out += 4;
        s->side_buf[s->side_p0][0] = (unsigned char) (main_data_begin >> 1);
        s->side_buf[s->side_p0][1] |= (main_data_begin & 1) << 7;
        memmove(out, s->side_buf[s->side_p0], p->side_bytes);
        out += p->side_bytes;
        memmove(out, s->main_buf + s->main_p0, mf);
        out += mf;
        s->main_bytes -= mf;
        s->main_p0 += mf;
        s->side_p0 = (s->side_p0 + 1) & 31;
    }
    bytesout = (int) (out - out0);
    s->tot_bytes_out += bytesout;
    s->ave_tot_bytes_out = s->ave_tot_bytes_out + ((((bytesout << 8) - s->ave_tot_bytes_out)) >> 7);
    if (s->main_p1 > 16384) {
        s->main_p1 = s->main_p1 - s->main_p0;
        memmove(s->main_buf, s->main_buf + s->main_p0, s->main_p1);
        s->main_p0 = 0;
    }
    return bytesout;
}

/* CMp3Enc::L3_audio_encode_Packet: also returns the frame as a self-contained packet */
int hxo_encode_frame_packet(hxo_encoder *e, const float *pcm, unsigned char *out, unsigned char *packet, int nbytes[2])
{
    int n;
    e->packet = packet;
    e->packet_bytes = e->packet_bytes2[0] = e->packet_bytes2[1] = 0;
    n = hxo_encode_frame(e, pcm, out);
    e->packet = 0;
    if (e->p.h_id) { nbytes[0] = e->packet_bytes; nbytes[1] = 0; }     /* mp3enc.cpp:3068-3069 */
    else { nbytes[0] = e->packet_bytes2[0]; nbytes[1] = e->packet_bytes2[1]; }     /* mp3enc.cpp:3363: two packets */
    return n;
}

int hxo_encode_frame_s16(hxo_encoder *e, const int16_t *pcm, unsigned char *out)
{
    float f[2304];
    int i, n = 1152 * e->p.nchan;
    for (i = 0; i < n; i++) f[i] = (float) pcm[i];
    return hxo_encode_frame(e, f, out);
}

/* test accessor: copy a named init-time table of an initialised encoder (returns bytes, -1 = unknown) */
long long hxo_debug_table(const hxo_encoder *e, const char *name, void *dst, long long cap)
{
    const hxo_params *p = &e->p;
    const void *src = 0;
    long long n = 0;
    static int v[16];
#define TAB(nm, obj) if (!strcmp(name, nm)) { src = &(obj); n = sizeof(obj); }
    TAB("psy_w", p->psyL.w) TAB("psy_cnt", p->psyL.cnt) TAB("psy_off", p->psyL.off) TAB("psy_nsum", p->psyL.nsum)
    TAB("psy_npart", p->psyL.npart) TAB("dct_coef", p->dct_coef) TAB("win", p->win) TAB("csa", p->csa)
    TAB("m18_w", p->m18_w) TAB("m18_w2", p->m18_w2) TAB("m18_c", p->m18_c)
    TAB("m6_v", p->m6_v) TAB("m6_v2", p->m6_v2) TAB("m6_c87", p->m6_c87)
    TAB("look_gain", p->look_gain) TAB("look_34igain", p->look_34igain) TAB("look_ix43", p->look_ix43)
    TAB("look_log_cbwmb", p->look_log_cbwmb) TAB("nBand_l", p->nBand_l) TAB("startBand_l", p->startBand_l)
    TAB("nsf", p->nsf) TAB("taperNT", p->taperNT) TAB("head", p->head) TAB("ec", p->ec)
#undef TAB
    if (!strcmp(name, "scalars")) {
        v[0] = p->nsb_limit; v[1] = p->nsb_limitMS[0]; v[2] = p->band_limit; v[3] = p->main_framebytes; v[4] = p->AveTargetBits;
        v[5] = p->initialMNR; v[6] = p->ms_flag; v[7] = p->hf_flag; v[8] = p->vbr_flag; v[9] = p->framebytes;
        v[10] = p->remainder; v[11] = p->ivbr_max; v[12] = p->vbr_pool_target; v[13] = p->samprate; v[14] = p->totbitrate; v[15] = p->nsb_limitMS[1];
        src = v; n = sizeof(v);
    }
    if (!src) return -1;
    if (n > cap) n = cap;
    memcpy(dst, src, (size_t) n);
    return n;
}
