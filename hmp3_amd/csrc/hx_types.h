// hx_types.h - POD layouts shared by the host runtime and the HIP kernels of the
// MI355X batched MP3 encoder.  Domain names follow the reference (granule, sfb, part2_3...).
#pragma once
#include <stdint.h>

// Same field order as the reference's E_CONTROL (pub/encapp.h:42-72): passed verbatim
// through the C-ABI.
struct HxControl {
    int mode, bitrate, samprate, nsbstereo, filter_select, freq_limit, nsb_limit;
    int layer, cr_bit, original, hf_flag, vbr_flag, vbr_mnr, vbr_br_limit, vbr_delta_mnr;
    int chan_add_f0, chan_add_f1, sparse_scale;
    int mnr_adjust[21];
    int cpu_select, quick, test1, test2, test3, short_block_threshold;
};

// MPEG_HEAD (pub/encapp.h:141-157)
struct HxMpegHead {
    int sync, id, option, prot, br_index, sr_index, pad, private_bit, mode, mode_ext, cr, original, emphasis;
};

// Psychoacoustic tables of one block class (long = 2 partitions per sfb)
struct HxPsyTab {
    int npart;          // spreading rows
    int npart_e;        // energy partitions
    int nsum[64];       // lines per partition
    int pstart[65];     // first line of partition
    int cnt[64], off[64], row[64];  // spreading row: length, first source partition, offset into w
    float w[2200];      // [0..63] absolute threshold x width, rows from 128
};

// Everything resolved at init for one configuration class (sample rate / bitrate / flags).
// Lives in device global memory; kernels index it through HxStream::cls.
struct HxParams {
    HxControl ec;           // echoed control (L3_audio_encode_info_ec)
    HxMpegHead head_info;
    unsigned char head[4];
    int totbitrate, samprate, sr_index, h_mode;
    int h_id, tix;          // 1 = MPEG-1, 0 = MPEG-2 LSF (one granule per frame); band-table row
    int nband, nsb, nsb_limit, nsb_ms0, nsb_ms1, band_limit, band_limit_stereo;
    int framebytes, remainder, divisor, main_framebytes, side_bytes, sf_bit_max, AveTargetBits;
    int ms_flag, hf_flag, vbr_flag, short_block_threshold, filter_dc;
    int nchan;              // 1 = mono (mode 3), 2 = stereo / joint stereo
    float filter_alpha;
    int ivbr_min, ivbr_max, vbr_main_framebytes[16], vbr_framebytes[16], vbr_pool_target;
    int initialMNR, test1, taperNT[22];
    int nBand_l_iso[22];
    int nBand_l[22], startBand_l[24], nBand_s[13], startBand_s[14];
    int nsf[2], nsf2[2], nsf3[2], nbmax[2], nbmax2[2], nbmax3[2];
    int look_log_cbwmb[22];
    float rnBand_l[22];
    unsigned char band_of_line[576];    // sfb index of each MDCT line (allocator's band table)
    // Blocked line runs of the stream walk's certified band sums (hx_alloc.hip, noise_sweep / inverse_sf2; hx_host.cpp band_runs):
    // lane l owns up to run_w consecutive lines of ONE band among the bands the gain search measures, the lanes of a band are
    // neighbours (at most 16).  lane_run[l] = first line >> 1 | (lines >> 1) << 9 | (lanes between l and its band's first lane) << 12;
    // band_last_lane[b] = the last lane of band b's run of lanes.
    int run_w;
    unsigned short lane_run[64];
    unsigned char band_last_lane[24];
    // 32-point analysis DCT: twiddles 2 cos(pi (2j+1) / 2N) of the size-N step at [N/2 + j] (heap order; [0] unused)
    float dct_tw[32];
    float win[4][36], csa[2][8];
    // N-point MDCT kernels (N = 18 long, 6 short): input twiddles, twiddles of the odd half, and the rows of the
    // N/2-point cosine transform that are not plain sums (9-point: outputs 2, 4, 8 / 1, 5, 7 / the factor of output 3)
    float mdct_pre18[18], mdct_odd18[9], dct9_even[3][4], dct9_odd[3][4], dct9_k3;
    float mdct_pre6[6], mdct_odd6[3], dct3_k;
    HxPsyTab psyL;
    HxPsyTab psyS;                      // short blocks: rows from w[0]
    float look_gain[128], look_34igain[128], look_ix43[256];
    // short-block allocator (reference bitallos.cpp:128-200)
    int nsfs, nbmax_s, look_log_cbwmb_s[16];
    unsigned char sband_of_line[192];   // short sfb index of each line of a 192-line window
    // streams the reference codes with its first-generation allocator (CBitAllo1, bitallo1.cpp): joint stereo with an
    // intensity part (is_flag) and dual channel.  Long blocks only; tables of bitallo1.cpp:107-211,444-543
    int is_flag, alloc1, a1_ill_is_pos;
    float a1_log_cbw[21];                       // 10 log10(band width)
    float a1_f_ix[256], a1_f_ixmax[256], a1_f_big_ix[256], a1_f_big_ixmax[256];    // noise estimators by quantised value
    int a1_bits[256], a1_is_pos[34];            // bit estimator x 16; intensity position by energy ratio
    float a1_gz0, a1_gz1, a1_gz2, a1_c707, a1_sparse[21];
};

// Class-independent tables.
#define HX_POW43_N 16384             // quantised values covered by the double-precision x^(4/3) table
struct HxGlobalTabs {
    double pow43[HX_POW43_N];           // i^(4/3) as the reference's pow() call returns it (noise of lines quantised beyond the 256-entry float table)
    float anwin[512];
    float anwin_r[512];                 // the same window in the order K1 uses it: [k][j][A tap, B tap]
    int mblog[256];
    float mbexp_lo[256], mbexp_hi[256];
    float pow34_exp[256], pow34_a[16], pow34_b[16];
    float quant_off[32];
    int logsub[84];
    unsigned short huff_code[1408];     // 1378 used (ISO Table B.7 at true dimensions)
    unsigned char huff_len[1408];
    unsigned short huff_off[32];
    unsigned char huff_dim[32], huff_lin[32];
    unsigned char quada_code[16], quada_len[16];
};

// Side information of one granule/channel (pub/l3e.h:72-96)
struct HxGr {
    int part2_3_length, big_values, global_gain, scalefac_compress;
    int window_switching_flag, block_type, mixed_block_flag;
    int table_select[3], subblock_gain[3];
    int region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    int aux_nquads, aux_bits, aux_not_null, aux_nreg[3];
};

#define HX_MAINBUF (16384 + 512 + 1440 + 256)

// Persistent per-stream state (device global memory).  It is the checkpoint unit: copying
// this struct plus the subband carry is a complete snapshot of a stream.
struct HxStream {
    int cls;                    // index into the HxParams array
    int frames_in;              // frames submitted so far
    // front end
    float dc[2];
    float pcm_hist[2][480];     // last 480 filtered samples per channel (oldest first)
    int attack_hist[2][32];     // energy history in mB (detect.c)
    int bt_prev;                // block type of the previous granule
    int short_flag_next_prev;   // short_flag_next of the previous granule
    float thr_prev[2][64];      // previous granule's unclamped thresholds x 2 (ecsave)
    // allocator
    int MNR, PoolFraction, call_count, ms_memory, NTadjust[2][22];
    // first-generation allocator (pub/bitallo1.h:80-140): gain steps carried per channel, bit-estimate feedback,
    // running noise-to-mask level
    int a1_gsf[2][21], a1_bitadjust[2], a1_call_count;
    float a1_running_a, a1_ave_alpha, a1_alpha;
    int sf_save[2][21];
    int gr_subblock_gain[2][2][3];
    // reservoir / frame assembly (mp3enc.cpp:2230-2333)
    int padcount;
    unsigned main_tot, main_sent, mf_tot;
    int main_bytes, main_p0, main_p1;
    unsigned side_p0, side_p1;
    unsigned frame_main_pos[32];
    int frame_mf_bytes[32];
    unsigned char mode_ext_buf[32], br_index_buf[32], side_buf[32][32];
    unsigned tot_frames_out, tot_bytes_out;
    int ave_tot_bytes_out;
    unsigned char main_buf[HX_MAINBUF];
};

// What the allocator needs at the start of a long-block granule and that does not depend on its carried state:
// written by k_prep per (stream, granule), read by k_alloc.  Energies are those of the input channels (L / R);
// x34max / gzero refer to the representation the frame is coded in (L / R, or M / S in a joint-stereo frame).
struct alignas(16) HxBandPrep {
    float xsxx[2][22];          // band energies of L and R
    float x34max[2][22];        // largest |x|^(3/4) of the band
    int n0[2][22];              // band energy per line in mB: L, R
    int n0ms[2][22];            // the same of M, S (joint-stereo frames only)
    int gzero[2][22];           // gain step at which the whole band quantises to zero
    int maskmb[2][22];          // masking threshold in mB after pre-echo control
};

// Optional per-frame debug taps written by the allocator kernel (tests only).
struct HxFrameDebug {
    int ms, ms_metric[2], byte_pool, MNR_after;
    int mask_mb[2][2][22];      // [gr][ch] mbLog of the per-sfb mask
    HxGr gr[2][2];
    int sf[2][2][22];
    int scfsi[2];
    int main_bytes;             // bytes of main data produced by this frame (before padding)
};

// ---- k_alloc -> k_pack hand-over: what the packer needs of a coded frame ----
// One (granule, channel): where its bits start in the frame's main data, the scalefactor fields in transmission
// order, the Huffman regions.  The quantised lines travel as int16 beside it.
struct alignas(16) HxSegOut {
    int start_bit;              // first bit of the segment within the frame's main data
    int huff_bits;              // Huffman bits counted by the allocator (the packer checks them)
    unsigned nreg01;            // pairs in region 0 | pairs in region 1 << 16
    unsigned nreg2_quads;       // pairs in region 2 | count1 quads << 16
    unsigned tabs;              // Huffman table of region 0 | region 1 << 8 | region 2 << 16 | count1 table << 24
    int not_null;               // 0 = the segment carries no bits at all
    unsigned short sf[40];      // (length << 8) | value of each transmitted scalefactor field
};
// One frame: its main data spans the pending slots from first_slot on (the first one has main_bytes bytes in use)
struct HxSlot { int off, mf; };     // a frame's slot in a stream's output: offset of its header, main-data bytes it holds
struct alignas(16) HxFrameOut {
    int bytes, raw_bytes;       // main data bytes with / without the zero stuffing up to byte_min
    int first_slot, main_bytes;
    long long packet_off;       // offset of the frame's main data in the packet buffer, -1 = no packet
    int pad_[2];
    HxSlot near[4];             // copies of slots first_slot .. first_slot + 3: the packer rarely needs more, and gets them in the record's round trip
};
#define HX_SLOTS_EXTRA 40           // slots pending from earlier calls (the ring holds 32)

// k_polyphase (hx_front.hip) and its launch (hx_cabi.hip): granules per workgroup (252 of 256 lanes busy: a lane is a time slot
// of a granule) and the workgroup size that follows.  One definition: the kernel's launch bounds, its LDS staging stride and
// register array are sized by the same numbers the host launches with.
// (7 granules: 126 of the workgroup's 128 lanes have a time slot, 37 KB of LDS, four workgroups per CU; with 14 - 252 of 256 lanes,
// 70 KB, two per CU - the loads of one workgroup's tile overlapped less of the other's arithmetic: 7 is +1.6 % per step at configs 2
// and 3; 3 and 5 granules are level with it, 10 in between)
#ifndef K1_GPB
#define K1_GPB 7
#endif
#define K1_THREADS ((K1_GPB * 18 + 63) / 64 * 64)

// The lines' signs travel from k_prep (or, for short-block and first-generation-allocator granules, the stream walk) to
// k_pack as one bit per line: 576 bits = 18 words per (granule, channel), padded to 20 so that a granule's 160 bytes keep
// 16-byte alignment.  (As one byte per line they were 0.6 GB written and 0.6 GB read per config-2 step.)
#define HX_SGN_WORDS 20

// Arguments of the allocator kernels (k_alloc / k_alloc_lsf), filled by the host runtime.
struct AllocArgs {
    HxStream *st;
    const HxParams *prm;
    const HxGlobalTabs *gt;
    const float *xr;            // [S][NG][2][576]
    const float *etab, *thr;    // [S][NG][2][64]
    const int *msbase;          // [S][NG]
    const unsigned char *bt;    // [S][NG]
    const unsigned char *btprev;    // [S] block type of the granule before this call
    unsigned char *out;         // [S][out_stride]
    int *out_bytes;             // [S]
    HxFrameDebug *dbg;          // [S][F] or null
    long long out_stride;
    int NG, S;
    int *status;
    unsigned long long *prof;
    unsigned char *packet;      // optional [S][F][packet_stride]: each frame as a self-contained packet
    long long packet_stride;
    int *packet_bytes;          // [S][F]
    int *frame_stats;           // optional [S][F][2]: frames / bytes emitted by the stream after each input frame
    int *pre_len, *carry_len;   // [S] bytes of pending frames' images at the call's start (k_pack_pre copies them in) / at its end (k_pack_carry saves them)
    const int *order;           // workgroup -> stream (longest-running first, from the previous call's durations), or null = identity
    unsigned *dur;              // [S] this call's duration of each stream's workgroup, 100 MHz ticks
    int *done_counter;          // [0] streams retired, [2] streams started by all launches so far (k_gate of a pipelined submit waits on the latter), [3] double-table line passes, [4] certified band sums that fell back to the strict sum,
                                // [5] positions of the launch order claimed so far in this launch, [6] workgroups of this launch that ran out of work (the last one zeroes both), [8 ..] parking
    int park_k;                 // > 0: the workgroups that share a CU with one of the first park_k workgroups of the launch order (the streams that ran
                                // longest in the previous call) keep their slot until that one retires (hx_alloc3.inc, "parking"); done_counter[8 ..] holds the CU ids
    int strict_sums;            // 1 = no certified band sums: every band is added in line order (HMP3AMD_EXACT_SUMS=1; tests)
    // from k_msscan / k_prep (hx_front.hip); xr holds the coded magnitudes for long-block granules
    const float *x34;           // [S][NG][2][576] x^(3/4) of the magnitudes (long-block granules)
    const unsigned *sgn;        // [S][NG][2][HX_SGN_WORDS] sign of each line, one bit per line in line order (bit j & 31 of word j >> 5)
    const HxBandPrep *band;     // [S][NG]
    const unsigned char *msflag;    // [S][NG] 1 = the granule's frame is coded M/S
    const int *msdec;           // [S][NG] the stereo metric after hysteresis (debug taps)
    const float *thrprev;       // [S][2][64] pre-echo memory the call started with
    // to k_pack (hx_pack.hip)
    short *ixq;                 // [S][NG][2][576] quantised magnitudes
    unsigned *sgn_w;            // the sign buffer again, writable: short-block granules store their reordered signs
    HxSegOut *seg;              // [S][NG][2]
    HxFrameOut *frm;            // [S][frames per call]
    HxSlot *slots;              // [S][frames per call + HX_SLOTS_EXTRA]
};

