/* hxo_int.h - ORACLE internals (test infrastructure). */
#ifndef HXO_INT_H
#define HXO_INT_H
#include "hxo.h"

#define HXO_MAX(a, b) ((a) > (b) ? (a) : (b))
#define HXO_MIN(a, b) ((a) < (b) ? (a) : (b))

float hxo_bits2f(uint32_t u);
uint32_t hxo_f2bits(float f);
void hxo_math_init(void);
int hxo_round(float x);
int hxo_logsubber(int n1, int n2);
float hxo_dblog(float x);
float hxo_anwin(int n);
extern float hxo_quant_off[32];

int hxo_huff_dim(int t);
int hxo_huff_linbits(int t);
int hxo_huff_code(int t, int x, int y);
int hxo_huff_len(int t, int x, int y);
int hxo_quada_code(int v);
int hxo_quada_len(int v);

int hxo_sfb_long_edge(int sr_index, int i);
int hxo_sfb_short_edge(int sr_index, int i);
int hxo_sfbl_limit(int sr_index, int band_limit);
int hxo_sfbs_limit(int sr_index, int band_limit);
int hxo_nearest_sf_band_freq(int tix, int samprate, int freq);
int hxo_samprate_mpeg1(int sr_index);
void hxo_init_transform_tables(hxo_params *p);
void hxo_init_psy_long(hxo_params *p);
void hxo_init_psy_short(hxo_params *p);
void hxo_count_init(void);

/* short-block allocator (hxo_short.c) */
int hxo_ms_metric_short(hxo_encoder *e, const float x[2][576]);
int hxo_bitallo_short(hxo_encoder *e, float xr[2][576], hxo_sigmask sm[2][36],
                      int min_bits, int target_bits, int max_bits, int bit_pool,
                      hxo_scalefact sf_out[2], hxo_gr gr[2], int ms_flag, int MNR);
void hxo_short_init(hxo_encoder *e);

/* first-generation allocator (hxo_alloc1.c) */
void hxo_a1_init(hxo_encoder *e);
int hxo_a1_ms_metric(hxo_encoder *e, const float x[2][576]);
void hxo_bitallo1(hxo_encoder *e, float xr[][576], hxo_sigmask sm[][36], int ch_arg, int nchan_arg,
                  int min_bits, int target_bits, int max_bits, hxo_scalefact sf_out[], hxo_gr gr[],
                  int ix[][576], unsigned char signx[][576], int ms_flag);
int hxo_pack_sf_lsf_is(hxo_bitw *w, hxo_scalefact *sf, int nsf_stereo);
#endif
