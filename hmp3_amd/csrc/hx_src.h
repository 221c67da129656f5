// hx_src.h - host-side sample-format / sample-rate converter in front of the encoder (hx_src.cpp)
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
typedef struct hx_src hx_src;
hx_src *hx_src_create(void);
void hx_src_destroy(hx_src *s);
/* Csrc::sr_convert_init (reference srcc.cpp:730): bytes the caller must hold per convert call, 0 = unsupported */
int hx_src_init(hx_src *s, int source, int channels, int bits, int is_float, int target, int target_channels,
                int *encode_cutoff_freq);
/* Csrc::sr_convert (reference srcc.cpp:795): 1152 samples per output channel into yout (fp32 at int16 scale);
   returns the input bytes consumed.  Reads up to 1152 * (source / target + 1) sample frames from xin. */
int hx_src_convert(hx_src *s, const unsigned char *xin, float *yout, int *out_bytes);
#ifdef __cplusplus
}
#endif
