/* mp3enc.h - drop-in for the reference's hmp3/src/pub/mp3enc.h (class CMp3Enc, pub/mp3enc.h:74-139) on top
 * of libhmp3amd.so.  A client that includes "mp3enc.h" and links the Helix objects puts this directory ahead of
 * the reference's pub/ on its include path and links -lhmp3amd instead; nothing else changes.  encapp.h (the
 * E_CONTROL / MPEG_HEAD / IN_OUT / INT_PAIR PODs) stays the client's own: the HX_* structs are layout-identical.
 * `make -C oracle shimcli` builds the reference's own command line (test/tomp3.cpp, unmodified) this way. */
#ifndef HMP3_AMD_SHIM_MP3ENC_H
#define HMP3_AMD_SHIM_MP3ENC_H
#include <stdlib.h>
#include <stdio.h>
#include <float.h>
#include <math.h>
#include <string.h>
#include <assert.h>
#include "hmp3_amd.h"
#include "encapp.h"     /* the client's: E_CONTROL, MPEG_HEAD, IN_OUT */
#include "hxtypes.h"    /* the client's: min / max (the reference header pulls it in, its callers rely on that) */

typedef struct
{
    int a;
    int b;
}
INT_PAIR;               /* pub/mp3enc.h:66-71 */

class CMp3Enc
{
  public:
    CMp3Enc() : h(hx_enc_create(0)) {}
    ~CMp3Enc() { hx_enc_destroy(h); }
    int L3_audio_encode_init(E_CONTROL *ec) { return hx_enc_L3_audio_encode_init(h, (const HX_E_CONTROL *) ec); }
    IN_OUT L3_audio_encode(float *pcm, unsigned char *bs_out) { return io(hx_enc_L3_audio_encode(h, pcm, bs_out)); }
    IN_OUT L3_audio_encode_Packet(float *pcm, unsigned char *bs_out, unsigned char *packet, int nbytes_out[2])
    { return io(hx_enc_L3_audio_encode_Packet(h, pcm, bs_out, packet, nbytes_out)); }
    int MP3_audio_encode_init(E_CONTROL *ec, int input_type = 16, int is_float = 0, int mpeg_select = 0, int mono_convert = 0)
    { return hx_enc_MP3_audio_encode_init(h, (const HX_E_CONTROL *) ec, input_type, is_float, mpeg_select, mono_convert); }
    IN_OUT MP3_audio_encode(unsigned char *pcm, unsigned char *bs_out) { return io(hx_enc_MP3_audio_encode(h, pcm, bs_out)); }
    IN_OUT MP3_audio_encode_Packet(unsigned char *pcm, unsigned char *bs_out, unsigned char *packet, int nbytes_out[2])
    { return io(hx_enc_MP3_audio_encode_Packet(h, pcm, bs_out, packet, nbytes_out)); }
    int L3_audio_encode_get_bitrate() { return hx_enc_get_bitrate(h); }
    float L3_audio_encode_get_bitrate_float() { return hx_enc_get_bitrate_float(h); }
    float L3_audio_encode_get_bitrate2_float() { return hx_enc_get_bitrate2_float(h); }
    unsigned int L3_audio_encode_get_frames() { return hx_enc_get_frames(h); }
    INT_PAIR L3_audio_encode_get_frames_bytes() { HX_INT_PAIR p = hx_enc_get_frames_bytes(h); INT_PAIR r; r.a = p.a; r.b = p.b; return r; }
    void L3_audio_encode_info_ec(E_CONTROL *ec) { hx_enc_info_ec(h, (HX_E_CONTROL *) ec); }
    void L3_audio_encode_info_head(MPEG_HEAD *head) { hx_enc_info_head(h, (HX_MPEG_HEAD *) head); }
    void L3_audio_encode_info_string(char *s) { hx_enc_info_string(h, s); }
    void out_stats() { hx_enc_out_stats(h); }           /* pub/mp3enc.h:141, test routine */

  private:
    static IN_OUT io(HX_IN_OUT x) { IN_OUT r; r.in_bytes = x.in_bytes; r.out_bytes = x.out_bytes; return r; }
    hx_enc *h;
};
#endif
