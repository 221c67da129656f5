/*
 * hmp3_amd.h - C ABI of the MI355X-native batched MP3 (MPEG-1 Layer III) encoder.
 *
 * Drop-in boundary for the Helix encoder's frame-encode path: every hx_enc_* entry point
 * replaces one public method of the reference's `class CMp3Enc`
 * (/root/reference/hmp3/src/pub/mp3enc.h:74-139), with the same argument meaning, the same
 * return values and the same error behaviour (init returns 0 on failure; encode cannot fail).
 * E_CONTROL / MPEG_HEAD / IN_OUT / INT_PAIR are the reference's plain structs
 * (pub/encapp.h:42-72,141-165; pub/mp3enc.h:66-71) and cross this ABI unchanged.
 *
 * The hx_batch_* entry points have no reference equivalent: they encode N independent streams
 * x F frames per call on one GPU, which is where the throughput comes from.  All buffers are
 * plain pointers; "device" variants take HIP device pointers and a hipStream_t (as void*).
 *
 * Everything runs on the GPU: there is no CPU fallback.  hx_batch_create / the hx_enc init calls
 * check that the device is a gfx950 (MI355X); if none is usable they return NULL / 0 and
 * hx_last_error() says why.  Every kernel launch is checked where it is made.
 */
#ifndef HMP3_AMD_H
#define HMP3_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pub/encapp.h:42-72 */
typedef struct {
    int mode, bitrate, samprate, nsbstereo, filter_select, freq_limit, nsb_limit;
    int layer, cr_bit, original, hf_flag, vbr_flag, vbr_mnr, vbr_br_limit, vbr_delta_mnr;
    int chan_add_f0, chan_add_f1, sparse_scale;
    int mnr_adjust[21];
    int cpu_select, quick, test1, test2, test3, short_block_threshold;
} HX_E_CONTROL;

/* pub/encapp.h:141-157 */
typedef struct {
    int sync, id, option, prot, br_index, sr_index, pad, private_bit, mode, mode_ext, cr, original, emphasis;
} HX_MPEG_HEAD;

typedef struct { int in_bytes, out_bytes; } HX_IN_OUT;      /* pub/encapp.h:160-165 */
typedef struct { int a, b; } HX_INT_PAIR;                   /* pub/mp3enc.h:66-71 */

typedef struct hx_enc hx_enc;       /* one stream, CMp3Enc-compatible */
typedef struct hx_batch hx_batch;   /* N streams on one GPU */

const char *hx_last_error(void);
int hx_device_count(void);
void hx_default_control(HX_E_CONTROL *ec);                  /* CLI defaults, test/tomp3.cpp:357-384 */

/* ---- CMp3Enc replacement (one stream, host buffers) ---- */
hx_enc *hx_enc_create(int device);                          /* CMp3Enc::CMp3Enc,  mp3enc.cpp:115 */
void hx_enc_destroy(hx_enc *e);                             /* CMp3Enc::~CMp3Enc, mp3enc.cpp:201 */
/* CMp3Enc::L3_audio_encode_init (mp3enc.cpp:220): returns bytes of float PCM per call (9216) or 0 */
int hx_enc_L3_audio_encode_init(hx_enc *e, const HX_E_CONTROL *ec);
/* CMp3Enc::L3_audio_encode (mp3enc.cpp:2031): 1152 x 2 floats at int16 scale, oldest first.
   From the third call on a call is one HIP-graph launch: the encoder's single-stream chain (PCM up from page-locked staging,
   the pipeline's kernels, whose last workgroup writes byte count / frame counter / bitstream to page-locked host memory and
   publishes a sequence word behind them) is recorded once and replayed; the bytes per call are those of the plain calls.
   The *_Packet calls take the plain path; the environment variable HMP3AMD_ENC_GRAPH=0 restores it for every call (2: replay,
   but wait with hipStreamSynchronize instead of polling the sequence word). */
HX_IN_OUT hx_enc_L3_audio_encode(hx_enc *e, const float *pcm, unsigned char *bs_out);
/* CMp3Enc::MP3_audio_encode_init (mp3enc.cpp:2655): 8/16/24/32-bit PCM or 32-bit float source at
   8 - 48 kHz, converted to the encode rate by the built-in converter; returns min input bytes per call or 0 */
int hx_enc_MP3_audio_encode_init(hx_enc *e, const HX_E_CONTROL *ec, int source_bits, int source_is_float,
                                 int mpeg_select, int mono_convert);
/* CMp3Enc::MP3_audio_encode (mp3enc.cpp:2812) */
HX_IN_OUT hx_enc_MP3_audio_encode(hx_enc *e, const unsigned char *pcm, unsigned char *bs_out);
/* (init returns the bytes a call needs buffered - 1153 sample frames when no rate conversion is involved -
   and in_bytes of a call is what the converter consumed; mpeg_select: 0 track the input rate, 1 an MPEG-1
   rate, 2 an MPEG-2 rate, else the encode rate in Hz) */
int hx_enc_get_bitrate(hx_enc *e);                          /* mp3enc.cpp:3444 */
float hx_enc_get_bitrate_float(hx_enc *e);                  /* mp3enc.cpp:3451 */
float hx_enc_get_bitrate2_float(hx_enc *e);                 /* mp3enc.cpp:3468 */
unsigned hx_enc_get_frames(hx_enc *e);                      /* mp3enc.cpp:3484 */
HX_INT_PAIR hx_enc_get_frames_bytes(hx_enc *e);             /* mp3enc.cpp:3512 */
void hx_enc_info_ec(hx_enc *e, HX_E_CONTROL *ec);           /* mp3enc.cpp:3491 */
void hx_enc_info_head(hx_enc *e, HX_MPEG_HEAD *head);       /* mp3enc.cpp:3498 */
/* pub/mp3enc.h:110-131 *_Packet: also return this call's frame as a self-contained ("reformatted")
   packet: header, side info with main_data_begin 0, unpadded main data; nbytes_out[0] = its size,
   nbytes_out[1] = 0.  At the MPEG-2 rates (16 / 22.05 / 24 kHz) a call yields two single-granule
   frames (mp3enc.cpp:3301-3440): two packets back to back, nbytes_out[0] then nbytes_out[1] bytes.
   bs_out or packet may be NULL. */
HX_IN_OUT hx_enc_L3_audio_encode_Packet(hx_enc *e, const float *pcm, unsigned char *bs_out, unsigned char *packet, int nbytes_out[2]);
HX_IN_OUT hx_enc_MP3_audio_encode_Packet(hx_enc *e, const unsigned char *pcm, unsigned char *bs_out, unsigned char *packet, int nbytes_out[2]);
void hx_enc_info_string(hx_enc *e, char *s);                /* mp3enc.cpp:3505, <= 80 chars */
/* CMp3Enc::out_stats (mp3enc.cpp:3522, "test routine"): the allocator's counters to stderr - the long-block
   allocator's call count; the reference's remaining columns are uninitialised diagnostics and print as 0 */
void hx_enc_out_stats(hx_enc *e);

/* ---- sample-format / sample-rate converter (host side) ----
   What CMp3Enc::MP3_audio_encode runs in front of every frame: Csrc (pub/srcc.h:87-97).  hx_enc_MP3_audio_encode
   uses it internally; it is exported for callers that feed the batched API from sources at other rates. */
typedef struct hx_src hx_src;
hx_src *hx_src_create(void);
void hx_src_destroy(hx_src *s);
/* Csrc::sr_convert_init (srcc.cpp:730): bytes to hold per convert call, 0 = unsupported pair */
int hx_src_init(hx_src *s, int source, int channels, int bits, int is_float, int target, int target_channels,
                int *encode_cutoff_freq);
/* Csrc::sr_convert (srcc.cpp:795): 1152 samples per output channel, fp32 at int16 scale; returns input bytes used */
int hx_src_convert(hx_src *s, const unsigned char *xin, float *yout, int *out_bytes);

/* ---- batched encode (N independent streams) ---- */
/* ec: nstreams controls, or one shared control when shared_control != 0.
   max_frames: largest nframes any later call will pass.  Returns NULL on failure. */
hx_batch *hx_batch_create(int device, int nstreams, const HX_E_CONTROL *ec, int shared_control, int max_frames);
void hx_batch_destroy(hx_batch *b);
int hx_batch_nstreams(const hx_batch *b);
/* start a new stream in slot i with the slot's configuration (the state CMp3Enc::L3_audio_encode_init leaves,
   mp3enc.cpp:278-287, 788-837); waits for work in flight, leaves the other streams alone */
int hx_batch_reset_stream(hx_batch *b, int i);
/* checkpoint / resume of one stream (header + encoder state + subband carry, hx_batch_stream_state_bytes bytes):
   what is saved from slot i continues, after hx_batch_set_stream_state, in any slot of any batch created with
   the same control for that slot (any size, any max_frames) - another GPU or a later process included.  src
   must hold hx_batch_stream_state_bytes bytes; a blob of another library build or saved under another
   control is refused (-1, hx_last_error). */
long long hx_batch_stream_state_bytes(const hx_batch *b);
int hx_batch_get_stream_state(hx_batch *b, int i, void *dst);
int hx_batch_set_stream_state(hx_batch *b, int i, const void *src);
/* worst-case bytes one stream can emit in a call of nframes frames */
long long hx_batch_out_stride(const hx_batch *b, int nframes);
/* PCM: int16 interleaved L/R, [nstreams][nframes*1152][2]; out: [nstreams][out_stride] bytes;
   out_bytes: [nstreams] bytes produced (whole frames; the frames emitted are exactly those the
   reference emits over the same nframes calls).  stream: hipStream_t or NULL.
   Asynchronous on `stream`.  Returns 0, or a negative error. */
int hx_batch_encode_s16_device(hx_batch *b, const int16_t *d_pcm, int nframes, unsigned char *d_out,
                               long long out_stride, int *d_out_bytes, void *stream);
/* same with host buffers (staged through the device, synchronous) */
int hx_batch_encode_s16_host(hx_batch *b, const int16_t *pcm, int nframes, unsigned char *out,
                             long long out_stride, int *out_bytes);
/* the same for fp32 PCM at int16 scale (+-32768), the form CMp3Enc::L3_audio_encode takes
   (pub/mp3enc.h:90-98; a1 of the path: srcc.cpp:805-808 scales [-1,1) floats by 32768 first) */
int hx_batch_encode_f32_device(hx_batch *b, const float *d_pcm, int nframes, unsigned char *d_out,
                               long long out_stride, int *d_out_bytes, void *stream);
int hx_batch_encode_f32_host(hx_batch *b, const float *pcm, int nframes, unsigned char *out,
                             long long out_stride, int *out_bytes);
/* Pipelined form of the device calls (no reference equivalent: the reference encodes one frame per
   call on the CPU).  A submit is asynchronous like the plain call, but its outputs are ordered on
   `stream` only by a later hx_batch_wait (or by the next plain / host-buffer call on the batch); the
   PCM must be ready on `stream` at the submit and stay unchanged until hx_batch_wait.  Consecutive
   submits overlap: the front-end kernels of call n+1 and the bit packing of call n-1 run on the SIMDs
   that the allocator kernel of call n leaves idle while its slowest streams finish.  Hand consecutive
   submits different d_out / d_out_bytes (two sets in turn): with overlapping ones the result is the
   same, but the allocator launch waits for the previous call's packing. */
int hx_batch_submit_s16_device(hx_batch *b, const int16_t *d_pcm, int nframes, unsigned char *d_out,
                               long long out_stride, int *d_out_bytes, void *stream);
int hx_batch_submit_f32_device(hx_batch *b, const float *d_pcm, int nframes, unsigned char *d_out,
                               long long out_stride, int *d_out_bytes, void *stream);
int hx_batch_wait(hx_batch *b, void *stream);
/* a submit's front end is held back until the previous call's allocator kernel occupies its share of the chip:
   until this percentage (default 90) of the workgroups the device can hold at once have started.  It then
   queues for the slots that finishing streams free and runs in that kernel's tail; 0 releases it at once. */
void hx_batch_set_gate(hx_batch *b, int percent);
/* The same pipelining for host buffers: the PCM of call n+1 crosses PCIe while call n is encoded and the
   bitstream of call n while call n+1 is.  pcm must stay unchanged, and out / out_bytes are undefined, until
   hx_batch_wait_host returns.  Use page-locked memory (hx_pinned_alloc) for copies that really overlap. */
int hx_batch_submit_s16_host(hx_batch *b, const int16_t *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes);
int hx_batch_submit_f32_host(hx_batch *b, const float *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes);
int hx_batch_wait_host(hx_batch *b);
void *hx_pinned_alloc(long long bytes);
void hx_pinned_free(void *p);
/* optional packet outputs of the batched calls: d_packet [nstreams][nframes][frame_stride] bytes,
   d_packet_bytes [nstreams][nframes][2] (the reference's nbytes_out[2] of every call: {size, 0},
   or the sizes of the two back-to-back packets of an MPEG-2 call); frame_stride >= the packet
   bytes of one call (4096 is always enough).  NULL switches them off.  Applies to the calls
   that follow. */
void hx_batch_packet_buffers(hx_batch *b, unsigned char *d_packet, long long frame_stride, int *d_packet_bytes);
/* optional per-frame counters of the batched calls: d_stats [nstreams][nframes][2] = the stream's
   get_frames() / bytes emitted so far after each input frame, i.e. what a caller of the per-frame
   API (CMp3Enc::L3_audio_encode_get_frames_bytes after every call) would have seen.  NULL = off. */
void hx_batch_frame_stats_buffer(hx_batch *b, int *d_stats);
/* fp32 host call that also returns those counters to a host array stats[nstreams][nframes][2] */
int hx_batch_encode_f32_host_stats(hx_batch *b, const float *pcm, int nframes, unsigned char *out,
                                   long long out_stride, int *out_bytes, int *stats);
/* the settings an encoder would run a control with (what L3_audio_encode_info_ec / _info_head report,
   mp3enc.cpp:839-866), host only; 0 = configuration rejected */
int hx_control_info(const HX_E_CONTROL *ec, HX_E_CONTROL *ec_out, HX_MPEG_HEAD *head_out);
/* status bits accumulated by the kernels: 2 = main data overflow (the reference would assert
   there), 4 = the Huffman bits packed for a channel differ from the bits counted for it (an internal
   consistency check of the two-wave packer).  0 = healthy; -1 = no answer (the batch became unusable after a
   failed device call, or the status could not be read).  Synchronises - and, like hx_batch_wait, first enqueues the
   packing that the last hx_batch_submit_*_device left for later: that writes the submit's output buffers, which must
   therefore still be valid.
   hx_batch_gate_timeouts: how many pipelined submits started their front end late because the gate on the previous
   allocator launch gave up waiting (results are correct, overlap was lost; a loaded or profiled GPU can cause it).
   It is a performance counter, not part of the health status. */
int hx_batch_status(hx_batch *b);
int hx_batch_gate_timeouts(hx_batch *b);
/* Which build of the per-stream rate-loop kernel the batch runs (chosen at create from the batch size; the environment
   variable HMP3AMD_K6 = fat | slim overrides): 0 = k_alloc, four streams per CU, 1 = k_alloc_slim, six per CU.  Both
   restate the same reference code (bitallo3.cpp:484-3149) and produce the same bytes.  hx_batch_resident_streams: how many
   of the batch's streams the device holds at once with that kernel. */
int hx_batch_k6_variant(const hx_batch *b);
int hx_batch_resident_streams(const hx_batch *b);
/* identifies the build: a hash of the library's sources and code-generation flags (hmp3_amd/build.sh) */
const char *hx_build_id(void);
/* total frames / bytes emitted so far by stream i (synchronises) */
HX_INT_PAIR hx_batch_frames_bytes(hx_batch *b, int stream_index);
/* mean device time of the dominant (allocator) kernel over the calls since the last query, in
   milliseconds, measured with HIP events on the launch stream; also returns the call count */
float hx_batch_alloc_kernel_ms(hx_batch *b, int *ncalls);

/* ---- host placement (no reference equivalent): a host-fed GPU reads ~50 GB/s of PCM over PCIe, so its page-locked
   buffers and the threads that submit its copies belong on the NUMA node the device hangs on.
   hx_device_numa_node: that node from sysfs (-1 = unknown or not a NUMA machine).  hx_bind_thread_to_device: restricts the
   calling thread to the CPUs of that node which the process may use and returns their number (0 = nothing changed); call it
   before hx_pinned_alloc so that first touch places the pages there.  hx_multi_* binds its per-device threads itself.
   "The CPUs the process may use" are the ones its loading thread had when the library was loaded: a thread bound to one
   device's node can be bound to another device's node afterwards.  hx_bind_thread_to_node: the same by node number. */
int hx_device_numa_node(int device);
int hx_bind_thread_to_device(int device);
int hx_bind_thread_to_node(int node);
/* The process's CPU set is captured when the library is loaded.  A process that is narrowed or moved afterwards (taskset -p,
   os.sched_setaffinity by a launcher, a cpuset change) calls this from any thread to take the set again from its main
   thread's current mask; returns the number of CPUs (0 = failed, nothing changed).  The bind calls retry with it once by
   themselves when the kernel rejects the mask they computed from the stale set. */
int hx_refresh_process_cpus(void);

/* ---- several GPUs of one node behind one handle (no reference equivalent) ----
   nstreams independent streams in contiguous blocks over ndev devices (devices[0..ndev), or devices
   0..ndev-1 when devices is NULL; ndev <= 0: every device present), block sizes differing by at most one;
   one hx_batch per device, one host thread per device and call, nothing exchanged between devices.
   Buffers as in the hx_batch_*_host calls, covering all nstreams streams; out_stride >= hx_multi_out_stride. */
typedef struct hx_multi hx_multi;
hx_multi *hx_multi_create(int ndev, const int *devices, int nstreams, const HX_E_CONTROL *ec, int shared_control, int max_frames);
void hx_multi_destroy(hx_multi *m);
int hx_multi_ndevices(const hx_multi *m);
int hx_multi_nstreams(const hx_multi *m);
int hx_multi_shard(const hx_multi *m, int k, int *device, int *first, int *count);   /* block k: its device and streams */
hx_batch *hx_multi_batch(hx_multi *m, int k);                                        /* block k's batch, for the per-batch calls */
long long hx_multi_out_stride(const hx_multi *m, int nframes);
int hx_multi_encode_s16_host(hx_multi *m, const int16_t *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes);
int hx_multi_encode_f32_host(hx_multi *m, const float *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes);
int hx_multi_encode_f32_host_stats(hx_multi *m, const float *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes, int *stats);
int hx_multi_status(hx_multi *m);

/* ---- test taps (tests only; synchronise) ---- */
/* name: "sb" "xr" "etab" "thr" "msbase" "bt" "eng" "dbg" (per-stage buffers), "ixq" "sgn" "seg" "frm" (what the allocator hands the
   packer), "dur" (per-stream duration of the last allocator launch, 100 MHz ticks), "place" (where position i of that launch's
   order ran: XCC id << 16 | HW_ID[15:0], i.e. CU [11:8], SH [12], SE [15:13]), "state", and two device counters (one int
   each, running over the batch's calls): "big_sweeps" = noise passes that took the double-precision x^(4/3) table,
   "strict_sums" = certified band sums that fell back to the reference's strict line-order sum (DESIGN.md section 2;
   HMP3AMD_EXACT_SUMS=1 sends every sum there).  Copies at most cap bytes, returns bytes, -1 for an unknown name. */
long long hx_batch_debug_read(hx_batch *b, const char *name, void *dst, long long cap);
void hx_batch_debug_enable(hx_batch *b, int on);
/* The first-generation allocator (intensity stereo, dual channel; reference bitallo1.cpp) calls libm's logf / log10f, which
   are not correctly rounded: what the reference encodes depends on the C library it is linked with.  The kernels restate
   glibc 2.35's (hmp3_amd/csrc/hx_libm32.h).  hx_libc_version: the host's C library; hx_libm_spot_check: how many of n
   sample arguments the host's logf / log10f treat differently (0: a reference built on this host and the kernels agree). */
const char *hx_libc_version(void);
int hx_libm_spot_check(int n);
/* 1 if the host tables of this control have the structure the low-footprint kernel derives them from (gain tables and
   the x^(3/4) exponent table as ldexp of 4 / 16 constants, mB tables within 16 bits); hx_batch_create checks the same
   before it picks that kernel.  -1 = configuration rejected.  Host only. */
int hx_debug_slim_tables_ok(const HX_E_CONTROL *ec);
/* host-side table generation for the CPU tests (no GPU): see hx_cabi.hip */
long long hx_debug_host_table(const HX_E_CONTROL *ec, const char *name, void *dst, long long cap);

/* ---- Xing / Info / LAME tag frame (first frame of a file; reference pub/xhead.h) ----
   Same arguments and return values as the reference functions; the seek-table state that the
   reference keeps in file statics lives in an hx_xing object. */
typedef struct hx_xing hx_xing;
hx_xing *hx_xing_create(void);
void hx_xing_destroy(hx_xing *x);
/* xhead.c:255 XingHeader: builds the tag frame into buf, returns its size in bytes (0 = does not fit) */
int hx_xing_header(hx_xing *x, int samprate, int h_mode, int cr_bit, int original_bit, int flags, int frames,
                   int bs_bytes, int vbr_scale, const unsigned char *toc, unsigned char *buf,
                   const unsigned char *buf20, const unsigned char *buf20b, int kbps);
/* xhead.c:695 XingHeaderTOC: record a seek point; returns how many encode calls to wait for the next one */
int hx_xing_toc(hx_xing *x, int frames, int bs_bytes);
/* xhead.c:486 XingHeaderUpdateInfo: final frame / byte counts, TOC, LAME info fields and CRCs; 1 = ok */
int hx_xing_update_info(hx_xing *x, unsigned frames, int bs_bytes, int vbr_scale, const unsigned char *toc,
                        unsigned char *buf, const unsigned char *buf20, const unsigned char *buf20b,
                        unsigned long long samples_audio, unsigned bytes_mp3, unsigned lowpass,
                        unsigned in_samplerate, unsigned out_samplerate, unsigned short musiccrc);
/* xhead.c:223 XingHeaderUpdateCRC (MusicCRC), xhead.c:236 XingHeaderBitrateIndex */
unsigned short hx_xing_update_crc(unsigned short crc, const unsigned char *data, int len);
int hx_xing_bitrate_index(int mpeg1, int kbps);

#ifdef __cplusplus
}
#endif
#endif
