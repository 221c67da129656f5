// hx_cabi.hip - C ABI (include/hmp3_amd.h) and launch sequence of the batched encoder.
// Host runtime: owns the device buffers of a batch (subband carry, spectra, psy data, stream
// state), groups streams into configuration classes and launches K1..K8 on one HIP stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <chrono>
#include <atomic>
#include <string>
#include "../../include/hmp3_amd.h"
#include "hx_types.h"
#include "hx_host.h"
#include "hx_src.h"

// kernels (hx_front.hip / hx_alloc.hip)
// (K1_GPB / K1_THREADS, k_polyphase's tile and launch dimension: hx_types.h)
__global__ void k_polyphase(const int16_t *pcm, long long nsamp, const HxStream *st, const HxParams *prm,
                            const HxGlobalTabs *gt, float *sb, int NG, int SG, const float *pcmf, int nchan, int *eng, int lsf);
__global__ void k_dcfilter(const int16_t *pcm, const float *pcm32, long long nsamp, HxStream *st, const HxParams *prm, float *pcmf, int S, int nchan);
__global__ void k_detect(HxStream *st, const HxParams *prm, const int *eng, unsigned char *flg, int *dbg_metric, unsigned char *bt,
                         unsigned char *btprev, int NG, int S, int lsf);
__global__ void k_spec(const float *sb, const HxStream *st, const HxParams *prm, const HxGlobalTabs *gt, const unsigned char *bt,
                       float *xr, float *etab, float *thr, int *msbase, int NG, int SG);
__global__ void k_spec_direct(const float *sb, const HxStream *st, const HxParams *prm, const HxGlobalTabs *gt, const unsigned char *bt,
                              float *xr, float *etab, float *thr, int *msbase, int NG, int SG);
__global__ void k_msscan(HxStream *st, const HxParams *prm, const int *msbase, const unsigned char *bt, unsigned char *msflag, int *msdec,
                         const float *thr, float *thrprev, int NG, int lsf, float *sb, int SG, const int16_t *pcm, long long nsamp, const float *pcmf, int nchan);
__global__ void k_prep(const float *xr, float *xmag_dbg, float *x34o, unsigned *sgn, HxBandPrep *band, const HxStream *st, const HxParams *prm, const HxGlobalTabs *gt,
                       const unsigned char *bt, const unsigned char *msflag, const float *etab, const float *thr, const float *thrprev, int NG, long long nunits);
__global__ void k_pack(const HxStream *st, const HxParams *prm, const HxGlobalTabs *gt, const short *ixq, const unsigned *sgn, const HxSegOut *seg,
                       const HxFrameOut *frm, const HxSlot *slots, unsigned char *out, long long out_stride, unsigned char *packet, int *status,
                       int frames_per_stream, int NG, int lsf, long long nframes_total, int solo, HxStream *st_w, const int *pre_len, const int *out_bytes,
                       const int *carry_len, unsigned *frames_out, unsigned char *host_out, const int *seq_src);
__global__ void k_pack_carry(HxStream *st, const unsigned char *out, long long out_stride, const int *out_bytes, const int *carry_len, unsigned *frames_out);
__global__ void k_pack_pre(const HxStream *st, unsigned char *out, long long out_stride, const int *pre_len);
__global__ void k_order(const unsigned *dur, int *order, int S);
__global__ void k_gate(const unsigned *done_counter, unsigned base, unsigned need, int *timeouts);
__global__ void k_alloc(AllocArgs a);
__global__ void k_alloc_slim(AllocArgs a);
__global__ void k_alloc_lsf(AllocArgs a);
__global__ void k_alloc1(AllocArgs a);
__global__ void k_alloc1_lsf(AllocArgs a);
extern "C" int k_alloc_lds_bytes();
extern "C" int k_alloc_slim_lds_bytes();
extern "C" int k_alloc_slim_persistent();      // 1: the kernel's workgroups claim streams from a counter (hx_alloc3.inc, HX_PERSIST)
extern "C" int k_alloc_lsf_lds_bytes();
extern "C" int k_alloc1_lds_bytes();
extern "C" int k_alloc1_lsf_lds_bytes();
#ifdef HX_DYN_LDS
#define K6_LDS(name) ((size_t) name##_lds_bytes())
#else
#define K6_LDS(name) ((size_t) 0)
#endif

static thread_local std::string g_err;
static void set_err(const char *fmt, const char *a = "")
{
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, a);
    g_err = buf;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err("HIP error: %s", hipGetErrorString(e_)); return -1; } } while (0)
#define HIPCHKN(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err("HIP error: %s", hipGetErrorString(e_)); return nullptr; } } while (0)
// every kernel launch is checked where it is made: a bad configuration or a lost device is reported
// with the kernel's name instead of surfacing at some later call
#define LAUNCH_LDS(kernel, grid, block, lds, stream, ...) do { hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); \
        hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { set_err("launch of " #kernel " failed: %s", hipGetErrorString(e_)); return -1; } } while (0)
#define LAUNCH(kernel, grid, block, stream, ...) LAUNCH_LDS(kernel, grid, block, 0, stream, __VA_ARGS__)

#define HX_PARK_MAX 64
struct hx_batch {
    int device = 0, S = 0, maxF = 0, ncls = 0;
    std::vector<HxParams> params;       // host copy per class
    std::vector<int> cls_of;            // stream -> class
    HxParams *d_prm = nullptr;
    HxGlobalTabs *d_gt = nullptr;
    HxStream *d_st = nullptr;
    float *d_sb = nullptr, *d_xr = nullptr, *d_etab = nullptr, *d_thr = nullptr;
    // k_msscan / k_prep -> k_alloc: x^(3/4), signs, band start values, stereo decision, pre-echo memory at call start
    float *d_x34 = nullptr, *d_thrprev = nullptr, *d_xrdbg = nullptr;
    unsigned *d_sgn = nullptr;          // the lines' signs, one bit per line: [S][NG][2][HX_SGN_WORDS]
    unsigned char *d_msflag = nullptr;
    HxBandPrep *d_band = nullptr;
    int *d_msdec = nullptr;
    // k_alloc -> k_pack: quantised lines, segment and frame records, slot lists
    short *d_ixq = nullptr, *d_ixq2 = nullptr;          // (second set: the submit path, where call n + 1 is allocated while call n is packed)
    HxSegOut *d_seg = nullptr, *d_seg2 = nullptr;
    HxFrameOut *d_frm = nullptr, *d_frm2 = nullptr;
    HxSlot *d_slots = nullptr, *d_slots2 = nullptr;
    unsigned *d_sgn3 = nullptr;                    // third set of the signs: written by the front end of call n + 2 while call n is packed
    int *d_lens = nullptr;                              // [2 sets][pre_len | carry_len][S]
    int *frame_stats = nullptr;         // caller's per-frame counters (device), optional
    unsigned char *pk_buf = nullptr; long long pk_stride = 0; int *pk_bytes = nullptr;   // caller's packet buffers (device), optional
    float *d_pcmf = nullptr;            // DC-blocked input, only when a stream uses filter_select = 1
    bool any_dc = false;
    int nchan = 2;                      // channels of the PCM input, the same for every stream of the batch
    int lsf = 0;                        // 1: an MPEG-2 LSF batch (16 / 22.05 / 24 kHz): every 1152-sample block yields two frames
    int slim = 0;                       // 1: the low-footprint stream walk k_alloc_slim (six streams per CU instead of four), chosen at create
    int alloc1 = 0;                     // 1: streams of the first-generation allocator (intensity stereo, dual channel): k_alloc1*
    int *d_eng = nullptr, *d_msbase = nullptr, *d_status = nullptr, *d_dbgmetric = nullptr;
    unsigned char *d_flg = nullptr, *d_bt = nullptr, *d_btprev = nullptr;
    HxFrameDebug *d_dbg = nullptr;
    unsigned long long *d_prof = nullptr;
    int lastNG = 0;                     // NG of the previous call (layout of the carry)
    bool debug = false;
    // staging for the host-buffer entry points
    int16_t *d_pcm = nullptr; unsigned char *d_out = nullptr; int *d_outbytes = nullptr;
    long long pcm_cap = 0, out_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double alloc_ms_sum = 0; int alloc_calls = 0;
    // hx_batch_submit_*: the front-end kernels of call n+1 run (low-priority stream) while k_alloc of
    // call n (high-priority stream) works through its slowest streams; a second set of the buffers
    // that hand granules from the front end to k_alloc makes that safe.
    float *d_xr2 = nullptr, *d_etab2 = nullptr, *d_thr2 = nullptr, *d_x342 = nullptr, *d_thrprev2 = nullptr;
    int *d_msbase2 = nullptr, *d_msdec2 = nullptr;
    unsigned char *d_bt2 = nullptr, *d_btprev2 = nullptr, *d_msflag2 = nullptr;
    unsigned *d_sgn2 = nullptr;
    HxBandPrep *d_band2 = nullptr;
    hipStream_t s_front = nullptr, s_alloc = nullptr, s_pack = nullptr;
    hipEvent_t ev_in = nullptr, ev_front[2] = {nullptr, nullptr}, ev_alloc[2] = {nullptr, nullptr};     // ev_alloc: a submit's packing is done (everything is)
    hipEvent_t ev_k6[2] = {nullptr, nullptr};           // a submit's allocator launch is done
    hipEvent_t ev_sgn[3] = {nullptr, nullptr, nullptr}; // the packing that read this set of signs is done
    // the packing of the latest device-buffer submit, not enqueued yet: it goes out behind the next submit's allocator launch
    // (released by a gate like the front end, into that launch's tail), or ungated at the next wait / plain call
    struct PackJob { bool pending = false; unsigned char *d_out = nullptr; long long out_stride = 0; int *d_out_bytes = nullptr; int nframes = 0, set = 0, sset = 0; } pack_job;
    long long nsubmit = 0;
    bool inflight = false;
    // hx_batch_submit_*_host: device staging for two calls in flight and the copy streams
    void *hs_pcm[2] = {nullptr, nullptr}; unsigned char *hs_out[2] = {nullptr, nullptr}; int *hs_nb[2] = {nullptr, nullptr};
    long long hs_pcm_cap = 0, hs_out_cap = 0;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr, s_host = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr}, ev_hfront[2] = {nullptr, nullptr};
    long long nhost = 0;
    unsigned *d_dur = nullptr;          // [S] duration of each stream's allocator workgroup in the last launch
    int *d_order = nullptr;             // [S] workgroup -> stream for the next launch (used when the batch exceeds what the chip holds at once)
    int *d_done = nullptr;              // [0] streams retired, [2] streams started by all k_alloc launches of this batch (wrap), [1] gate time-outs, [3] line passes on the double x^(4/3) table, [4] certified band sums that fell back to the strict sum
    int resident = 0;                   // allocator workgroups the device holds at once
    long long alloc_launches = 0;
    unsigned long long cfg_hash = 0;    // fingerprint of the resolved configuration classes (checkpoint blobs carry their stream's)
    // a submit's front end is released once this share of the previous allocator launch's resident set has started.
    // Not 100: the gate's own wavefront holds register space on one SIMD, so the last allocator workgroup of a full
    // chip cannot start before a stream retires (measured: 10 .. 99 % all give the same step time, 100 % loses 30 %)
    int gate_percent = 90;
    bool capturing = false;             // the pass is being recorded into a HIP graph (hx_enc_*): no timing events, nothing that queries the stream
    unsigned *cap_frames = nullptr;     // one-stream encoder: where k_pack_carry leaves the stream's frame counter (next to the byte count)
    unsigned char *cap_host = nullptr;  // one-stream encoder: page-locked host memory the packing workgroup publishes the call's results to (hx_pack.hip)
    bool poisoned = false;              // a HIP call failed in the middle of a pass: the event bookkeeping is incomplete, further calls are refused
    // longest-first workgroup order: 2 = for every batch with more streams than the chip has CUs (default: below that no two
    // streams share a CU and the order decides nothing), 3 = always (tests), 1 = only for batches beyond the resident set,
    // 0 = never (HMP3AMD_LPT).
    // Beyond the resident set it keeps the launch's last round short.  Within it the order decides which streams share a CU:
    // workgroups are dealt over XCDs and CUs in turn, so a CU's four streams are 256 apart in launch order - in stream order
    // those are streams of one residue class, and a batch whose slow streams recur with a period (BASELINE config 5: correlation
    // by stream mod 4) had them all on the same CUs; sorted by the previous call's duration a CU gets one stream of each quartile.
    // (Round 4: config 2 +1.0 %, its worst-case signal set +2.1 %.)
    int lpt = 2;
    int ncu = 256;                      // compute units of the device
    int park_k = 8;                     // HMP3AMD_PARK: the CUs of this many longest streams are kept free of other kernels' workgroups (0 = off; see hx_alloc3.inc, "parking")
    int park_pair = 0;                  // HMP3AMD_PARK_PAIR=1: also the CU that shares the instruction cache with a straggler's
    int front_chunk = 0;                // HMP3AMD_FRONT_CHUNK: streams per pass of the front-end chain (0 = the whole batch at once)
    int strict_sums = 0;                // HMP3AMD_EXACT_SUMS=1: the stream walk adds every band in line order instead of certifying a parallel sum (tests)
};

extern "C" const char *hx_last_error(void) { return g_err.c_str(); }

// hash of the sources this library was built from (hmp3_amd/build.sh passes it): profiles/ and bench.py use it to tell
// whether committed counter profiles belong to the loaded build
#ifndef HX_BUILD_ID
#define HX_BUILD_ID "unknown"
#endif
extern "C" const char *hx_build_id(void) { return HX_BUILD_ID; }

extern "C" int hx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void hx_default_control(HX_E_CONTROL *ec) { hx_host_default_control((HxControl *) ec); }

static int flush_pack(hx_batch *b, long long gate_base);

extern "C" void hx_batch_destroy(hx_batch *b)
{
    if (!b) return;
    hipSetDevice(b->device);
    // A packing that was never asked for (no hx_batch_wait / plain call / status read after the last device-buffer submit)
    // is dropped, not enqueued: it would write into output buffers the caller may have freed already.
    b->pack_job.pending = false;
    hipDeviceSynchronize();
    void *ptrs[] = {b->d_prm, b->d_gt, b->d_st, b->d_sb, b->d_xr, b->d_etab, b->d_thr, b->d_eng, b->d_msbase,
                    b->d_status, b->d_dbgmetric, b->d_flg, b->d_bt, b->d_btprev, b->d_dbg, b->d_pcm, b->d_out, b->d_outbytes, b->d_pcmf, b->d_prof,
                    b->d_xr2, b->d_etab2, b->d_thr2, b->d_msbase2, b->d_bt2, b->d_btprev2, b->d_done,
                    b->d_x34, b->d_thrprev, b->d_xrdbg, b->d_sgn, b->d_msflag, b->d_band, b->d_msdec,
                    b->d_x342, b->d_thrprev2, b->d_sgn2, b->d_msflag2, b->d_band2, b->d_msdec2,
                    b->d_ixq, b->d_seg, b->d_frm, b->d_slots, b->d_dur, b->d_order,
                    b->d_ixq2, b->d_seg2, b->d_frm2, b->d_slots2, b->d_sgn3, b->d_lens};
    for (void *p : ptrs) if (p) hipFree(p);
    for (int i = 0; i < 2; i++) {
        if (b->hs_pcm[i]) hipFree(b->hs_pcm[i]);
        if (b->hs_out[i]) hipFree(b->hs_out[i]);
        if (b->hs_nb[i]) hipFree(b->hs_nb[i]);
        if (b->ev_h2d[i]) hipEventDestroy(b->ev_h2d[i]);
        if (b->ev_d2h[i]) hipEventDestroy(b->ev_d2h[i]);
        if (b->ev_hfront[i]) hipEventDestroy(b->ev_hfront[i]);
    }
    if (b->s_h2d) hipStreamDestroy(b->s_h2d);
    if (b->s_d2h) hipStreamDestroy(b->s_d2h);
    if (b->s_host) hipStreamDestroy(b->s_host);
    if (b->s_front) hipStreamDestroy(b->s_front);
    if (b->s_alloc) hipStreamDestroy(b->s_alloc);
    if (b->s_pack) hipStreamDestroy(b->s_pack);
    hipEvent_t evs[] = {b->ev_in, b->ev_front[0], b->ev_front[1], b->ev_alloc[0], b->ev_alloc[1], b->ev_k6[0], b->ev_k6[1], b->ev_sgn[0], b->ev_sgn[1], b->ev_sgn[2]};
    for (hipEvent_t e : evs) if (e) hipEventDestroy(e);
    for (auto &pr : b->pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    delete b;
}

extern "C" hx_batch *hx_batch_create(int device, int nstreams, const HX_E_CONTROL *ec, int shared_control, int max_frames)
{
    if (nstreams <= 0 || max_frames <= 0 || !ec) { set_err("bad arguments"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_err("no HIP device available: the encoder has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= ndev) { set_err("device index out of range"); return nullptr; }
    {   // written for gfx950 (MI355X) only: the code object holds no other target
        hipDeviceProp_t prop;
        HIPCHKN(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_err("device is %s: this library runs on gfx950 (MI355X) only", prop.gcnArchName); return nullptr; }
    }
    // the launch bookkeeping (streams x granules x 9 energies per channel) is 32-bit
    if ((long long) nstreams * max_frames > 32LL * 1024 * 1024) { set_err("nstreams * max_frames exceeds 32 Mi frames per call: split the batch"); return nullptr; }
    HIPCHKN(hipSetDevice(device));
    hx_batch *b = new hx_batch;
    b->device = device; b->S = nstreams; b->maxF = max_frames;
    b->cls_of.resize(nstreams);
    // group streams into configuration classes
    std::vector<HxControl> seen;
    for (int s = 0; s < nstreams; s++) {
        const HxControl *c = (const HxControl *) (shared_control ? ec : ec + s);
        int k = -1;
        for (size_t i = 0; i < seen.size(); i++) if (memcmp(&seen[i], c, sizeof(HxControl)) == 0) { k = (int) i; break; }
        if (k < 0) {
            HxParams p;
            if (!hx_resolve(c, &p)) {
                // (a limit of this library's layout is named as such: the reference would have taken the configuration)
                if (*hx_resolve_error()) set_err("configuration rejected: %s", hx_resolve_error());
                else set_err("configuration rejected (the reference's L3_audio_encode_init returns 0 for it)");
                delete b; return nullptr;
            }
            if (p.filter_dc) b->any_dc = true;
            if (b->params.empty()) { b->nchan = p.nchan; b->lsf = p.h_id ? 0 : 1; b->alloc1 = p.alloc1; }
            else if (p.alloc1 != b->alloc1) { set_err("intensity-stereo / dual-channel streams (first-generation allocator) cannot share a batch with the others"); delete b; return nullptr; }
            else if (p.nchan != b->nchan) { set_err("mono and stereo streams cannot share a batch (the PCM layout differs)"); delete b; return nullptr; }
            else if ((p.h_id ? 0 : 1) != b->lsf) { set_err("MPEG-1 and MPEG-2 sample rates cannot share a batch (frames per call differ)"); delete b; return nullptr; }
            seen.push_back(*c);
            b->params.push_back(p);
            k = (int) seen.size() - 1;
        }
        b->cls_of[s] = k;
        if (shared_control) { for (int t = 1; t < nstreams; t++) b->cls_of[t] = 0; break; }
    }
    b->ncls = (int) b->params.size();
    const long long S = nstreams, NG = 2LL * max_frames;
    std::vector<HxGlobalTabs> gt_host(1);       // (144 KB: not on the stack)
    HxGlobalTabs &gt = gt_host[0];
    hx_global_tabs(&gt);
    std::vector<HxStream> st(nstreams);
    for (int s = 0; s < nstreams; s++) hx_stream_reset(&b->params[b->cls_of[s]], b->cls_of[s], &st[s]);
#define ALLOC(ptr, bytes) do { if (hipMalloc((void **) &(ptr), (size_t) (bytes)) != hipSuccess) { set_err("hipMalloc failed"); hx_batch_destroy(b); return nullptr; } } while (0)
    ALLOC(b->d_prm, sizeof(HxParams) * b->ncls + 256);        // (k_spec reads a spreading row in 16-byte pieces, up to 60 bytes past its end)
    ALLOC(b->d_gt, sizeof(HxGlobalTabs));
    ALLOC(b->d_st, sizeof(HxStream) * S);
    ALLOC(b->d_sb, sizeof(float) * S * 2 * (NG + 3) * 576);
    ALLOC(b->d_xr, sizeof(float) * S * NG * 1152);
    ALLOC(b->d_etab, sizeof(float) * S * NG * 128);
    ALLOC(b->d_thr, sizeof(float) * S * NG * 128);
    ALLOC(b->d_x34, sizeof(float) * S * NG * 1152);
    ALLOC(b->d_sgn, sizeof(unsigned) * S * NG * 2 * HX_SGN_WORDS);
    ALLOC(b->d_band, sizeof(HxBandPrep) * S * NG);
    ALLOC(b->d_msflag, S * NG);
    ALLOC(b->d_msdec, sizeof(int) * S * NG);
    ALLOC(b->d_thrprev, sizeof(float) * S * 128);
    ALLOC(b->d_ixq, sizeof(short) * S * NG * 1152);
    ALLOC(b->d_seg, sizeof(HxSegOut) * S * NG * 2);
    ALLOC(b->d_frm, sizeof(HxFrameOut) * S * NG);
    ALLOC(b->d_slots, sizeof(HxSlot) * S * (NG + HX_SLOTS_EXTRA));
    ALLOC(b->d_eng, sizeof(int) * S * 2 * NG * 9);
    ALLOC(b->d_msbase, sizeof(int) * S * NG);
    ALLOC(b->d_flg, S * NG);
    ALLOC(b->d_bt, S * NG);
    ALLOC(b->d_btprev, S);
    ALLOC(b->d_status, sizeof(int));
    ALLOC(b->d_outbytes, sizeof(int) * S);
    if (const char *e = getenv("HMP3AMD_LPT")) b->lpt = atoi(e);
    if (const char *e = getenv("HMP3AMD_EXACT_SUMS")) b->strict_sums = atoi(e) != 0;
    if (const char *e = getenv("HMP3AMD_FRONT_CHUNK")) b->front_chunk = atoi(e);
    if (const char *e = getenv("HMP3AMD_PARK_PAIR")) b->park_pair = atoi(e) != 0;
    if (const char *e = getenv("HMP3AMD_PARK")) { b->park_k = atoi(e); if (b->park_k < 0) b->park_k = 0; if (b->park_k > HX_PARK_MAX) b->park_k = HX_PARK_MAX; }
    ALLOC(b->d_lens, sizeof(int) * 4 * S);
    ALLOC(b->d_dur, sizeof(unsigned) * 2 * S);        // [S] durations, [S] where workgroup i of the last launch ran (see "place")
    ALLOC(b->d_order, sizeof(int) * S);
    HIPCHKN(hipMemset(b->d_dur, 0, sizeof(unsigned) * 2 * S));
    ALLOC(b->d_done, (8 + HX_PARK_MAX) * sizeof(int));       // ([8 ..]: CU ids of the parking scheme, hx_alloc3.inc)
    HIPCHKN(hipMemset(b->d_done, 0, (8 + HX_PARK_MAX) * sizeof(int)));
    {
        hipDeviceProp_t prop;
        int per_cu = 0;
        HIPCHKN(hipGetDeviceProperties(&prop, device));
        const void *kern = b->alloc1 ? (b->lsf ? (const void *) k_alloc1_lsf : (const void *) k_alloc1) : (b->lsf ? (const void *) k_alloc_lsf : (const void *) k_alloc);
        const size_t dyn = b->alloc1 ? (b->lsf ? K6_LDS(k_alloc1_lsf) : K6_LDS(k_alloc1)) : (b->lsf ? K6_LDS(k_alloc_lsf) : K6_LDS(k_alloc));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 128, dyn) != hipSuccess || per_cu <= 0) per_cu = 4;
        b->resident = per_cu * prop.multiProcessorCount;
        b->ncu = prop.multiProcessorCount;
        // Two builds of the MPEG-1 stream walk.  A batch that the chip holds at once (config 2: 1024 streams on 256 CUs x 4)
        // runs the one written for 256 registers and 38 KB of LDS per stream; a larger one runs k_alloc_slim, whose streams
        // take 168 registers and 26.5 KB, six to a CU: a stream is slower there, 1.5 x as many are in flight.
        // HMP3AMD_K6 = fat | slim overrides the choice (tests run every case on both).
        bool slim_ok = !b->alloc1 && !b->lsf;
        for (int k = 0; k < b->ncls && slim_ok; k++) slim_ok = hx_slim_tables_ok(&b->params[k], &gt) != 0;
        const char *e = getenv("HMP3AMD_K6");
        if (e && strcmp(e, "slim") != 0 && strcmp(e, "fat") != 0) { set_err("HMP3AMD_K6 must be 'fat' or 'slim'"); hx_batch_destroy(b); return nullptr; }
        const bool want = e ? (strcmp(e, "slim") == 0) : (S > b->resident);
        if (e && strcmp(e, "slim") == 0 && !slim_ok && !b->alloc1 && !b->lsf) { set_err("HMP3AMD_K6=slim: the host's tables do not have the structure k_alloc_slim derives them from"); hx_batch_destroy(b); return nullptr; }
        if (want && slim_ok) {
            b->slim = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) k_alloc_slim, 128, K6_LDS(k_alloc_slim)) != hipSuccess || per_cu <= 0) per_cu = 6;
            b->resident = per_cu * prop.multiProcessorCount;
        }
    }
    if (b->any_dc) ALLOC(b->d_pcmf, sizeof(float) * S * max_frames * 1152 * b->nchan);
    HIPCHKN(hipMemcpy(b->d_prm, b->params.data(), sizeof(HxParams) * b->ncls, hipMemcpyHostToDevice));
    HIPCHKN(hipMemcpy(b->d_gt, &gt, sizeof(gt), hipMemcpyHostToDevice));
    HIPCHKN(hipMemcpy(b->d_st, st.data(), sizeof(HxStream) * S, hipMemcpyHostToDevice));
    HIPCHKN(hipMemset(b->d_sb, 0, sizeof(float) * S * 2 * (NG + 3) * 576));
    HIPCHKN(hipMemset(b->d_status, 0, sizeof(int)));
    b->lastNG = 0;
    return b;
}

extern "C" int hx_batch_nstreams(const hx_batch *b) { return b ? b->S : 0; }

// Checkpoint of one stream: its HxStream record followed by the three carried subband granules of each
// channel.  With it a stream continues in another slot, another batch of the same configuration, another GPU or
// after a restart exactly where it stopped (the reference's equivalent is a copy of the CMp3Enc object).
// The blob starts with a header {magic, format version, sizeof(HxStream), fingerprint of the stream's resolved
// configuration}: a blob from another library build (other state layout) or saved under another control is
// refused instead of silently yielding a corrupt bitstream.
struct HxStateHeader { unsigned magic, version, state_bytes, pad; unsigned long long cfg; };
#define HX_STATE_MAGIC 0x53335848u      // "HX3S"
#define HX_STATE_VERSION 3u
static unsigned long long cfg_fingerprint(const HxParams &p)
{
    unsigned long long h = 1469598103934665603ull;      // FNV-1a over the echoed control and the derived frame constants
    auto mix = [&](const void *d, size_t n) { const unsigned char *c = (const unsigned char *) d; for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; } };
    mix(&p.ec, sizeof(p.ec));
    const int v[] = {p.totbitrate, p.samprate, p.h_mode, p.h_id, p.nchan, p.nsb_limit, p.band_limit, p.framebytes, p.main_framebytes, p.side_bytes,
                     p.ms_flag, p.hf_flag, p.vbr_flag, p.initialMNR, p.short_block_threshold};
    mix(v, sizeof(v));
    return h;
}
extern "C" long long hx_batch_stream_state_bytes(const hx_batch *b) { (void) b; return (long long) (sizeof(HxStateHeader) + sizeof(HxStream) + 2 * 3 * 576 * sizeof(float)); }

static int stream_state_copy(hx_batch *b, int i, void *host, bool save)
{
    if (!b || i < 0 || i >= b->S || !host) { set_err("bad arguments"); return -1; }
    HIPCHK(hipSetDevice(b->device));
    if (b->s_pack) flush_pack(b, -1);
    HIPCHK(hipDeviceSynchronize());
    HxStateHeader hd = {HX_STATE_MAGIC, HX_STATE_VERSION, (unsigned) sizeof(HxStream), 0, cfg_fingerprint(b->params[b->cls_of[i]])};
    if (save) memcpy(host, &hd, sizeof(hd));
    else {
        HxStateHeader in;
        memcpy(&in, host, sizeof(in));
        if (in.magic != HX_STATE_MAGIC || in.version != HX_STATE_VERSION || in.state_bytes != hd.state_bytes) { set_err("not a stream-state blob of this library build"); return -1; }
        if (in.cfg != hd.cfg) { set_err("the stream state was saved under a different configuration than slot's"); return -1; }
    }
    char *h = (char *) host + sizeof(HxStateHeader);
    const size_t per = (size_t) (2 * b->maxF + 3) * 576;        // floats per (stream, channel) in the subband buffer
    if (save) {
        HIPCHK(hipMemcpy(h, b->d_st + i, sizeof(HxStream), hipMemcpyDeviceToHost));
        for (int c = 0; c < 2; c++)
            HIPCHK(hipMemcpy(h + sizeof(HxStream) + (size_t) c * 3 * 576 * sizeof(float), b->d_sb + ((size_t) i * 2 + c) * per, 3 * 576 * sizeof(float), hipMemcpyDeviceToHost));
    } else {
        HxStream st;
        memcpy(&st, h, sizeof(HxStream));
        st.cls = b->cls_of[i];      // the class index is the receiving batch's
        HIPCHK(hipMemcpy(b->d_st + i, &st, sizeof(HxStream), hipMemcpyHostToDevice));
        for (int c = 0; c < 2; c++)
            HIPCHK(hipMemcpy(b->d_sb + ((size_t) i * 2 + c) * per, h + sizeof(HxStream) + (size_t) c * 3 * 576 * sizeof(float), 3 * 576 * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}
extern "C" int hx_batch_get_stream_state(hx_batch *b, int i, void *dst) { return stream_state_copy(b, i, dst, true); }
extern "C" int hx_batch_set_stream_state(hx_batch *b, int i, const void *src) { return stream_state_copy(b, i, (void *) src, false); }

// Start a new stream in slot i (same configuration as the slot's previous stream): the state a freshly created
// batch would have for it - reservoir, histories, allocator feedback, subband carry - so a long-lived batch can
// take over new inputs as old ones end.  Waits for the work in flight; the other streams are not touched.
extern "C" int hx_batch_reset_stream(hx_batch *b, int i)
{
    if (!b || i < 0 || i >= b->S) { set_err("stream index out of range"); return -1; }
    HIPCHK(hipSetDevice(b->device));
    if (b->s_pack) flush_pack(b, -1);
    HIPCHK(hipDeviceSynchronize());
    HxStream *st = new HxStream;
    hx_stream_reset(&b->params[b->cls_of[i]], b->cls_of[i], st);
    hipError_t e = hipMemcpy(b->d_st + i, st, sizeof(HxStream), hipMemcpyHostToDevice);
    delete st;
    if (e != hipSuccess) { set_err("HIP error: %s", hipGetErrorString(e)); return -1; }
    const size_t per = (size_t) (2 * b->maxF + 3) * 576 * sizeof(float);     // subband slots of one (stream, channel)
    HIPCHK(hipMemset((char *) b->d_sb + (size_t) i * 2 * per, 0, 2 * per));
    return 0;
}

extern "C" long long hx_batch_out_stride(const hx_batch *b, int nframes)
{
    if (!b) return 0;
    // nframes new frames plus the images of the frames still pending from earlier calls (their
    // free space is at most the 511-byte reservoir, so a handful of frames; 4 KB covers them)
    int maxframe = 0;
    for (const HxParams &p : b->params) {
        int fb = p.vbr_flag ? p.vbr_framebytes[p.ivbr_max] : p.framebytes + 1;
        if (fb > maxframe) maxframe = fb;
    }
    long long n = (long long) ((b->lsf ? 2 : 1) * nframes + 2) * maxframe + 4096;
    return (n + 255) & ~255LL;
}

extern "C" void hx_batch_packet_buffers(hx_batch *b, unsigned char *d_packet, long long frame_stride, int *d_packet_bytes)
{
    b->pk_buf = d_packet; b->pk_stride = frame_stride; b->pk_bytes = d_packet_bytes;
}

extern "C" void hx_batch_frame_stats_buffer(hx_batch *b, int *d_stats) { b->frame_stats = d_stats; }

extern "C" void hx_batch_debug_enable(hx_batch *b, int on)
{
    b->debug = on != 0;
    if (on && !b->d_dbg) {
        hipSetDevice(b->device);
        hipMalloc((void **) &b->d_dbg, sizeof(HxFrameDebug) * (size_t) b->S * b->maxF);
        hipMalloc((void **) &b->d_xrdbg, sizeof(float) * (size_t) b->S * 2 * b->maxF * 1152);
        hipMalloc((void **) &b->d_dbgmetric, sizeof(int) * (size_t) b->S * 2 * b->maxF * 2);
        hipMalloc((void **) &b->d_prof, sizeof(unsigned long long) * (size_t) b->S * 64);
        hipMemset(b->d_prof, 0, sizeof(unsigned long long) * (size_t) b->S * 64);
    }
}

// streams, events and the second buffer set of the submit path, created at the first submit
static int pipe_init(hx_batch *b)
{
    if (b->s_front) return 0;
    const long long S = b->S, NG = 2LL * b->maxF;
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));          // lo = least urgent, hi = most urgent
    HIPCHK(hipStreamCreateWithPriority(&b->s_front, hipStreamNonBlocking, lo));
    HIPCHK(hipStreamCreateWithPriority(&b->s_alloc, hipStreamNonBlocking, hi));
    HIPCHK(hipStreamCreateWithPriority(&b->s_pack, hipStreamNonBlocking, lo));
    HIPCHK(hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipEventCreateWithFlags(&b->ev_front[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->ev_alloc[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->ev_k6[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 3; i++) HIPCHK(hipEventCreateWithFlags(&b->ev_sgn[i], hipEventDisableTiming));
    HIPCHK(hipMalloc((void **) &b->d_ixq2, sizeof(short) * S * NG * 1152));
    HIPCHK(hipMalloc((void **) &b->d_seg2, sizeof(HxSegOut) * S * NG * 2));
    HIPCHK(hipMalloc((void **) &b->d_frm2, sizeof(HxFrameOut) * S * NG));
    HIPCHK(hipMalloc((void **) &b->d_slots2, sizeof(HxSlot) * S * (NG + HX_SLOTS_EXTRA)));
    HIPCHK(hipMalloc((void **) &b->d_sgn3, sizeof(unsigned) * S * NG * 2 * HX_SGN_WORDS));
    HIPCHK(hipMalloc((void **) &b->d_xr2, sizeof(float) * S * NG * 1152));
    HIPCHK(hipMalloc((void **) &b->d_etab2, sizeof(float) * S * NG * 128));
    HIPCHK(hipMalloc((void **) &b->d_thr2, sizeof(float) * S * NG * 128));
    HIPCHK(hipMalloc((void **) &b->d_msbase2, sizeof(int) * S * NG));
    HIPCHK(hipMalloc((void **) &b->d_bt2, S * NG));
    HIPCHK(hipMalloc((void **) &b->d_btprev2, S));
    HIPCHK(hipMalloc((void **) &b->d_x342, sizeof(float) * S * NG * 1152));
    HIPCHK(hipMalloc((void **) &b->d_sgn2, sizeof(unsigned) * S * NG * 2 * HX_SGN_WORDS));
    HIPCHK(hipMalloc((void **) &b->d_band2, sizeof(HxBandPrep) * S * NG));
    HIPCHK(hipMalloc((void **) &b->d_msflag2, S * NG));
    HIPCHK(hipMalloc((void **) &b->d_msdec2, sizeof(int) * S * NG));
    HIPCHK(hipMalloc((void **) &b->d_thrprev2, sizeof(float) * S * 128));
    return 0;
}

// Argument checks of every encode entry point, made before anything is allocated, copied or launched.
static int check_call(const hx_batch *b, const void *pcm, int nframes, const void *out, long long out_stride, const void *out_bytes)
{
    if (!b) { set_err("null batch"); return -1; }
    if (b->poisoned) { set_err("the batch is unusable after a failed device call: destroy it"); return -1; }
    if (nframes <= 0 || nframes > b->maxF) { set_err("nframes out of range (1 .. max_frames of hx_batch_create)"); return -1; }
    if (!pcm || !out || !out_bytes) { set_err("null buffer"); return -1; }
    if (out_stride < hx_batch_out_stride(b, nframes)) { set_err("out_stride is smaller than hx_batch_out_stride(b, nframes)"); return -1; }
    return 0;
}

// the packing kernels of one call on stream qp (see encode_core)
static int enqueue_pack(hx_batch *b, unsigned char *d_out, long long out_stride, int *d_out_bytes, int nframes, int set, int sset, hipStream_t qp)
{
    const int S = b->S, NG = 2 * nframes;
    unsigned *const x_sgn = sset == 2 ? b->d_sgn3 : (sset ? b->d_sgn2 : b->d_sgn);
    short *const x_ixq = set ? b->d_ixq2 : b->d_ixq;
    HxSegOut *const x_seg = set ? b->d_seg2 : b->d_seg;
    HxFrameOut *const x_frm = set ? b->d_frm2 : b->d_frm;
    HxSlot *const x_slots = set ? b->d_slots2 : b->d_slots;
    int *const x_prelen = b->d_lens + (2 * set) * (long long) b->S, *const x_carrylen = b->d_lens + (2 * set + 1) * (long long) b->S;
    const int fps = (b->lsf ? 2 : 1) * nframes;
    const long long total = (long long) S * fps;
    // a handful of frames in all (the one-stream encoder's calls): one workgroup does the three kernels' work (hx_pack.hip, solo)
    const int solo = (S <= 4 && total <= 8) ? S : 0;
    if (solo) {
        LAUNCH(k_pack, dim3(1), dim3(256), qp, (const HxStream *) b->d_st, (const HxParams *) b->d_prm, (const HxGlobalTabs *) b->d_gt,
               (const short *) x_ixq, (const unsigned *) x_sgn, (const HxSegOut *) x_seg, (const HxFrameOut *) x_frm, (const HxSlot *) x_slots,
               d_out, out_stride, b->pk_buf, b->d_status, fps, NG, b->lsf, total, solo, b->d_st, (const int *) x_prelen, (const int *) d_out_bytes, (const int *) x_carrylen, b->cap_frames,
               (solo == 1) ? b->cap_host : (unsigned char *) nullptr, (const int *) (b->d_done + 2));
        return 0;
    }
    LAUNCH(k_pack_pre, dim3(S), dim3(64), qp, (const HxStream *) b->d_st, d_out, out_stride, (const int *) x_prelen);
    LAUNCH(k_pack, dim3((unsigned) (total < 8LL * 256 * 8 ? total : 8LL * 256 * 8)), dim3(256), qp, (const HxStream *) b->d_st, (const HxParams *) b->d_prm, (const HxGlobalTabs *) b->d_gt,
           (const short *) x_ixq, (const unsigned *) x_sgn, (const HxSegOut *) x_seg, (const HxFrameOut *) x_frm, (const HxSlot *) x_slots,
           d_out, out_stride, b->pk_buf, b->d_status, fps, NG, b->lsf, total, 0, (HxStream *) nullptr, (const int *) nullptr, (const int *) nullptr, (const int *) nullptr, (unsigned *) nullptr, (unsigned char *) nullptr, (const int *) nullptr);
    LAUNCH(k_pack_carry, dim3(S), dim3(64), qp, b->d_st, (const unsigned char *) d_out, out_stride, (const int *) d_out_bytes, (const int *) x_carrylen, b->cap_frames);
    return 0;
}

// the deferred packing of the latest submit: out now, on the packing stream; gate_base >= 0: behind a gate on the allocator launch
// that was just enqueued (the one after the job's own)
static int flush_pack(hx_batch *b, long long gate_base)
{
    hx_batch::PackJob &j = b->pack_job;
    if (!j.pending) return 0;
    j.pending = false;
    HIPCHK(hipStreamWaitEvent(b->s_pack, b->ev_k6[j.set], 0));
    if (gate_base >= 0 && b->gate_percent > 0) {
        const long long fill = b->S < b->resident ? b->S : b->resident;
        LAUNCH(k_gate, dim3(1), dim3(64), b->s_pack, (const unsigned *) (b->d_done + 2), (unsigned) gate_base, (unsigned) (fill * b->gate_percent / 100), b->d_done + 1);
    }
    if (enqueue_pack(b, j.d_out, j.out_stride, j.d_out_bytes, j.nframes, j.set, j.sset, b->s_pack) != 0) return -1;
    HIPCHK(hipEventRecord(b->ev_alloc[j.set], b->s_pack));
    HIPCHK(hipEventRecord(b->ev_sgn[j.sset], b->s_pack));
    return 0;
}

// one pass of the pipeline over the batch; the input is int16 (d_pcm) or fp32 at int16 scale (d_pcm32).
// pipelined = 0: every kernel on the caller's stream.  pipelined = 1 (hx_batch_submit_*): front end
// and k_alloc on the batch's own two streams, ordered by events (see hx_batch).
static int encode_pass(hx_batch *b, const int16_t *d_pcm, const float *d_pcm32, int nframes, unsigned char *d_out,
                       long long out_stride, int *d_out_bytes, void *stream, int pipelined);
static int encode_core(hx_batch *b, const int16_t *d_pcm, const float *d_pcm32, int nframes, unsigned char *d_out,
                       long long out_stride, int *d_out_bytes, void *stream, int pipelined = 0)
{
    if (check_call(b, d_pcm ? (const void *) d_pcm : (const void *) d_pcm32, nframes, d_out, out_stride, d_out_bytes) != 0) return -1;
    const int r = encode_pass(b, d_pcm, d_pcm32, nframes, d_out, out_stride, d_out_bytes, stream, pipelined);
    // A launch or HIP call that fails inside a pass leaves events unrecorded and buffer sets half handed over: the batch is
    // not reusable (this only happens on a device error).  Later calls are refused; hx_batch_destroy just synchronises.
    if (r != 0) b->poisoned = true;
    return r;
}
static int encode_pass(hx_batch *b, const int16_t *d_pcm, const float *d_pcm32, int nframes, unsigned char *d_out,
                       long long out_stride, int *d_out_bytes, void *stream, int pipelined)
{
    hipStream_t q = (hipStream_t) stream, qa = q;
    HIPCHK(hipSetDevice(b->device));
    int set = 0, flushed_set = -1;
    if (pipelined) {
        if (pipe_init(b) != 0) return -1;
        if (pipelined == 2 && b->pack_job.pending) {                // (a host-buffer submit behind device-buffer ones)
            flushed_set = b->pack_job.set;
            if (flush_pack(b, -1) != 0) return -1;
        }
        set = (int) (b->nsubmit & 1);
        HIPCHK(hipEventRecord(b->ev_in, q));                        // the caller's PCM is ready from here on
        HIPCHK(hipStreamWaitEvent(b->s_front, b->ev_in, 0));
        if (b->nsubmit >= 2) HIPCHK(hipStreamWaitEvent(b->s_front, b->ev_k6[set], 0));        // k_alloc of submit n-2 is done with this set
        if (b->nsubmit >= 3) HIPCHK(hipStreamWaitEvent(b->s_front, b->ev_sgn[b->nsubmit % 3], 0));     // ... and the packing of submit n-3 with this set of signs
        q = b->s_front; qa = b->s_alloc;
        if (b->alloc_launches > 0 && b->gate_percent > 0) {        // start in the previous allocator kernel's tail, not at its start
            const unsigned base = (unsigned) ((unsigned long long) (b->alloc_launches - 1) * (unsigned long long) b->S);   // wraps with the counter
            const long long fill = b->S < b->resident ? b->S : b->resident;     // workgroups of the previous launch the device holds at once
            const unsigned need = (unsigned) (fill * b->gate_percent / 100);
            LAUNCH(k_gate, dim3(1), dim3(64), q, (const unsigned *) (b->d_done + 2), base, need, b->d_done + 1);
        }
    } else if (b->inflight) {                                       // a plain call behind submits: order it after them
        if (flush_pack(b, -1) != 0) return -1;
        const int last = (int) ((b->nsubmit - 1) & 1);
        HIPCHK(hipStreamWaitEvent(q, b->ev_front[last], 0));
        HIPCHK(hipStreamWaitEvent(q, b->ev_alloc[last], 0));
        b->inflight = false;
    }
    float *const x_xr = set ? b->d_xr2 : b->d_xr, *const x_etab = set ? b->d_etab2 : b->d_etab, *const x_thr = set ? b->d_thr2 : b->d_thr;
    int *const x_msbase = set ? b->d_msbase2 : b->d_msbase;
    unsigned char *const x_bt = set ? b->d_bt2 : b->d_bt, *const x_btprev = set ? b->d_btprev2 : b->d_btprev;
    float *const x_x34 = set ? b->d_x342 : b->d_x34, *const x_thrprev = set ? b->d_thrprev2 : b->d_thrprev;
    // (the signs are also read by the packing, which may still be busy with submit n-2 when the front end of submit n
    // writes them: three sets in rotation)
    const int sset = pipelined ? (int) (b->nsubmit % 3) : 0;
    unsigned *const x_sgn = sset == 2 ? b->d_sgn3 : (sset ? b->d_sgn2 : b->d_sgn);
    unsigned char *const x_msflag = set ? b->d_msflag2 : b->d_msflag;
    short *const x_ixq = set ? b->d_ixq2 : b->d_ixq;
    HxSegOut *const x_seg = set ? b->d_seg2 : b->d_seg;
    HxFrameOut *const x_frm = set ? b->d_frm2 : b->d_frm;
    HxSlot *const x_slots = set ? b->d_slots2 : b->d_slots;
    int *const x_prelen = b->d_lens + (2 * set) * (long long) b->S, *const x_carrylen = b->d_lens + (2 * set + 1) * (long long) b->S;
    HxBandPrep *const x_band = set ? b->d_band2 : b->d_band;
    int *const x_msdec = set ? b->d_msdec2 : b->d_msdec;
    const int S = b->S, NG = 2 * nframes;
    const long long nsamp = 1152LL * nframes;
    // The subband carry sits in slots NG_prev..NG_prev+2 only if the previous call used another
    // frame count; k_msscan always rolls it to slots 0..2, so nothing to do here.
    dim3 g1(S, (NG + K1_GPB - 1) / K1_GPB);
    const int SG = 2 * b->maxF + 3;     // subband slots per (stream, channel): fixed layout
    const float *pcmf = b->any_dc ? b->d_pcmf : d_pcm32;       // fp32 samples the polyphase reads, or null for int16
    if (b->any_dc) LAUNCH(k_dcfilter, dim3((b->nchan * S + 63) / 64), dim3(64), q, d_pcm, d_pcm32, nsamp, b->d_st, b->d_prm, b->d_pcmf, S, b->nchan);
    // (the carry in slots 0..2 is not written by k_polyphase, so the two may run in either order)
    // The chain runs over the whole batch, or (b->front_chunk, HMP3AMD_FRONT_CHUNK) over blocks of streams one after the
    // other: a block's subband samples and spectrum (2.4 + 1.2 MB per stream at 256 frames) are then still in the memory-side
    // cache when the next kernel of the chain reads them.  Every kernel indexes its per-stream arrays from the block's first stream.
    const int C = (b->front_chunk > 0 && b->front_chunk < S && !b->debug) ? b->front_chunk : S;
    auto front = [&](int s0, int Sc) -> int {
        const long long o = s0;
        float *sb_c = b->d_sb + o * 2 * SG * 576;
        int *eng_c = b->d_eng + o * 2 * NG * 9;
        HxStream *st_c = b->d_st + o;
        const int16_t *pcm_c = d_pcm ? d_pcm + o * nsamp * b->nchan : nullptr;
        const float *pcmf_c = pcmf ? pcmf + o * nsamp * b->nchan : nullptr;
        unsigned char *bt_c = x_bt + o * NG;
        dim3 g1c(Sc, (NG + K1_GPB - 1) / K1_GPB);
        // (the detector energies of the carried granule are formed by k_polyphase's first tile of a stream, the carries rolled by
        // k_msscan, flags and block types by one kernel - round 6: three launches less per call, which is what a one-stream call
        // is made of)
        LAUNCH(k_polyphase, g1c, dim3(K1_THREADS), q, pcm_c, nsamp, st_c, b->d_prm, b->d_gt, sb_c, NG, SG, pcmf_c, b->nchan, eng_c, b->lsf);
        LAUNCH(k_detect, dim3((Sc + 3) / 4), dim3(256), q, st_c, b->d_prm, eng_c, b->d_flg + o * NG,
               b->debug ? b->d_dbgmetric : nullptr, bt_c, x_btprev + o, NG, Sc, b->lsf);
        // (the form of K4 that goes with the stream-walk kernel: hx_front.hip, spec_granule)
        if (b->slim) LAUNCH(k_spec_direct, dim3((unsigned) ((long long) Sc * nframes)), dim3(128), q, sb_c, st_c, b->d_prm, b->d_gt, bt_c, x_xr + o * NG * 1152,
                            x_etab + o * NG * 128, x_thr + o * NG * 128, x_msbase + o * NG, NG, SG);
        else LAUNCH(k_spec, dim3((unsigned) ((long long) Sc * nframes)), dim3(128), q, sb_c, st_c, b->d_prm, b->d_gt, bt_c, x_xr + o * NG * 1152,
                    x_etab + o * NG * 128, x_thr + o * NG * 128, x_msbase + o * NG, NG, SG);
        // stereo decisions and the pre-echo hand-over (serial per stream) with the carries of the subband buffer and the PCM
        // history (they belong to the front end: k_alloc does not touch them), then the allocator's state-independent start
        // values per granule; the magnitudes replace the spectrum in place, so the tests' tap of it is taken first
        LAUNCH(k_msscan, dim3(Sc), dim3(64), q, st_c, b->d_prm, x_msbase + o * NG, bt_c, x_msflag + o * NG, x_msdec + o * NG, x_thr + o * NG * 128, x_thrprev + o * 128, NG, b->lsf,
               sb_c, SG, pcm_c, nsamp, pcmf_c, b->nchan);
        LAUNCH(k_prep, dim3((unsigned) (((long long) Sc * NG + 3) / 4)), dim3(256), q, (const float *) (x_xr + o * NG * 1152), (b->debug && b->d_xrdbg) ? b->d_xrdbg : (float *) nullptr, b->debug ? x_x34 : (float *) nullptr,
               x_sgn + o * NG * 2 * HX_SGN_WORDS, x_band + o * NG, st_c, b->d_prm, b->d_gt, bt_c, x_msflag + o * NG,
               x_etab + o * NG * 128, x_thr + o * NG * 128, x_thrprev + o * 128, NG, (long long) Sc * NG);
        return 0;
    };
    for (int s0 = 0; s0 < S; s0 += C) if (front(s0, (S - s0 < C) ? S - s0 : C) != 0) return -1;
    if (pipelined) {
        HIPCHK(hipEventRecord(b->ev_front[set], q));
        HIPCHK(hipStreamWaitEvent(qa, b->ev_front[set], 0));
        if (b->nsubmit >= 2) HIPCHK(hipStreamWaitEvent(qa, b->ev_alloc[set], 0));     // the packing of submit n-2 is done with the lines / records of this set
        // A caller that hands consecutive submits the same output buffers gets them one after the other: the previous
        // submit's packing (which writes and reads its `out`) then has to be through before this allocator launch
        // starts putting headers into it.  Alternate two sets of output buffers to have them overlap.
        const hx_batch::PackJob &j = b->pack_job;
        if (j.pending) {
            const unsigned char *o0 = j.d_out, *o1 = j.d_out + (long long) b->S * j.out_stride, *n0 = d_out, *n1 = d_out + (long long) b->S * out_stride;
            const char *b0 = (const char *) j.d_out_bytes, *b1 = b0 + sizeof(int) * (size_t) b->S, *m0 = (const char *) d_out_bytes, *m1 = m0 + sizeof(int) * (size_t) b->S;
            if ((o0 < n1 && n0 < o1) || (b0 < m1 && m0 < b1)) {
                const int js = j.set;
                if (flush_pack(b, -1) != 0) return -1;
                HIPCHK(hipStreamWaitEvent(qa, b->ev_alloc[js], 0));
            }
        }
    }
    AllocArgs a;
    a.st = b->d_st; a.prm = b->d_prm; a.gt = b->d_gt; a.xr = x_xr; a.etab = x_etab; a.thr = x_thr;
    a.msbase = x_msbase; a.bt = x_bt; a.btprev = x_btprev; a.out = d_out; a.out_bytes = d_out_bytes;
    a.dbg = b->debug ? b->d_dbg : nullptr; a.out_stride = out_stride; a.NG = NG; a.S = S; a.status = b->d_status; a.prof = b->d_prof;
    a.packet = b->pk_buf; a.packet_stride = b->pk_stride; a.packet_bytes = b->pk_bytes; a.frame_stats = b->frame_stats;
    a.done_counter = b->d_done;
    a.strict_sums = b->strict_sums;
    a.dur = b->d_dur;
    a.order = nullptr;
    a.park_k = 0;
    if ((S > b->resident && b->lpt) || (b->lpt == 2 && S > b->ncu) || b->lpt == 3) {       // longest first (see hx_batch::lpt)
        LAUNCH(k_order, dim3(1), dim3(1024), qa, (const unsigned *) b->d_dur, b->d_order, S);
        a.order = b->d_order;
        // parking (hx_alloc3.inc): only when every stream has a slot from the launch's start - beyond the resident set a
        // parked slot would keep a waiting stream out - and from the second call on (the order is the previous call's)
        if (S <= b->resident && b->alloc_launches > 0 && !b->alloc1 && !b->lsf) a.park_k = (b->park_k < S / 8 ? b->park_k : S / 8) | (b->park_pair << 16);
    }
    a.x34 = x_x34; a.sgn = x_sgn; a.band = x_band; a.msflag = x_msflag; a.msdec = x_msdec; a.thrprev = x_thrprev;
    a.ixq = x_ixq; a.sgn_w = x_sgn; a.seg = x_seg; a.frm = x_frm; a.slots = x_slots;
    a.pre_len = x_prelen; a.carry_len = x_carrylen;
    b->alloc_launches++;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!b->capturing) {
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, qa));
    }
    // persistent workgroups: as many as the chip holds at once (or one per stream if that is fewer); each walks one stream of
    // the launch order after the other (hx_alloc3.inc)
    // (built into k_alloc_slim, the kernel of batches beyond the resident set; the 256-register kernels keep one workgroup per stream)
    const int G = (b->slim && !b->alloc1 && !b->lsf && k_alloc_slim_persistent() && S > b->resident) ? b->resident : S;
    if (b->alloc1) { if (b->lsf) LAUNCH_LDS(k_alloc1_lsf, dim3(G), dim3(128), K6_LDS(k_alloc1_lsf), qa, a); else LAUNCH_LDS(k_alloc1, dim3(G), dim3(128), K6_LDS(k_alloc1), qa, a); }
    else if (b->lsf) LAUNCH_LDS(k_alloc_lsf, dim3(G), dim3(128), K6_LDS(k_alloc_lsf), qa, a);
    else if (b->slim) LAUNCH_LDS(k_alloc_slim, dim3(G), dim3(128), K6_LDS(k_alloc_slim), qa, a);
    else LAUNCH_LDS(k_alloc, dim3(G), dim3(128), K6_LDS(k_alloc), qa, a);
    if (!b->capturing) {
        HIPCHK(hipEventRecord(e1, qa));
        b->pending.push_back({e0, e1});
    }
    // Every frame of the call packed at once, between the pending frames' images coming out of the stream state and the
    // incomplete ones' going back in.  A plain call (and a host-buffer submit) packs right behind its allocator launch.  A
    // device-buffer submit leaves its packing for later: it is enqueued on a stream of its own behind the NEXT submit's
    // allocator launch and a gate on it, and so runs - like that submit's successor's front end - in that launch's tail
    // instead of between two allocator launches.
    if (pipelined == 1) {
        HIPCHK(hipEventRecord(b->ev_k6[set], qa));
        if (flush_pack(b, (long long) ((unsigned long long) (b->alloc_launches - 1) * (unsigned long long) b->S)) != 0) return -1;   // the previous submit's
        hx_batch::PackJob &j = b->pack_job;
        j.pending = true; j.d_out = d_out; j.out_stride = out_stride; j.d_out_bytes = d_out_bytes; j.nframes = nframes; j.set = set; j.sset = sset;
    } else {
        if (pipelined) HIPCHK(hipEventRecord(b->ev_k6[set], qa));
        // the previous device-buffer submit's packing went out on the packing stream above: its k_pack_carry writes the
        // carried frame images that this call's k_pack_pre reads
        if (flushed_set >= 0) HIPCHK(hipStreamWaitEvent(qa, b->ev_alloc[flushed_set], 0));
        if (enqueue_pack(b, d_out, out_stride, d_out_bytes, nframes, set, sset, qa) != 0) return -1;
        if (pipelined) { HIPCHK(hipEventRecord(b->ev_alloc[set], qa)); HIPCHK(hipEventRecord(b->ev_sgn[sset], qa)); }
    }
    while (!b->capturing && b->pending.size() > 512) {       // a caller that never asks for the timings must not accumulate events
        const auto old = b->pending.front();
        if (hipEventQuery(old.second) != hipSuccess) break;
        float ms = 0;
        if (hipEventElapsedTime(&ms, old.first, old.second) == hipSuccess) { b->alloc_ms_sum += ms; b->alloc_calls++; }
        hipEventDestroy(old.first);
        hipEventDestroy(old.second);
        b->pending.erase(b->pending.begin());
    }
    if (pipelined) {
        b->nsubmit++;
        b->inflight = true;
    }
    HIPCHK(hipGetLastError());
    b->lastNG = NG;
    return 0;
}

extern "C" int hx_batch_encode_s16_device(hx_batch *b, const int16_t *d_pcm, int nframes, unsigned char *d_out,
                                          long long out_stride, int *d_out_bytes, void *stream)
{
    return encode_core(b, d_pcm, nullptr, nframes, d_out, out_stride, d_out_bytes, stream);
}

extern "C" int hx_batch_encode_f32_device(hx_batch *b, const float *d_pcm, int nframes, unsigned char *d_out,
                                          long long out_stride, int *d_out_bytes, void *stream)
{
    return encode_core(b, nullptr, d_pcm, nframes, d_out, out_stride, d_out_bytes, stream);
}

// Pipelined form of the device calls.  A submit returns at once like the plain call, but its output
// (d_out, d_out_bytes) is ordered on the caller's stream only by a later hx_batch_wait (or by the next
// plain call / host-buffer call on the batch).  The PCM must be ready on `stream` at the submit, and
// d_pcm must stay unchanged until the submit's front end has run (hx_batch_wait covers that too).
// Consecutive submits overlap: the front end of call n+1 fills the SIMDs that k_alloc of call n
// leaves idle while its slowest streams finish.
extern "C" int hx_batch_submit_s16_device(hx_batch *b, const int16_t *d_pcm, int nframes, unsigned char *d_out,
                                          long long out_stride, int *d_out_bytes, void *stream)
{
    return encode_core(b, d_pcm, nullptr, nframes, d_out, out_stride, d_out_bytes, stream, 1);
}

extern "C" int hx_batch_submit_f32_device(hx_batch *b, const float *d_pcm, int nframes, unsigned char *d_out,
                                          long long out_stride, int *d_out_bytes, void *stream)
{
    return encode_core(b, nullptr, d_pcm, nframes, d_out, out_stride, d_out_bytes, stream, 1);
}

// share (percent) of the previous allocator launch's resident workgroups that must have started before a submit's front end is released; 0 = no gate
extern "C" void hx_batch_set_gate(hx_batch *b, int percent) { if (b) b->gate_percent = percent < 0 ? 0 : (percent > 100 ? 100 : percent); }

// make `stream` wait for everything submitted so far
static int batch_wait_pass(hx_batch *b, void *stream)
{
    HIPCHK(hipSetDevice(b->device));
    if (flush_pack(b, -1) != 0) return -1;
    const int last = (int) ((b->nsubmit - 1) & 1);
    HIPCHK(hipStreamWaitEvent((hipStream_t) stream, b->ev_front[last], 0));
    HIPCHK(hipStreamWaitEvent((hipStream_t) stream, b->ev_alloc[last], 0));
    return 0;
}
extern "C" int hx_batch_wait(hx_batch *b, void *stream)
{
    if (!b) return -1;
    if (b->poisoned) { set_err("the batch is unusable after a failed device call: destroy it"); return -1; }
    if (!b->inflight) return 0;
    const int r = batch_wait_pass(b, stream);
    if (r != 0) b->poisoned = true;         // (the deferred packing may be half enqueued)
    return r;
}

// ---- pipelined host-buffer calls ----
// The PCM of call n+1 crosses PCIe while call n is encoded, and the bitstream of call n while call
// n+1 is: two sets of device staging buffers, one stream per copy direction, events in between.
// Truly asynchronous only with page-locked host memory (hx_pinned_alloc); with pageable memory the
// copies fall back to staged, mostly synchronous transfers and the result is still correct.
extern "C" void *hx_pinned_alloc(long long bytes)
{
    void *p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t) bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void hx_pinned_free(void *p) { if (p) hipHostFree(p); }

static int submit_host_pass(hx_batch *b, const void *pcm, int is_f32, int nframes, unsigned char *out, long long out_stride, int *out_bytes);
static int submit_host(hx_batch *b, const void *pcm, int is_f32, int nframes, unsigned char *out, long long out_stride, int *out_bytes)
{
    if (check_call(b, pcm, nframes, out, out_stride, out_bytes) != 0) return -1;
    // like encode_core: a HIP call that fails once staging buffers, events or the call counter have been touched leaves the
    // batch half updated; it refuses further calls instead of running on
    const int r = submit_host_pass(b, pcm, is_f32, nframes, out, out_stride, out_bytes);
    if (r != 0) b->poisoned = true;
    return r;
}
static int submit_host_pass(hx_batch *b, const void *pcm, int is_f32, int nframes, unsigned char *out, long long out_stride, int *out_bytes)
{
    HIPCHK(hipSetDevice(b->device));
    const long long pbytes = (long long) b->S * nframes * 1152 * b->nchan * (is_f32 ? sizeof(float) : sizeof(int16_t)), obytes = (long long) b->S * out_stride;
    if (!b->s_h2d) {
        HIPCHK(hipStreamCreateWithFlags(&b->s_h2d, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b->s_d2h, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b->s_host, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIPCHK(hipEventCreateWithFlags(&b->ev_h2d[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b->ev_d2h[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b->ev_hfront[i], hipEventDisableTiming));
            HIPCHK(hipMalloc((void **) &b->hs_nb[i], sizeof(int) * b->S));
        }
    }
    if (pbytes > b->hs_pcm_cap || obytes > b->hs_out_cap) {     // (re)size the staging: drain first
        if (b->s_pack) flush_pack(b, -1);
        HIPCHK(hipDeviceSynchronize());
        for (int i = 0; i < 2; i++) {
            if (pbytes > b->hs_pcm_cap) { if (b->hs_pcm[i]) hipFree(b->hs_pcm[i]); HIPCHK(hipMalloc(&b->hs_pcm[i], (size_t) pbytes)); }
            if (obytes > b->hs_out_cap) { if (b->hs_out[i]) hipFree(b->hs_out[i]); HIPCHK(hipMalloc((void **) &b->hs_out[i], (size_t) obytes)); }
        }
        if (pbytes > b->hs_pcm_cap) b->hs_pcm_cap = pbytes;
        if (obytes > b->hs_out_cap) b->hs_out_cap = obytes;
    }
    const int k = (int) (b->nhost & 1);
    if (b->nhost >= 2) {
        HIPCHK(hipStreamWaitEvent(b->s_h2d, b->ev_hfront[k], 0));   // the front end of call n-2 has read this PCM buffer
        HIPCHK(hipStreamWaitEvent(b->s_host, b->ev_d2h[k], 0));     // the bitstream of call n-2 has left this output buffer
    }
    HIPCHK(hipMemcpyAsync(b->hs_pcm[k], pcm, (size_t) pbytes, hipMemcpyHostToDevice, b->s_h2d));
    HIPCHK(hipEventRecord(b->ev_h2d[k], b->s_h2d));
    HIPCHK(hipStreamWaitEvent(b->s_host, b->ev_h2d[k], 0));
    const int set = (int) (b->nsubmit & 1);
    int r = encode_core(b, is_f32 ? nullptr : (const int16_t *) b->hs_pcm[k], is_f32 ? (const float *) b->hs_pcm[k] : nullptr, nframes,
                        b->hs_out[k], out_stride, b->hs_nb[k], b->s_host, 2);
    if (r) return r;
    HIPCHK(hipEventRecord(b->ev_hfront[k], b->s_front));
    HIPCHK(hipStreamWaitEvent(b->s_d2h, b->ev_alloc[set], 0));
    HIPCHK(hipMemcpyAsync(out_bytes, b->hs_nb[k], sizeof(int) * b->S, hipMemcpyDeviceToHost, b->s_d2h));
    HIPCHK(hipMemcpyAsync(out, b->hs_out[k], (size_t) obytes, hipMemcpyDeviceToHost, b->s_d2h));
    HIPCHK(hipEventRecord(b->ev_d2h[k], b->s_d2h));
    b->nhost++;
    return 0;
}

extern "C" int hx_batch_submit_s16_host(hx_batch *b, const int16_t *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes)
{
    return submit_host(b, pcm, 0, nframes, out, out_stride, out_bytes);
}

extern "C" int hx_batch_submit_f32_host(hx_batch *b, const float *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes)
{
    return submit_host(b, pcm, 1, nframes, out, out_stride, out_bytes);
}

// block until the outputs of every submitted host call are in host memory
extern "C" int hx_batch_wait_host(hx_batch *b)
{
    if (!b) return -1;
    if (b->poisoned) { set_err("the batch is unusable after a failed device call: destroy it"); return -1; }
    if (hipSetDevice(b->device) != hipSuccess || (b->s_d2h && (hipStreamSynchronize(b->s_front) != hipSuccess || hipStreamSynchronize(b->s_d2h) != hipSuccess))) {
        set_err("HIP error while waiting for the host-buffer calls");
        b->poisoned = true;
        return -1;
    }
    return 0;
}

extern "C" float hx_batch_alloc_kernel_ms(hx_batch *b, int *ncalls)
{
    hipSetDevice(b->device);
    for (auto &pr : b->pending) {
        hipEventSynchronize(pr.second);
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { b->alloc_ms_sum += ms; b->alloc_calls++; }
        hipEventDestroy(pr.first);
        hipEventDestroy(pr.second);
    }
    b->pending.clear();
    float mean = b->alloc_calls ? (float) (b->alloc_ms_sum / b->alloc_calls) : 0.0f;
    if (ncalls) *ncalls = b->alloc_calls;
    b->alloc_ms_sum = 0;
    b->alloc_calls = 0;
    return mean;
}

static int encode_host(hx_batch *b, const void *pcm, int is_f32, int nframes, unsigned char *out, long long out_stride, int *out_bytes)
{
    if (check_call(b, pcm, nframes, out, out_stride, out_bytes) != 0) return -1;
    HIPCHK(hipSetDevice(b->device));
    long long pbytes = (long long) b->S * nframes * 1152 * b->nchan * (is_f32 ? sizeof(float) : sizeof(int16_t)), obytes = (long long) b->S * out_stride;
    if (pbytes > b->pcm_cap) { if (b->d_pcm) hipFree(b->d_pcm); HIPCHK(hipMalloc((void **) &b->d_pcm, pbytes)); b->pcm_cap = pbytes; }
    if (obytes > b->out_cap) { if (b->d_out) hipFree(b->d_out); HIPCHK(hipMalloc((void **) &b->d_out, obytes)); b->out_cap = obytes; }
    HIPCHK(hipMemcpy(b->d_pcm, pcm, pbytes, hipMemcpyHostToDevice));
    int r = encode_core(b, is_f32 ? nullptr : (const int16_t *) b->d_pcm, is_f32 ? (const float *) b->d_pcm : nullptr, nframes,
                        b->d_out, out_stride, b->d_outbytes, nullptr);
    if (r) return r;
    if (b->s_pack) flush_pack(b, -1);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_bytes, b->d_outbytes, sizeof(int) * b->S, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out, b->d_out, obytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int hx_batch_encode_s16_host(hx_batch *b, const int16_t *pcm, int nframes, unsigned char *out,
                                        long long out_stride, int *out_bytes)
{
    return encode_host(b, pcm, 0, nframes, out, out_stride, out_bytes);
}

// host-buffer call that also returns the per-frame counters (see hx_batch_frame_stats_buffer)
extern "C" int hx_batch_encode_f32_host_stats(hx_batch *b, const float *pcm, int nframes, unsigned char *out,
                                              long long out_stride, int *out_bytes, int *stats)
{
    if (!stats) { set_err("null buffer"); return -1; }
    if (check_call(b, pcm, nframes, out, out_stride, out_bytes) != 0) return -1;
    HIPCHK(hipSetDevice(b->device));
    int *d_stats = nullptr;
    const size_t n = sizeof(int) * (size_t) b->S * nframes * 2;
    HIPCHK(hipMalloc((void **) &d_stats, n));
    int *saved = b->frame_stats;
    b->frame_stats = d_stats;
    int r = encode_host(b, pcm, 1, nframes, out, out_stride, out_bytes);
    b->frame_stats = saved;
    if (r == 0 && hipMemcpy(stats, d_stats, n, hipMemcpyDeviceToHost) != hipSuccess) r = -1;
    hipFree(d_stats);
    return r;
}

// what CMp3Enc::L3_audio_encode_info_ec / _info_head would report for a control, without creating an
// encoder (host only).  Returns 0 if the configuration is rejected.
extern "C" int hx_control_info(const HX_E_CONTROL *ec, HX_E_CONTROL *ec_out, HX_MPEG_HEAD *head_out)
{
    HxParams p;
    if (!hx_resolve((const HxControl *) ec, &p)) return 0;
    if (ec_out) memcpy(ec_out, &p.ec, sizeof(HxControl));
    if (head_out) memcpy(head_out, &p.head_info, sizeof(HxMpegHead));
    return 1;
}

extern "C" int hx_batch_encode_f32_host(hx_batch *b, const float *pcm, int nframes, unsigned char *out,
                                        long long out_stride, int *out_bytes)
{
    return encode_host(b, pcm, 1, nframes, out, out_stride, out_bytes);
}

extern "C" int hx_batch_status(hx_batch *b)
{
    int v = -1;
    if (!b) return -1;
    if (b->poisoned) return -1;
    hipSetDevice(b->device);
    if (b->s_pack && flush_pack(b, -1) != 0) { b->poisoned = true; return -1; }     // (the last device-buffer submit's packing: it writes that submit's output buffers)
    hipDeviceSynchronize();
    // (a gate that gave up waiting costs overlap, not correctness: it is counted in hx_batch_gate_timeouts, not here)
    if (hipMemcpy(&v, b->d_status, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

// which build of the stream walk the batch runs: 0 = k_alloc (or the MPEG-2 / first-generation kernels), 1 = k_alloc_slim;
// streams resident at once on the device
extern "C" int hx_batch_k6_variant(const hx_batch *b) { return b ? b->slim : -1; }
extern "C" int hx_batch_resident_streams(const hx_batch *b) { return b ? b->resident : -1; }

// submits whose front end started late because its gate gave up waiting (see hx_batch_set_gate); synchronises
extern "C" int hx_batch_gate_timeouts(hx_batch *b)
{
    int gate[2] = {0, 0};
    if (!b) return -1;
    hipSetDevice(b->device);
    if (b->s_pack) flush_pack(b, -1);
    hipDeviceSynchronize();
    if (hipMemcpy(gate, b->d_done, sizeof(gate), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return gate[1];
}

extern "C" HX_INT_PAIR hx_batch_frames_bytes(hx_batch *b, int i)
{
    HX_INT_PAIR r = {0, 0};
    if (!b || i < 0 || i >= b->S) return r;
    hipSetDevice(b->device);
    if (b->s_pack) flush_pack(b, -1);
    hipDeviceSynchronize();
    unsigned v[2];
    hipMemcpy(&v[0], (char *) (b->d_st + i) + offsetof(HxStream, tot_frames_out), 4, hipMemcpyDeviceToHost);
    hipMemcpy(&v[1], (char *) (b->d_st + i) + offsetof(HxStream, tot_bytes_out), 4, hipMemcpyDeviceToHost);
    r.a = (int) v[0]; r.b = (int) v[1];
    return r;
}

extern "C" long long hx_batch_debug_read(hx_batch *b, const char *name, void *dst, long long cap)
{
    if (!b || !name || !dst) return -1;
    hipSetDevice(b->device);
    if (b->s_pack) flush_pack(b, -1);
    hipDeviceSynchronize();
    const long long S = b->S, NG = b->lastNG;
    const void *src = nullptr;
    long long n = 0;
    std::string k(name);
    if (k == "sb") { src = b->d_sb; n = sizeof(float) * S * 2 * (2LL * b->maxF + 3) * 576; }
    else if (k == "xr") { src = b->d_xr; n = sizeof(float) * S * NG * 1152; }       // the spectrum
    else if (k == "xmag") { src = b->d_xrdbg; n = sizeof(float) * S * NG * 1152; }  // the magnitudes k_prep works on (written in debug mode only)
    else if (k == "x34") { src = b->d_x34; n = sizeof(float) * S * NG * 1152; }
    else if (k == "band") { src = b->d_band; n = sizeof(HxBandPrep) * S * NG; }
    else if (k == "msflag") { src = b->d_msflag; n = S * NG; }
    else if (k == "ixq") { src = b->d_ixq; n = sizeof(short) * S * NG * 1152; }
    else if (k == "sgn") { src = b->d_sgn; n = (long long) sizeof(unsigned) * S * NG * 2 * HX_SGN_WORDS; }      // one bit per line, HX_SGN_WORDS words per (granule, channel)
    else if (k == "seg") { src = b->d_seg; n = sizeof(HxSegOut) * S * NG * 2; }
    else if (k == "frm") { src = b->d_frm; n = sizeof(HxFrameOut) * S * NG; }
    else if (k == "etab") { src = b->d_etab; n = sizeof(float) * S * NG * 128; }
    else if (k == "thr") { src = b->d_thr; n = sizeof(float) * S * NG * 128; }
    else if (k == "msbase") { src = b->d_msbase; n = sizeof(int) * S * NG; }
    else if (k == "place") { src = b->d_dur + S; n = sizeof(unsigned) * S; }       // per WORKGROUP of the last allocator launch (launch order): XCC id << 16 | HW_ID[15:0] (CU [11:8], SH [12], SE [15:13])
    else if (k == "dur") { src = b->d_dur; n = sizeof(unsigned) * S; }              // the last allocator launch's per-stream durations, 100 MHz ticks
    else if (k == "big_sweeps") { src = b->d_done + 3; n = sizeof(int); }        // gain-search line passes that took the double x^(4/3) table
    else if (k == "strict_sums") { src = b->d_done + 4; n = sizeof(int); }       // certified band sums that fell back to the strict line-order sum
    else if (k == "bt") { src = b->d_bt; n = S * NG; }
    else if (k == "eng") { src = b->d_eng; n = sizeof(int) * S * 2 * NG * 9; }
    else if (k == "dbg" && b->d_dbg) { src = b->d_dbg; n = sizeof(HxFrameDebug) * S * (NG / 2); }
    else if (k == "prof" && b->d_prof) { src = b->d_prof; n = sizeof(unsigned long long) * S * 64; }
    else if (k == "state") { src = b->d_st; n = sizeof(HxStream) * S; }
    else if (k == "attack" && b->d_dbgmetric) { src = b->d_dbgmetric; n = sizeof(int) * S * NG * 2; }
    if (!src) return -1;
    if (n > cap) n = cap;
    hipMemcpy(dst, src, (size_t) n, hipMemcpyDeviceToHost);
    return n;
}

// host-side table generation exposed for the CPU tests (no GPU needed): resolves `ec` and copies
// the named table; returns bytes copied, 0 if the configuration is rejected, -1 for a bad name
extern "C" const char *hx_libc_version(void) { return hx_host_libc_version(); }
extern "C" int hx_libm_spot_check(int n) { return hx_libm32_spot_check(n); }

// 1 if the tables resolved for this control have the structure k_alloc_slim derives them from (hx_host.cpp: hx_slim_tables_ok)
extern "C" int hx_debug_slim_tables_ok(const HX_E_CONTROL *ec)
{
    static HxParams p;
    static HxGlobalTabs g;
    if (!hx_resolve((const HxControl *) ec, &p)) return -1;
    hx_global_tabs(&g);
    return hx_slim_tables_ok(&p, &g);
}

extern "C" long long hx_debug_host_table(const HX_E_CONTROL *ec, const char *name, void *dst, long long cap)
{
    static HxParams p;
    static HxGlobalTabs g;
    if (!hx_resolve((const HxControl *) ec, &p)) return 0;
    hx_global_tabs(&g);
    std::string k(name);
    const void *src = nullptr;
    long long n = 0;
#define TAB(nm, obj) else if (k == nm) { src = &(obj); n = sizeof(obj); }
    if (k == "psy_w") { src = p.psyL.w; n = sizeof(p.psyL.w); }
    TAB("psy_cnt", p.psyL.cnt) TAB("psy_off", p.psyL.off) TAB("psy_nsum", p.psyL.nsum) TAB("psy_npart", p.psyL.npart)
    TAB("dct_tw", p.dct_tw) TAB("win", p.win) TAB("csa", p.csa) TAB("mdct_pre18", p.mdct_pre18) TAB("mdct_odd18", p.mdct_odd18)
    TAB("dct9_even", p.dct9_even) TAB("dct9_odd", p.dct9_odd) TAB("dct9_k3", p.dct9_k3) TAB("mdct_pre6", p.mdct_pre6) TAB("mdct_odd6", p.mdct_odd6) TAB("dct3_k", p.dct3_k)
    TAB("look_gain", p.look_gain) TAB("look_34igain", p.look_34igain) TAB("look_ix43", p.look_ix43)
    TAB("look_log_cbwmb", p.look_log_cbwmb) TAB("nBand_l", p.nBand_l) TAB("startBand_l", p.startBand_l)
    TAB("nsf", p.nsf) TAB("taperNT", p.taperNT) TAB("head", p.head) TAB("ec", p.ec)
    TAB("anwin", g.anwin) TAB("mblog", g.mblog) TAB("mbexp_lo", g.mbexp_lo) TAB("mbexp_hi", g.mbexp_hi)
    TAB("pow34_exp", g.pow34_exp) TAB("quant_off", g.quant_off) TAB("logsub", g.logsub)
    TAB("huff_code", g.huff_code) TAB("huff_len", g.huff_len)
    TAB("lane_run", p.lane_run) TAB("band_last_lane", p.band_last_lane)
    TAB("nchan", p.nchan)
#undef TAB
    if (k == "run_w") { static int v[1]; v[0] = p.run_w; src = v; n = sizeof(v); }
    if (k == "scalars") {
        static int v[16];
        v[0] = p.nsb_limit; v[1] = p.nsb_ms0; v[2] = p.band_limit; v[3] = p.main_framebytes; v[4] = p.AveTargetBits;
        v[5] = p.initialMNR; v[6] = p.ms_flag; v[7] = p.hf_flag; v[8] = p.vbr_flag; v[9] = p.framebytes;
        v[10] = p.remainder; v[11] = p.ivbr_max; v[12] = p.vbr_pool_target; v[13] = p.samprate; v[14] = p.totbitrate; v[15] = p.nsb_ms1;
        src = v; n = sizeof(v);
    }
    if (!src) return -1;
    if (n > cap) n = cap;
    memcpy(dst, src, (size_t) n);
    return n;
}

// ------------------------------------------------------------------------------------------
// CMp3Enc-compatible single-stream encoder: a batch of one, one frame per call.
struct hx_enc {
    int device = 0;
    hx_batch *b = nullptr;
    HxParams p;
    int src_bits = 0, src_float = 0;
    int src_chan = 2;                   // channels of the caller's PCM (2 with mono_convert: down-mixed to one)
    std::vector<int16_t> pcm16;
    std::vector<unsigned char> outbuf;
    unsigned frames = 0, bytes = 0;
    int ave = 0;
    hx_src *src = nullptr;              // converter of the MP3_audio_encode entry points
    unsigned char *d_packet = nullptr;  // one reformatted frame (device), allocated on first *_Packet call
    int *d_packet_bytes = nullptr;
    // One call = one graph launch: the whole single-stream chain (PCM up, the pipeline's kernels, byte count / frame counter /
    // bitstream down) is recorded once into a HIP graph over page-locked staging buffers and replayed per call
    // (reference call being replaced: CMp3Enc::L3_audio_encode, mp3enc.cpp:2031-2073, and MP3_audio_encode, :2812-2866).
    hipStream_t gq = nullptr;
    unsigned char *d_encbuf = nullptr;  // device: [byte count | frame counter | ... HX_ENC_GRAPH_OFF | the call's bitstream]
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    float *h_pcm = nullptr;             // page-locked: one 1152-sample block, float at int16 scale
    unsigned char *h_out = nullptr;     // page-locked, coherent: [byte count | frame counter | sequence word | ... 256 | the call's bitstream], written by the packing workgroup
    int graph_state = 0;                // 0 = not built yet, 1 = ready, -1 = not available (disabled, or the build failed: plain calls)
    bool spin_off = false;              // HMP3AMD_ENC_GRAPH=2: always wait with hipStreamSynchronize (A/B of the wait)
    int plain_calls = 0;                // calls made the plain way since init (the first ones: they also load the kernels' code objects)
};
#define HX_ENC_GRAPH_OFF 256

static void enc_graph_drop(hx_enc *e)
{
    if (e->gexec) { hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
    if (e->graph) { hipGraphDestroy(e->graph); e->graph = nullptr; }
    if (e->gq) { hipStreamDestroy(e->gq); e->gq = nullptr; }
    if (e->d_encbuf) { hipFree(e->d_encbuf); e->d_encbuf = nullptr; }
    if (e->h_pcm) { hipHostFree(e->h_pcm); e->h_pcm = nullptr; }
    if (e->h_out) { hipHostFree(e->h_out); e->h_out = nullptr; }
    e->graph_state = 0;
    e->plain_calls = 0;
}

extern "C" hx_enc *hx_enc_create(int device)
{
    hx_enc *e = new hx_enc;
    e->device = device;
    return e;
}

extern "C" void hx_enc_destroy(hx_enc *e)
{
    if (!e) return;
    enc_graph_drop(e);
    if (e->d_packet) hipFree(e->d_packet);
    if (e->d_packet_bytes) hipFree(e->d_packet_bytes);
    hx_src_destroy(e->src);
    hx_batch_destroy(e->b);
    delete e;
}

extern "C" int hx_enc_L3_audio_encode_init(hx_enc *e, const HX_E_CONTROL *ec)
{
    enc_graph_drop(e);
    if (e->b) { hx_batch_destroy(e->b); e->b = nullptr; }       // re-init is legal (mp3enc.cpp:267-272)
    int r = hx_resolve((const HxControl *) ec, &e->p);
    if (!r) { if (*hx_resolve_error()) set_err("configuration rejected: %s", hx_resolve_error()); else set_err("configuration rejected"); return 0; }
    e->b = hx_batch_create(e->device, 1, ec, 1, 1);
    if (!e->b) return 0;
    e->frames = e->bytes = 0; e->ave = 0;
    e->pcm16.assign(2304, 0);
    e->outbuf.assign((size_t) hx_batch_out_stride(e->b, 1) + 65536, 0);
    e->src_bits = 0;
    return r;
}

// Record the single-stream chain of e->b into a graph (see hx_enc).  Returns 0 when e->gexec is ready.
// The graph is the PCM's copy up from page-locked staging and the pipeline's kernels; the call's results - byte count, frame
// counter, bitstream - are written to page-locked host memory by the packing workgroup itself, which publishes a sequence word
// behind system-scope fences (hx_pack.hip, k_pack solo): the call polls that word.
static int enc_graph_build(hx_enc *e)
{
    hx_batch *b = e->b;
    HIPCHK(hipSetDevice(b->device));
    const long long stride = (long long) e->outbuf.size();
    const long long pbytes = 1152LL * b->nchan * (long long) sizeof(float);
    // allocated before the recording starts (no allocation inside one)
    if (pbytes > b->pcm_cap) { if (b->d_pcm) hipFree(b->d_pcm); HIPCHK(hipMalloc((void **) &b->d_pcm, pbytes)); b->pcm_cap = pbytes; }
    HIPCHK(hipMalloc((void **) &e->d_encbuf, (size_t) (HX_ENC_GRAPH_OFF + stride)));
    HIPCHK(hipMemset(e->d_encbuf, 0, (size_t) (HX_ENC_GRAPH_OFF + stride)));
    HIPCHK(hipHostMalloc((void **) &e->h_pcm, (size_t) pbytes, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &e->h_out, (size_t) (HX_ENC_GRAPH_OFF + stride + 16), hipHostMallocCoherent));
    memset(e->h_out, 0, (size_t) (HX_ENC_GRAPH_OFF + stride + 16));
    HIPCHK(hipMemcpy(e->h_out + 8, b->d_done + 2, sizeof(int), hipMemcpyDeviceToHost));       // the sequence word as the device has it now
    HIPCHK(hipStreamCreateWithFlags(&e->gq, hipStreamNonBlocking));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipStreamBeginCapture(e->gq, hipStreamCaptureModeThreadLocal));
    int r = 0;
    b->capturing = true;
    b->cap_frames = reinterpret_cast<unsigned *>(e->d_encbuf) + 1;
    b->cap_host = e->h_out;
    if (hipMemcpyAsync(b->d_pcm, e->h_pcm, (size_t) pbytes, hipMemcpyHostToDevice, e->gq) != hipSuccess) r = -1;
    if (!r) r = encode_core(b, nullptr, (const float *) b->d_pcm, 1, e->d_encbuf + HX_ENC_GRAPH_OFF, stride, reinterpret_cast<int *>(e->d_encbuf), e->gq);
    b->capturing = false;
    b->cap_frames = nullptr;
    b->cap_host = nullptr;
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(e->gq, &g);       // (always ended, also after a failure inside)
    if (r || ce != hipSuccess || !g) { if (g) hipGraphDestroy(g); (void) hipGetLastError(); set_err("recording the single-stream graph failed"); return -1; }
    e->graph = g;
    if (hipGraphInstantiate(&e->gexec, e->graph, nullptr, nullptr, 0) != hipSuccess) { (void) hipGetLastError(); set_err("hipGraphInstantiate failed"); return -1; }
    // (the recording has executed nothing; the counters the pass advanced on the host - launches, the carry's layout - are
    // the ones a real pass leaves behind, and the batch was not poisoned)
    b->poisoned = false;
    return 0;
}

static HX_IN_OUT encode_one(hx_enc *e, const void *pcm, int is_f32, unsigned char *bs_out, int in_bytes)
{
    HX_IN_OUT x = {in_bytes, 0};
    int nb = 0;
    hx_batch *b = e->b;
    // The graph replays a call with exactly the arguments it was recorded with: anything optional (packets, per-frame
    // counters, debug taps: all set per call by the entry points that need them) goes the plain way, and so do the first
    // two calls (which load the kernels) and int16 input (the CMp3Enc entry points hand over float).
    const bool plain = !is_f32 || b->debug || b->pk_buf || b->frame_stats || b->poisoned || b->inflight || e->graph_state < 0 || e->plain_calls < 2;
    if (!plain && e->graph_state == 0) {
        const char *env = getenv("HMP3AMD_ENC_GRAPH");
        e->spin_off = env && atoi(env) == 2;
        if (env && atoi(env) == 0) e->graph_state = -1;
        else if (enc_graph_build(e) == 0) e->graph_state = 1;
        else {      // plain calls from here on (the staging stays allocated until the encoder is re-initialised or destroyed)
            if (e->gexec) { hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
            e->graph_state = -1;
        }
    }
    if (!plain && e->graph_state == 1) {
        memcpy(e->h_pcm, pcm, (size_t) 1152 * b->nchan * sizeof(float));
        const volatile int *seq = reinterpret_cast<const volatile int *>(e->h_out) + 2;
        const int before = *seq;
        bool ok = hipGraphLaunch(e->gexec, e->gq) == hipSuccess;
        if (ok) {
            // wait on the sequence word the packing workgroup publishes behind its results (a few microseconds sooner than the
            // runtime's own wait); after 2 ms - a descheduled process, a contended device - leave the waiting to the runtime,
            // behind which the kernel's writes are complete as well
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while (!e->spin_off && *seq == before) {
                __builtin_ia32_pause();
                if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            if (*seq == before || e->spin_off) ok = hipStreamSynchronize(e->gq) == hipSuccess;
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (ok) {
            nb = *reinterpret_cast<const volatile int *>(e->h_out);
            if (nb < 0 || nb > (int) e->outbuf.size()) nb = 0;      // (cannot happen: the byte count is bounded by the stride)
            memcpy(bs_out, e->h_out + HX_ENC_GRAPH_OFF, (size_t) nb);
            x.out_bytes = nb;
            e->bytes += nb;
            e->ave = e->ave + ((((nb << 8) - e->ave)) >> (e->p.h_id ? 7 : 6));    // mp3enc.cpp:2328 / :2589
            e->frames = reinterpret_cast<const volatile unsigned *>(e->h_out)[1];
        } else {
            (void) hipGetLastError();
            set_err("replaying the single-stream graph failed");
            b->poisoned = true;
        }
        return x;
    }
    long long stride = (long long) e->outbuf.size();
    if (encode_host(e->b, pcm, is_f32, 1, e->outbuf.data(), stride, &nb) == 0) {
        memcpy(bs_out, e->outbuf.data(), nb);
        x.out_bytes = nb;
        e->bytes += nb;
        e->ave = e->ave + ((((nb << 8) - e->ave)) >> (e->p.h_id ? 7 : 6));    // mp3enc.cpp:2328 / :2589
        e->frames = (unsigned) hx_batch_frames_bytes(e->b, 0).a;
        e->plain_calls++;
    }
    return x;
}

extern "C" HX_IN_OUT hx_enc_L3_audio_encode(hx_enc *e, const float *pcm, unsigned char *bs_out)
{
    // float at int16 scale (pub/mp3enc.h:90-98), taken as is: the polyphase kernel reads fp32
    return encode_one(e, pcm, 1, bs_out, 4608 * e->p.nchan);
}

// CMp3Enc::L3_audio_encode_Packet / MP3_audio_encode_Packet (pub/mp3enc.h:110-131): the normal
// bitstream in bs_out (may be NULL) plus this call's frame as a self-contained packet
// (nbytes_out[0] bytes, nbytes_out[1] = 0; at the MPEG-2 rates two packets, nbytes_out[0] then
// nbytes_out[1] bytes); packet may be NULL.
extern "C" HX_IN_OUT hx_enc_MP3_audio_encode(hx_enc *e, const unsigned char *pcm, unsigned char *bs_out);
static HX_IN_OUT encode_packet(hx_enc *e, const void *pcm, int mp3_entry, unsigned char *bs_out, unsigned char *packet, int nbytes_out[2])
{
    std::vector<unsigned char> scratch;
    if (!bs_out) { scratch.resize(e->outbuf.size()); bs_out = scratch.data(); }
    if (packet) {
        hipSetDevice(e->device);
        if (!e->d_packet) { hipMalloc((void **) &e->d_packet, 4096); hipMalloc((void **) &e->d_packet_bytes, 2 * sizeof(int)); }
        hx_batch_packet_buffers(e->b, e->d_packet, 4096, e->d_packet_bytes);
    }
    HX_IN_OUT x = mp3_entry ? hx_enc_MP3_audio_encode(e, (const unsigned char *) pcm, bs_out) : hx_enc_L3_audio_encode(e, (const float *) pcm, bs_out);
    if (packet) {
        int n[2] = {0, 0};      // an MPEG-2 call returns two single-granule packets back to back (mp3enc.cpp:3363)
        hipMemcpy(n, e->d_packet_bytes, 2 * sizeof(int), hipMemcpyDeviceToHost);
        hipMemcpy(packet, e->d_packet, (size_t) (n[0] + n[1]), hipMemcpyDeviceToHost);
        nbytes_out[0] = n[0];
        nbytes_out[1] = n[1];
        hx_batch_packet_buffers(e->b, nullptr, 0, nullptr);
    }
    return x;
}

extern "C" HX_IN_OUT hx_enc_L3_audio_encode_Packet(hx_enc *e, const float *pcm, unsigned char *bs_out, unsigned char *packet, int nbytes_out[2])
{
    return encode_packet(e, pcm, 0, bs_out, packet, nbytes_out);
}

extern "C" HX_IN_OUT hx_enc_MP3_audio_encode_Packet(hx_enc *e, const unsigned char *pcm, unsigned char *bs_out, unsigned char *packet, int nbytes_out[2])
{
    return encode_packet(e, pcm, 1, bs_out, packet, nbytes_out);
}

static int nearest_rate(const int *table, int n, int x)
{
    int best = table[0], d0 = abs(table[0] - x);
    for (int i = 0; i < n; i++) { const int d = abs(table[i] - x); if (d < d0) { d0 = d; best = table[i]; } }
    return best;
}

// CMp3Enc::MP3_audio_encode_init (reference mp3enc.cpp:2655-2808): pick the encode rate for the source
// rate and mpeg_select (0 track the input, 1 an MPEG-1 rate, 2 an MPEG-2 rate, else that rate), set up the
// sample-format / rate converter (hx_src.cpp) and the encoder behind it.  Returns the bytes the caller
// must hold before every hx_enc_MP3_audio_encode call (more than one call consumes: 1153 sample frames
// when the rates are equal), 0 on failure.
extern "C" int hx_enc_MP3_audio_encode_init(hx_enc *e, const HX_E_CONTROL *ec, int source_bits, int source_is_float,
                                            int mpeg_select, int mono_convert)
{
    static const int rate_table[6] = {22050, 24000, 16000, 44100, 48000, 32000};
    const int source = ec->samprate;
    if (source < 4000 || source > 48000) { set_err("source sample rate out of range"); return 0; }
    const int source_chan = (ec->mode == 3) ? 1 : 2;            // the source is mono iff ec->mode == 3
    const int target_chan = mono_convert ? 1 : source_chan;
    int target = 0;
    if (mpeg_select < 0) mpeg_select = 0;
    switch (mpeg_select) {
    case 0:
        if (source < 16000) { target = nearest_rate(rate_table, 3, 2 * source); if (target == 2 * source) break; }
        target = nearest_rate(rate_table, 6, source);
        break;
    case 1:
        if (source < 16000) { target = nearest_rate(rate_table + 3, 3, 4 * source); if (target == 4 * source) break; }
        if (source < 32000) { target = nearest_rate(rate_table + 3, 3, 2 * source); if (target == 2 * source) break; }
        target = nearest_rate(rate_table + 3, 3, source);
        break;
    case 2:
        if (source < 16000) { target = nearest_rate(rate_table, 3, 2 * source); if (target == 2 * source) break; }
        if (source > 24000) { target = nearest_rate(rate_table, 3, source / 2); if (2 * target == source) break; }
        target = nearest_rate(rate_table, 3, source);
        break;
    default:
        target = nearest_rate(rate_table, 6, mpeg_select);
        if (target != mpeg_select) { set_err("mpeg_select is not an MPEG sample rate"); return 0; }
        break;
    }
    if (!e->src) e->src = hx_src_create();
    int cutoff = 0;
    const int min_input_bytes = hx_src_init(e->src, source, source_chan, source_bits, source_is_float, target, target_chan, &cutoff);
    if (min_input_bytes <= 0) { set_err("the sample-rate converter cannot handle this source format / rate pair"); return 0; }
    int nsb_limit = (64 * cutoff + target / 2) / target;
    if (nsb_limit > 30) nsb_limit = 30;
    HX_E_CONTROL ec2 = *ec;
    ec2.samprate = target;
    if (target_chan == 1) ec2.mode = 3;
    if (source < target) {          // up-sampled input has nothing above the source's band
        if (ec2.nsb_limit <= 0) ec2.nsb_limit = 30;
        if (ec2.nsb_limit > nsb_limit) ec2.nsb_limit = nsb_limit;
    }
    ec2.layer = 3;
    if (!hx_enc_L3_audio_encode_init(e, &ec2)) return 0;
    e->src_bits = source_bits;
    e->src_float = source_is_float;
    e->src_chan = source_chan;
    return min_input_bytes;
}

// CMp3Enc::MP3_audio_encode (mp3enc.cpp:2812-2828): convert, then encode; in_bytes is what the converter used.
// pcm must hold the bytes hx_enc_MP3_audio_encode_init returned, and be readable for
// 1152 * (source rate / encode rate + 1) sample frames (the converter stages that many).
extern "C" HX_IN_OUT hx_enc_MP3_audio_encode(hx_enc *e, const unsigned char *pcm, unsigned char *bs_out)
{
    float t[2304];
    const int in_bytes = hx_src_convert(e->src, pcm, t, nullptr);
    HX_IN_OUT x = hx_enc_L3_audio_encode(e, t, bs_out);
    x.in_bytes = in_bytes;
    return x;
}

extern "C" void hx_enc_out_stats(hx_enc *e)
{
    int calls = 0;
    if (e && e->b) {
        hipSetDevice(e->b->device);
        hipDeviceSynchronize();
        hipMemcpy(&calls, (char *) e->b->d_st + offsetof(HxStream, call_count), sizeof(int), hipMemcpyDeviceToHost);
    }
    fprintf(stderr, "\n ba long  %6d %6d %6d %6d %6d %6d %6d %6d %6d", calls, 0, 0, 0, 0, 0, 0, 0, 0);
}

extern "C" unsigned hx_enc_get_frames(hx_enc *e) { return e->frames; }
extern "C" HX_INT_PAIR hx_enc_get_frames_bytes(hx_enc *e) { HX_INT_PAIR r = {(int) e->frames, (int) e->bytes}; return r; }
extern "C" float hx_enc_get_bitrate_float(hx_enc *e)
{
    if (e->frames <= 0) return 0.0f;
    const float samples = e->p.h_id ? 1152.0f : 576.0f;        // per frame: MPEG-1 / MPEG-2 (mp3enc.cpp:3456-3462)
    return ((0.001f * 8.0f) * e->bytes * e->p.samprate / (samples * e->frames));
}
extern "C" int hx_enc_get_bitrate(hx_enc *e) { return (int) (hx_enc_get_bitrate_float(e) + 0.5f); }
extern "C" float hx_enc_get_bitrate2_float(hx_enc *e)
{
    if (e->frames <= 0) return 0.0f;
    return (float) ((0.001f * 8.0f / (1152.0 * 256.0)) * e->ave * e->p.samprate);
}
extern "C" void hx_enc_info_ec(hx_enc *e, HX_E_CONTROL *ec) { memcpy(ec, &e->p.ec, sizeof(HxControl)); }
extern "C" void hx_enc_info_head(hx_enc *e, HX_MPEG_HEAD *h) { memcpy(h, &e->p.head_info, sizeof(HxMpegHead)); }
extern "C" void hx_enc_info_string(hx_enc *e, char *s)
{
    static const char *mode_msg[4] = {"stereo", "joint stereo", "dual", "mono"};
    const HxControl *ec = &e->p.ec;
    s += sprintf(s, "Layer III   %s ", mode_msg[e->p.h_mode & 3]);
    s += sprintf(s, "  %ldHz ", (long) e->p.samprate);
    if (ec->vbr_flag == 0) s += sprintf(s, "  %dkbps ", e->p.totbitrate);
    else {
        s += sprintf(s, " VBR-%d", ec->vbr_mnr);
        if (ec->vbr_delta_mnr) s += sprintf(s, "(%d)", ec->vbr_delta_mnr);
    }
    if (ec->hf_flag) { s += sprintf(s, "  hf"); if (ec->hf_flag & 2) s += sprintf(s, "2"); }
}

// ------------------------------------------------------------------------------------------
// Several GPUs of one node behind one handle: the streams are split into contiguous
// blocks, one hx_batch per device, and every call runs one host thread per device on its block of the
// caller's buffers.  Streams are independent, so nothing is exchanged between the devices.
#include <thread>
#include <sched.h>
#include <unistd.h>

// ---- host placement: the NUMA node of a device, and threads / page-locked buffers next to it ----
// A host-fed GPU takes 49 GB/s of PCM over PCIe (bench.py host_fed); eight of them read 394 GB/s of host memory.  That only
// works out of the memory of the socket the GPU hangs on: a rank (or a dispatcher thread) binds itself to the CPUs of its
// device's NUMA node before it allocates its page-locked buffers (first touch puts the pages there) and stays there for its
// copies' submission.  Everything here is best effort: no sysfs entry, one node, or a CPU set that the cgroup does not allow
// leaves the thread where it was and reports -1 / 0.
static int read_int_file(const char *path, int *v)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    const int ok = fscanf(f, "%d", v) == 1;
    fclose(f);
    return ok ? 0 : -1;
}

// NUMA node of HIP device `device` (-1: unknown / not a NUMA machine), from its PCI address in sysfs
extern "C" int hx_device_numa_node(int device)
{
    char bus[64] = {0}, path[256];
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { (void) hipGetLastError(); return -1; }
    if (hipDeviceGetPCIBusId(bus, (int) sizeof(bus), device) != hipSuccess) { (void) hipGetLastError(); return -1; }     // (the error is not left behind for the next launch check)
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char) (*c - 'A' + 'a');      // sysfs spells the address in lower case
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (read_int_file(path, &node) != 0) return -1;
    return node;
}

// The CPUs this process may use, captured once when the library is loaded (= before any thread was bound by it): a thread
// that was bound to one device's node must be able to move to another device's node afterwards, so a node's CPU list is
// intersected with this set, not with the calling thread's current mask.
static cpu_set_t g_proc_cpus;
static bool g_proc_cpus_ok = false;
__attribute__((constructor)) static void capture_process_cpus()
{
    CPU_ZERO(&g_proc_cpus);
    g_proc_cpus_ok = sched_getaffinity(0, sizeof(g_proc_cpus), &g_proc_cpus) == 0;
}
// A process whose CPU set changes after the library was loaded (a launcher that calls sched_setaffinity / taskset on the
// running process, a cpuset change; in Python the library loads lazily, so "when it was loaded" depends on import order)
// takes the set again from its main thread's current mask: returns the number of CPUs, 0 on failure.  The bind calls also
// do this once by themselves when the kernel refuses the mask they computed (EINVAL: none of its CPUs is allowed any more).
extern "C" int hx_refresh_process_cpus(void)
{
    cpu_set_t now;
    CPU_ZERO(&now);
    if (sched_getaffinity(getpid(), sizeof(now), &now) != 0) return 0;      // (pid = the main thread's id)
    g_proc_cpus = now;
    g_proc_cpus_ok = true;
    return CPU_COUNT(&now);
}

// the CPUs of a node that this process may use: parses /sys/devices/system/node/node<N>/cpulist ("0-15,128-143")
static int node_cpus_allowed(int node, cpu_set_t *out)
{
    char path[128], buf[4096];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const bool got = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!got || !g_proc_cpus_ok) return 0;
    CPU_ZERO(out);
    int n = 0;
    for (char *p = buf; *p;) {
        char *e;
        long a = strtol(p, &e, 10), z = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; z = strtol(p, &e, 10); }
        for (long c = a; c <= z && c < CPU_SETSIZE; c++) if (CPU_ISSET((int) c, &g_proc_cpus)) { CPU_SET((int) c, out); n++; }
        p = (*e == ',') ? e + 1 : e;
        if (*e != ',') break;
    }
    return n;
}

// bind the calling thread to the allowed CPUs of NUMA node `node`; returns how many CPUs that is (0: left as it was)
extern "C" int hx_bind_thread_to_node(int node)
{
    if (node < 0) return 0;
    cpu_set_t set;
    int n = node_cpus_allowed(node, &set);
    if (n > 0 && sched_setaffinity(0, sizeof(set), &set) == 0) return n;
    // the process's CPU set may have been narrowed since it was captured: take it again, try once more
    if (hx_refresh_process_cpus() <= 0) return 0;
    n = node_cpus_allowed(node, &set);
    if (n <= 0) return 0;
    return sched_setaffinity(0, sizeof(set), &set) == 0 ? n : 0;
}

// bind the calling thread to the allowed CPUs of `device`'s NUMA node; returns how many CPUs that is (0: left as it was)
extern "C" int hx_bind_thread_to_device(int device)
{
    return hx_bind_thread_to_node(hx_device_numa_node(device));
}

struct hx_multi {
    std::vector<hx_batch *> part;
    std::vector<int> first, count, device;
    std::vector<cpu_set_t> cpus;        // per device: the CPUs of its NUMA node this process may use (resolved once, at creation)
    std::vector<int> ncpus;             // ... and how many (0: unknown, the device's thread stays where it is)
    int S = 0, nchan = 2;
};

extern "C" void hx_multi_destroy(hx_multi *m)
{
    if (!m) return;
    for (hx_batch *b : m->part) hx_batch_destroy(b);
    delete m;
}

extern "C" hx_multi *hx_multi_create(int ndev, const int *devices, int nstreams, const HX_E_CONTROL *ec, int shared_control, int max_frames)
{
    if (nstreams <= 0 || max_frames <= 0 || !ec) { set_err("bad arguments"); return nullptr; }
    const int have = hx_device_count();
    if (ndev <= 0) ndev = have;
    if (ndev > nstreams) ndev = nstreams;
    if (ndev <= 0) { set_err("no HIP device available: the encoder has no CPU fallback"); return nullptr; }
    hx_multi *m = new hx_multi;
    m->S = nstreams;
    const int base = nstreams / ndev, rem = nstreams % ndev;       // block sizes differ by at most one (hmp3_amd/shard.py)
    for (int k = 0; k < ndev; k++) {
        const int first = k * base + (k < rem ? k : rem), count = base + (k < rem ? 1 : 0);
        const int dev = devices ? devices[k] : k;
        hx_batch *b = hx_batch_create(dev, count, shared_control ? ec : ec + first, shared_control, max_frames);
        if (!b) { hx_multi_destroy(m); return nullptr; }            // hx_last_error is hx_batch_create's
        m->part.push_back(b); m->first.push_back(first); m->count.push_back(count); m->device.push_back(dev);
        cpu_set_t cs;
        CPU_ZERO(&cs);
        const int node = hx_device_numa_node(dev);
        m->ncpus.push_back(node >= 0 ? node_cpus_allowed(node, &cs) : 0);
        m->cpus.push_back(cs);
        if (k == 0) m->nchan = b->nchan;
        else if (b->nchan != m->nchan || b->lsf != m->part[0]->lsf) { set_err("mono / stereo and MPEG-1 / MPEG-2 streams cannot share a batch"); hx_multi_destroy(m); return nullptr; }
    }
    return m;
}

extern "C" int hx_multi_ndevices(const hx_multi *m) { return m ? (int) m->part.size() : 0; }
extern "C" int hx_multi_nstreams(const hx_multi *m) { return m ? m->S : 0; }
extern "C" hx_batch *hx_multi_batch(hx_multi *m, int k) { return (m && k >= 0 && k < (int) m->part.size()) ? m->part[k] : nullptr; }
extern "C" int hx_multi_shard(const hx_multi *m, int k, int *device, int *first, int *count)
{
    if (!m || k < 0 || k >= (int) m->part.size()) return -1;
    if (device) *device = m->device[k];
    if (first) *first = m->first[k];
    if (count) *count = m->count[k];
    return 0;
}

extern "C" long long hx_multi_out_stride(const hx_multi *m, int nframes)
{
    long long n = 0;
    if (m) for (hx_batch *b : m->part) { const long long v = hx_batch_out_stride(b, nframes); if (v > n) n = v; }
    return n;
}

// one thread per device; kind 0 = int16, 1 = fp32, 2 = fp32 with per-frame counters
static int multi_call(hx_multi *m, int kind, const void *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes, int *stats)
{
    if (!m || !pcm || !out || !out_bytes || (kind == 2 && !stats)) { set_err("null buffer"); return -1; }
    if (out_stride < hx_multi_out_stride(m, nframes)) { set_err("out_stride is smaller than hx_multi_out_stride(m, nframes)"); return -1; }
    const size_t n = m->part.size(), esz = kind ? sizeof(float) : sizeof(int16_t);
    std::vector<int> rc(n, 0);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    for (size_t k = 0; k < n; k++)
        th.emplace_back([&, k]() {
            if (m->ncpus[k] > 0) sched_setaffinity(0, sizeof(cpu_set_t), &m->cpus[k]);     // this device's copies are issued from its own socket (best effort)
            const long long f = m->first[k];
            const char *p = (const char *) pcm + (size_t) f * nframes * 1152 * m->nchan * esz;
            unsigned char *o = out + f * out_stride;
            if (kind == 0) rc[k] = hx_batch_encode_s16_host(m->part[k], (const int16_t *) p, nframes, o, out_stride, out_bytes + f);
            else if (kind == 1) rc[k] = hx_batch_encode_f32_host(m->part[k], (const float *) p, nframes, o, out_stride, out_bytes + f);
            else rc[k] = hx_batch_encode_f32_host_stats(m->part[k], (const float *) p, nframes, o, out_stride, out_bytes + f, stats + f * nframes * 2);
            if (rc[k]) err[k] = hx_last_error();        // the message is thread-local: hand it to the caller's thread
        });
    for (std::thread &t : th) t.join();
    for (size_t k = 0; k < n; k++) if (rc[k]) { set_err("%s", err[k].c_str()); return rc[k]; }
    return 0;
}

extern "C" int hx_multi_encode_s16_host(hx_multi *m, const int16_t *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes)
{
    return multi_call(m, 0, pcm, nframes, out, out_stride, out_bytes, nullptr);
}
extern "C" int hx_multi_encode_f32_host(hx_multi *m, const float *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes)
{
    return multi_call(m, 1, pcm, nframes, out, out_stride, out_bytes, nullptr);
}
extern "C" int hx_multi_encode_f32_host_stats(hx_multi *m, const float *pcm, int nframes, unsigned char *out, long long out_stride, int *out_bytes, int *stats)
{
    return multi_call(m, 2, pcm, nframes, out, out_stride, out_bytes, stats);
}
extern "C" int hx_multi_status(hx_multi *m)
{
    int v = 0;
    if (!m) return -1;
    for (hx_batch *b : m->part) v |= hx_batch_status(b);
    return v;
}
