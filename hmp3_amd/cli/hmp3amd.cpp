// hmp3amd - file front end over libhmp3amd.
//   hmp3amd <input.wav|-> <output.mp3|-> [flags]                 one file, frame by frame (hx_enc_* API)
//   hmp3amd -batch in1.wav out1.mp3 in2.wav out2.mp3 ... [flags]  many files at once (hx_batch_* API)
// Same flags, encode loop and output files as the reference CLI (reference
// test/tomp3.cpp:336-602 main, :645-1088 ff_encode): Xing/Info tag frame first, audio frames, four
// frames of silence behind the input, drain until every submitted frame is out, then the tag is
// completed in place.  Accepted: RIFF/WAVE, mono or stereo, 8/16/24/32-bit PCM or 32-bit float,
// 8 - 48 kHz (rates other than 16 / 22.05 / 24 / 32 / 44.1 / 48 kHz, or -A, go through the sample-rate
// converter first, as in the reference); everything else fails like an unsupported file.
// Batch mode encodes all files as one batch of streams (same channel count, same flags) and
// reproduces per file exactly what the single-file loop writes.  Containers: RIFF, RIFX, RF64 / BW64, Wave64.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hmp3_amd.h"

namespace {

struct WavInfo { int channels = 0, rate = 0, bits = 0, type = 0, bigendian = 0; uint64_t data_bytes = 0; };
struct Options { HX_E_CONTROL ec; int xing_flag = 3 | 0x40, ignore_length = 0, mpeg_select = 0, ec_display = 0, ngpus = 0; };

// a header field of n bytes in the file's byte order
uint64_t field(const unsigned char *p, int n, int be)
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint64_t) p[be ? n - 1 - i : i] << (8 * i);
    return v;
}

bool read_exact(FILE *f, void *dst, size_t n) { return fread(dst, 1, n, f) == n; }
bool skip_bytes(FILE *f, uint64_t n)
{
    unsigned char tmp[4096];
    while (n) { size_t k = n < sizeof(tmp) ? (size_t) n : sizeof(tmp); if (fread(tmp, 1, k, f) != k) return false; n -= k; }
    return true;
}

// Walk to the start of the audio data (works on pipes: no seeking).  Containers and rules as the
// reference's pcmhead_file (pcmhpm.c:203-429): RIFF, RIFX (big-endian fields and samples), RF64 / BW64
// (64-bit data size in the ds64 chunk) and Sony Wave64 (GUID chunks, sizes include the 24-byte chunk
// header, bodies padded to 8).  bits = 8 * block align / channels, or the valid-bits field of a
// WAVE_FORMAT_EXTENSIBLE header; an odd data size is rounded up to even (the pad byte is read as audio).
bool wav_header(FILE *f, WavInfo *w)
{
    static const unsigned char w64_riff[16] = {0x72, 0x69, 0x66, 0x66, 0x2E, 0x91, 0xCF, 0x11, 0xA5, 0xD6, 0x28, 0xDB, 0x04, 0xC1, 0x00, 0x00};
    static const unsigned char w64_tail[12] = {0xF3, 0xAC, 0xD3, 0x11, 0x8C, 0xD1, 0x00, 0xC0, 0x4F, 0x8E, 0xDB, 0x8A};   // wave / fmt / data GUIDs after their four letters
    static const unsigned char ext_tail[14] = {0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xaa, 0x00, 0x38, 0x9b, 0x71};
    unsigned char h[24];
    bool w64 = false, rf64 = false;
    uint64_t ds64_data = 0;
    if (!read_exact(f, h, 8)) return false;
    if (!memcmp(h, "RIFF", 4)) { }
    else if (!memcmp(h, "RF64", 4) || !memcmp(h, "BW64", 4)) rf64 = true;
    else if (!memcmp(h, "RIFX", 4)) w->bigendian = 1;
    else {
        if (!read_exact(f, h + 8, 16) || memcmp(h, w64_riff, 16)) return false;
        w64 = true;
    }
    const int be = w->bigendian;
    if (w64) { if (!read_exact(f, h, 16) || memcmp(h, "wave", 4) || memcmp(h + 4, w64_tail, 12)) return false; }
    else if (!read_exact(f, h, 4) || memcmp(h, "WAVE", 4)) return false;
    if (rf64) {
        unsigned char d[36];
        if (!read_exact(f, d, 36) || memcmp(d, "ds64", 4)) return false;
        uint64_t size = field(d + 4, 4, be) + 8;
        if (size < 36) return false;
        if (size & 1) size++;
        ds64_data = field(d + 16, 8, be);
        if (!skip_bytes(f, size - 36)) return false;
    }
    bool have_fmt = false;
    for (;;) {
        uint64_t n;     // body bytes to consume, padding included
        bool is_fmt, is_data;
        if (w64) {
            if (!read_exact(f, h, 24)) return false;
            n = field(h + 16, 8, 0);
            if (n < 24) return false;
            n -= 24;
            if (n % 8) n += 8 - n % 8;
            const bool ours = !memcmp(h + 4, w64_tail, 12);
            is_fmt = ours && !memcmp(h, "fmt ", 4);
            is_data = ours && !memcmp(h, "data", 4);
        } else {
            if (!read_exact(f, h, 8)) return false;
            n = field(h + 4, 4, be);
            is_fmt = !memcmp(h, "fmt ", 4);
            is_data = !memcmp(h, "data", 4);
            if ((n & 1) && n != 0xFFFFFFFFu) n++;
            if (is_data && rf64 && n == 0xFFFFFFFFu) { n = ds64_data; if (n & 1) n++; }
        }
        if (is_fmt && !have_fmt) {
            if (n < 16 || n >= (1u << 31)) return false;
            std::vector<unsigned char> b((size_t) n);
            if (!read_exact(f, b.data(), b.size())) return false;
            w->type = (int) field(&b[0], 2, be);
            w->channels = (int) field(&b[2], 2, be);
            w->rate = (int) field(&b[4], 4, be);
            w->bits = (int) field(&b[14], 2, be);
            if (w->channels > 0) w->bits = 8 * ((int) field(&b[12], 2, be) / w->channels);
            if (w->type == 65534 && n >= 18 + 22 && field(&b[16], 2, be) >= 22) {      // WAVE_FORMAT_EXTENSIBLE
                if (!memcmp(&b[24 + 2], ext_tail, 14)) w->type = (int) field(&b[24], 2, be);
                w->bits = (int) field(&b[18], 2, be);
            }
            have_fmt = true;
        } else if (is_data && have_fmt) {
            w->data_bytes = n;
            return true;
        } else {
            if (n >= (1u << 31)) return false;
            if (!skip_bytes(f, n)) return false;
        }
    }
}

void usage()
{
    fprintf(stderr,
            "\n hmp3amd <input.wav|-> <output.mp3|-> [flags]"
            "\n hmp3amd -batch in1.wav out1.mp3 in2.wav out2.mp3 ... [flags]"
            "\n   -Bn  kbps per channel (CBR)      -Vn  VBR quality 0..150 (default 50)"
            "\n   -Mn  0 stereo, 1 joint stereo, 3 mono      -Fn  low-pass Hz      -HFn high-frequency mode"
            "\n   -SBTn short-block threshold      -S1  DC blocker       -Xn  0 no tag, 1 Xing, 2/3 + TOC, default + info"
            "\n   -Cn -On copyright / original bits   -Ln VBR bitrate cap   -Tn -TXn tuning   -IL ignore the WAV length field"
            "\n   -An  encode rate: 0 track the input (default), 1 an MPEG-1 rate, 2 an MPEG-2 rate, else that rate in Hz"
            "\n   -EC  print the encoder settings in use    -D  no progress display    -Gn  GPUs used by -batch (default all)\n");
}

// one input file, read and checked
struct Input {
    WavInfo wi;
    std::vector<unsigned char> data;    // the audio bytes; pad() appends the silence the encode loop runs into
    uint64_t audio_bytes = 0;
    int frame_in = 0, mono_convert = 0, is_float = 0;
    HX_E_CONTROL ec;                    // as given to the encoder
    HX_E_CONTROL ec_used;               // as reported back (settings in use)
    HX_MPEG_HEAD head;
    size_t size = 0;                    // bytes the encode loop may consume: audio + 4 x init_bytes of zero bytes
    // The reference refills a large buffer and, when it meets the end of the data, appends 4 x bytes_in_init
    // zero bytes (tomp3.cpp:925-934); a call is made while at least bytes_in_init bytes are left (:904-941).
    // A linear buffer of (data ++ zeros) gives the same sequence of calls.  The converter may stage more
    // than it consumes, so some slack follows.
    void pad(int init_bytes) { size = audio_bytes + 4 * (size_t) init_bytes; data.resize(size + (1 << 17), 0); }
};

// open the input, parse and check its header; on success *fp is positioned at the first audio byte and
// *indatasize is the number of audio bytes to read (UINT64_MAX: until the end of the file)
bool open_input(const char *path, const Options &opt, Input *in, FILE **fp, uint64_t *indatasize_out)
{
    FILE *f = strcmp(path, "-") ? fopen(path, "rb") : stdin;
    if (!f) { fprintf(stderr, "\n CANNOT_OPEN_INPUT_FILE %s\n", path); return false; }
    *fp = f;
    const int ignore_length = opt.ignore_length || f == stdin;
    if (!wav_header(f, &in->wi)) { fprintf(stderr, "\n UNRECOGNIZED PCM FILE TYPE\n"); return false; }
    const WavInfo &wi = in->wi;
    // a data size of 0xFFFFFFFF means "until the end of the file" (tomp3.cpp:751-768)
    const uint64_t indatasize = (ignore_length || wi.data_bytes == 0xFFFFFFFFu) ? UINT64_MAX : wi.data_bytes;
    if (indatasize == 0) { fprintf(stderr, "\n INPUT FILE CONTAINS NO AUDIO\n"); return false; }
    *indatasize_out = indatasize;
    fprintf(stderr, "\n pcm file:  channels = %d  bits = %d,  rate = %d  type = %d", wi.channels, wi.bits, wi.rate, wi.type);
    in->is_float = wi.type == 3;
    if ((wi.channels != 1 && wi.channels != 2) ||
        !((wi.type == 1 && (wi.bits == 8 || wi.bits == 16 || wi.bits == 24 || wi.bits == 32)) || (in->is_float && wi.bits == 32)) ||
        wi.rate < 8000 || wi.rate > 48000) {
        fprintf(stderr, "\n UNSUPPORTED PCM FILE TYPE\n Mono or stereo, 8/16/24/32-bit PCM or 32-bit float, 8000 Hz - 48000 Hz.\n");
        return false;
    }
    in->ec = opt.ec;
    if (in->ec.mode < 0) in->ec.mode = 0;
    in->mono_convert = in->ec.mode == 3;        // -M3: encode one channel (tomp3.cpp:562-563)
    if (wi.channels == 1) in->ec.mode = 3;
    else if (in->ec.mode == 3) in->ec.mode = 1;
    in->ec.samprate = wi.rate;
    in->frame_in = 1152 * wi.channels * (wi.bits / 8);
    return true;
}

// cvt_to_pcm (pcmhpm.c:454-484): big-endian samples to host byte order
void to_host_order(const WavInfo &wi, unsigned char *p, size_t nbytes)
{
    if (!wi.bigendian || wi.bits <= 8) return;
    const size_t bs = (size_t) wi.bits / 8, ns = nbytes / bs;
    for (size_t i = 0; i < ns; i++) std::reverse(p + i * bs, p + (i + 1) * bs);
}

// a whole input in memory (regular files of the batched routes)
bool load_input(const char *path, const Options &opt, Input *in)
{
    FILE *f = nullptr;
    uint64_t indatasize = 0;
    if (!open_input(path, opt, in, &f, &indatasize)) { if (f && f != stdin) fclose(f); return false; }
    const WavInfo &wi = in->wi;
    std::vector<unsigned char> chunk(1 << 20);
    while (in->data.size() < indatasize) {
        size_t want = chunk.size();
        if (in->data.size() + want > indatasize) want = (size_t) (indatasize - in->data.size());
        const size_t got = fread(chunk.data(), 1, want, f);
        in->data.insert(in->data.end(), chunk.begin(), chunk.begin() + got);
        if (got < want) break;
    }
    to_host_order(wi, in->data.data(), in->data.size());
    in->audio_bytes = in->data.size();
    if (f != stdin) fclose(f);
    return true;
}

// the tag frame, its running seek table and the MusicCRC of one output file (tomp3.cpp:871-896, :976-984, :1055-1072)
struct Tagger {
    hx_xing *xg = hx_xing_create();
    std::vector<unsigned char> tag = std::vector<unsigned char>(2048, 0);
    int head_flags = 0, head_bytes = 0, vbr_scale = -1, toc_counter = 0;
    unsigned crc = 0;
    ~Tagger() { hx_xing_destroy(xg); }
    void begin(const Input &in, int xing_flag)
    {
        if (xing_flag) head_flags = 1 | 2 | 8;
        if (xing_flag & 2) head_flags |= 4 | 0x40;
        if (!xing_flag) return;
        if (in.ec_used.vbr_flag) vbr_scale = in.ec_used.vbr_mnr;
        head_bytes = hx_xing_header(xg, in.ec_used.samprate, in.head.mode, in.ec_used.cr_bit, in.ec_used.original, head_flags, 0, 0,
                                    vbr_scale, nullptr, tag.data(), nullptr, nullptr, in.ec_used.bitrate * in.wi.channels);
    }
    void after_call(unsigned frames_out, unsigned bytes_out)    // once per input frame of the main loop
    {
        if ((head_flags & 4) && --toc_counter <= 0) toc_counter = hx_xing_toc(xg, (int) frames_out + 1, (int) bytes_out + head_bytes);
    }
    void bytes(const unsigned char *p, int n) { crc = hx_xing_update_crc((unsigned short) crc, p, n); }
    void finish(const Input &in, unsigned frames, uint64_t out_bytes)
    {
        const uint64_t samples_audio = in.audio_bytes / (uint64_t) (in.wi.channels * (in.wi.bits / 8));
        hx_xing_update_info(xg, frames, (int) out_bytes, vbr_scale, nullptr, tag.data(), nullptr, nullptr, samples_audio,
                            (unsigned) out_bytes, (unsigned) in.ec_used.freq_limit, (unsigned) in.wi.rate, (unsigned) in.ec_used.samprate,
                            (unsigned short) crc);
    }
};

int encode_loaded(std::vector<Input> &in, const std::vector<const char *> &files, const Options &opt);

bool mpeg_rate(int r) { return r == 32000 || r == 44100 || r == 48000 || r == 16000 || r == 22050 || r == 24000; }

// tomp3.cpp:1169-1196 (-EC)
void print_ec(const HX_E_CONTROL *ec)
{
    fprintf(stderr, "\n-------------------------------------------------------------------------------");
    fprintf(stderr, "\nec->layer =         %d\t\t\tec->mode =          %d", ec->layer, ec->mode);
    fprintf(stderr, "\nec->bitrate =       %d\t\t\tec->samprate =      %d", ec->bitrate, ec->samprate);
    fprintf(stderr, "\nec->nsbstereo =     %d\t\t\tec->freq_limit =    %d", ec->nsbstereo, ec->freq_limit);
    fprintf(stderr, "\nec->filter_select = %d\t\t\tec->nsb_limit =     %d", ec->filter_select, ec->nsb_limit);
    fprintf(stderr, "\nec->cr_bit =        %d\t\t\tec->original =      %d", ec->cr_bit, ec->original);
    fprintf(stderr, "\nec->hf_flag =       %d\t\t\tec->vbr_flag =      %d", ec->hf_flag, ec->vbr_flag);
    fprintf(stderr, "\nec->vbr_mnr =       %d\t\t\tec->vbr_br_limit =  %d", ec->vbr_mnr, ec->vbr_br_limit);
    fprintf(stderr, "\nec->sparse_scale =  %d\t\t\tec->chan_add_f0 =   %d", ec->sparse_scale, ec->chan_add_f0);
    fprintf(stderr, "\nec->vbr_delta_mnr = %d\t\t\tec->chan_add_f1 =   %d", ec->vbr_delta_mnr, ec->chan_add_f1);
    fprintf(stderr, "\nec->quick =         %d\t\t\tec->cpu_select =    %d", ec->quick, ec->cpu_select);
    fprintf(stderr, "\nec->test1 =         %d\t\t\tec->short_block_threshold = %d", ec->test1, ec->short_block_threshold);
    fprintf(stderr, "\n-------------------------------------------------------------------------------");
}

// regular files up to this size take the batched route (whole file in memory); larger ones and pipes stream
const uint64_t kBatchRouteMaxBytes = 1ull << 30;

bool regular_file_size(const char *path, uint64_t *size)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    const bool ok = fseek(f, 0, SEEK_END) == 0;
    const long n = ok ? ftell(f) : -1;
    fclose(f);
    if (n < 0) return false;
    *size = (uint64_t) n;
    return true;
}

// The reference's loop (tomp3.cpp:897-1050) with its bounded buffers: the input is read in pieces into a
// sliding window, every call's bytes are written as they arrive, the tag is completed by seeking back at the
// end.  Memory does not grow with the length of the input, so an unbounded pipe works.
int encode_streaming(const char *fin, const char *fout, const Options &opt)
{
    Input in;
    FILE *fi = nullptr;
    uint64_t indatasize = 0;
    if (!open_input(fin, opt, &in, &fi, &indatasize)) { if (fi && fi != stdin) fclose(fi); return 1; }
    int rc = 1;
    hx_enc *enc = hx_enc_create(0);
    FILE *out = nullptr;
    do {
        // bytes a call needs in the buffer (more than it consumes); with a sample-rate conversion a call consumes a varying amount
        const int init_bytes = enc ? hx_enc_MP3_audio_encode_init(enc, &in.ec, in.wi.bits, in.is_float, opt.mpeg_select, in.mono_convert) : 0;
        if (!init_bytes) { fprintf(stderr, "\n ENCODER INIT FAIL: %s\n", hx_last_error()); break; }
        out = strcmp(fout, "-") ? fopen(fout, "w+b") : stdout;
        if (!out) { fprintf(stderr, "\n CANNOT CREATE OUTPUT FILE\n"); break; }
        char info[128];
        hx_enc_info_string(enc, info);
        fprintf(stderr, "\n %s\n", info);
        hx_enc_info_ec(enc, &in.ec_used);       // the settings actually in use
        hx_enc_info_head(enc, &in.head);
        if (opt.ec_display) print_ec(&in.ec_used);
        Tagger tg;
        tg.begin(in, opt.xing_flag);
        uint64_t out_bytes = tg.head_bytes;
        if (tg.head_bytes && fwrite(tg.tag.data(), 1, tg.head_bytes, out) != (size_t) tg.head_bytes) { fprintf(stderr, "\n FILE WRITE ERROR\n"); break; }

        // window over (audio ++ 4 x init_bytes zero bytes): a call is made while at least init_bytes are
        // left (tomp3.cpp:904-941); the converter may stage more than it consumes, hence the slack
        const size_t slack = 1 << 17, piece = 1 << 20;
        std::vector<unsigned char> win(piece + (size_t) init_bytes + slack, 0), bs(128 * 1024), zero((size_t) 4 * init_bytes + slack, 0);
        size_t lo = 0, hi = 0;                  // valid bytes of the window: [lo, hi)
        // The reference appends the four calls' worth of silence only if it fits a limit derived from its input buffer
        // (128 frames of stereo floats) and a per-format factor (tomp3.cpp:268,802,925; pcmhpm.c:450): with 32-bit
        // samples and a down-conversion by more than 1.14 (24-bit: 2.03) it never does, and the tail of the input that
        // is shorter than one call's need is dropped.  Found by tools/fuzz_cli.py.
        const uint64_t ref_bufbytes = 128ull * sizeof(float) * 2304, fmt_factor = (uint64_t) in.wi.channels * (uint64_t) ((in.wi.bits * 7) / 8);
        const bool pad_fits = fmt_factor != 0 && (((ref_bufbytes << 1) / fmt_factor) & ~1ull) > 4ull * (uint64_t) init_bytes;
        uint64_t audio = 0, zeros_left = pad_fits ? 4ull * (uint64_t) init_bytes : 0;
        bool eof = false, werr = false;
        unsigned frames_expected = 0;
        auto emit = [&](const HX_IN_OUT &x) {
            if (x.out_bytes && fwrite(bs.data(), 1, x.out_bytes, out) != (size_t) x.out_bytes) werr = true;
            tg.bytes(bs.data(), x.out_bytes);
            out_bytes += x.out_bytes;
        };
        for (;;) {
            if (hi - lo < (size_t) init_bytes && !(eof && zeros_left == 0)) {      // refill
                memmove(win.data(), win.data() + lo, hi - lo);
                hi -= lo; lo = 0;
                while (hi + 8 <= piece && !(eof && zeros_left == 0)) {     // (less than a sample's room left counts as full)
                    if (!eof) {
                        size_t want = piece - hi;
                        if ((uint64_t) want > indatasize - audio) want = (size_t) (indatasize - audio);
                        else {      // every piece ends on a sample boundary of the data (24-bit samples do not divide 1 MiB), so
                                    // that the byte swap of big-endian input never tears a sample; the data's torn tail is padding
                            const size_t bsz = (size_t) std::max(in.wi.bits / 8, 1), over = (size_t) ((audio + want) % bsz);
                            if (want > over) want -= over;
                        }
                        const size_t got = want ? fread(win.data() + hi, 1, want, fi) : 0;
                        to_host_order(in.wi, win.data() + hi, got);
                        hi += got; audio += got;
                        if (got < want || audio >= indatasize) eof = true;
                    } else {
                        const size_t z = (size_t) std::min<uint64_t>(zeros_left, piece - hi);
                        memset(win.data() + hi, 0, z);
                        hi += z; zeros_left -= z;
                    }
                }
                memset(win.data() + hi, 0, win.size() - hi);
            }
            if (hi - lo < (size_t) init_bytes) break;
            const HX_IN_OUT x = hx_enc_MP3_audio_encode(enc, win.data() + lo, bs.data());
            emit(x);
            lo += (size_t) x.in_bytes;
            frames_expected++;
            const HX_INT_PAIR fb = hx_enc_get_frames_bytes(enc);
            tg.after_call((unsigned) fb.a, (unsigned) fb.b);
            if (werr) break;
        }
        in.audio_bytes = audio;
        if (in.ec_used.samprate < 32000) frames_expected *= 2;                          // MPEG-2: two frames per call (tomp3.cpp:1022)
        // drain, tomp3.cpp:1020-1036 (bounded: an encoder that stopped emitting - a failed device call - must not hang the tool)
        for (unsigned spare = frames_expected + 64; !werr && hx_enc_get_frames(enc) < frames_expected && spare; spare--)
            emit(hx_enc_MP3_audio_encode(enc, zero.data(), bs.data()));
        if (hx_enc_get_frames(enc) < frames_expected) { fprintf(stderr, "\n ENCODER FAIL: %s\n", hx_last_error()); break; }
        if (werr) { fprintf(stderr, "\n FILE WRITE ERROR\n"); break; }
        const unsigned frames = hx_enc_get_frames(enc);
        if (opt.xing_flag) {
            tg.finish(in, frames, out_bytes);
            if (out == stdout || fseek(out, 0, SEEK_SET) != 0) fprintf(stderr, "\n OUTPUT IS NOT SEEKABLE: TAG FRAME LEFT WITHOUT TOTALS");
            else fwrite(tg.tag.data(), 1, tg.head_bytes, out);
        }
        fprintf(stderr, "\n %u frames, %llu bytes, %.2f kbps\n", frames, (unsigned long long) out_bytes, hx_enc_get_bitrate_float(enc));
        rc = frames == 0 ? 1 : 0;
    } while (0);
    if (out && out != stdout) fclose(out);
    if (fi && fi != stdin) fclose(fi);
    hx_enc_destroy(enc);
    return rc;
}

int encode_one_file(const char *fin, const char *fout, const Options &opt)
{
    // A regular file of bounded size at an MPEG rate with a seekable output goes through the batched API as a
    // batch of one (96 frames per call instead of one), which writes exactly what the frame-by-frame loop
    // writes.  Pipes, very large files and inputs that need a rate conversion stream frame by frame.
    uint64_t fsize = 0;
    if (!opt.mpeg_select && strcmp(fin, "-") && strcmp(fout, "-") && regular_file_size(fin, &fsize) && fsize <= kBatchRouteMaxBytes) {
        std::vector<Input> one(1);
        if (!load_input(fin, opt, &one[0])) return 1;
        if (mpeg_rate(one[0].wi.rate)) return encode_loaded(one, {fin, fout}, opt);
    }
    return encode_streaming(fin, fout, opt);
}

// samples of one input frame as fp32 at int16 scale, the way Csrc::sr_convert / src_filter_to_mono_case0
// produce them (srcc.cpp:804-836, srccf.cpp:458-468)
// what hx_enc_MP3_audio_encode_init returns when source and encode rate are equal: 1153 sample frames
int init_bytes_same_rate(const Input &in) { return 1153 * in.wi.channels * (in.wi.bits / 8); }
// calls of the single-file loop: one per frame_in bytes while at least init_bytes are left
size_t ncalls(const Input &in)
{
    const size_t init = (size_t) init_bytes_same_rate(in);
    return in.size >= init ? (in.size - init) / (size_t) in.frame_in + 1 : 0;
}

void frame_to_float(const Input &in, const unsigned char *src, float *dst)
{
    const int ns = 1152 * in.wi.channels;
    if (in.wi.bits == 32 && in.is_float) { const float *f = (const float *) src; for (int i = 0; i < ns; i++) dst[i] = f[i] * 32768.0f; }
    else if (in.wi.bits == 32) { const int *s = (const int *) src; for (int i = 0; i < ns; i++) dst[i] = (float) (s[i] / 65536.0f); }
    else if (in.wi.bits == 24)
        for (int i = 0; i < ns; i++) {
            const unsigned char *b = src + 3 * i;
            const int s = (int) (((unsigned) b[2] << 24) | ((unsigned) b[1] << 16) | ((unsigned) b[0] << 8)) >> 8;
            dst[i] = (float) ((float) s / 256.0f);
        }
    else if (in.wi.bits == 16) { const int16_t *s = (const int16_t *) src; for (int i = 0; i < ns; i++) dst[i] = (float) s[i]; }
    else for (int i = 0; i < ns; i++) dst[i] = (((float) src[i]) - 128.0f) * (256.0f);
    if (in.wi.channels == 2 && in.mono_convert)
        for (int i = 0; i < 1152; i++) dst[i] = (float) ((dst[2 * i] + dst[2 * i + 1]) * 0.5);
}

int encode_batch(const std::vector<const char *> &files, const Options &opt)
{
    const int S = (int) files.size() / 2;
    std::vector<Input> in(S);
    for (int i = 0; i < S; i++) if (!load_input(files[2 * i], opt, &in[i])) return 1;
    return encode_loaded(in, files, opt);
}

int encode_loaded(std::vector<Input> &in, const std::vector<const char *> &files, const Options &opt)
{
    const int S = (int) in.size();
    std::vector<HX_E_CONTROL> ctl(S);
    int nch = 0;
    size_t max_calls = 0;
    for (int i = 0; i < S; i++) {
        HX_E_CONTROL ec = in[i].ec;
        if (in[i].mono_convert) ec.mode = 3;
        const int c = ec.mode == 3 ? 1 : 2;
        if (nch && c != nch) { fprintf(stderr, "\n -batch needs files that all encode to the same channel count\n"); return 1; }
        nch = c;
        if (!hx_control_info(&ec, &in[i].ec_used, &in[i].head)) { fprintf(stderr, "\n ENCODER INIT FAIL (%s)\n", files[2 * i]); return 1; }
        ctl[i] = ec;
        if (opt.mpeg_select || !mpeg_rate(in[i].wi.rate)) {
            fprintf(stderr, "\n -batch encodes files at their own MPEG sample rate; %s needs a rate conversion (use the single-file mode)\n", files[2 * i]);
            return 1;
        }
        in[i].pad(init_bytes_same_rate(in[i]));
        const size_t calls = ncalls(in[i]);
        if (calls > max_calls) max_calls = calls;
    }
    const int CH = 96;                                  // frames per batched call
    const size_t total = max_calls + 32;                // room for the drain frames (a reservoir never spans that many)
    // the files spread over the node's GPUs in contiguous blocks (-Gn limits the count), one host thread per device
    hx_multi *b = hx_multi_create(opt.ngpus, nullptr, S, ctl.data(), 0, CH);
    if (!b) { fprintf(stderr, "\n ENCODER INIT FAIL: %s\n", hx_last_error()); return 1; }
    if (S > 1) fprintf(stderr, "\n %d files on %d GPU(s)", S, hx_multi_ndevices(b));
    if (opt.ec_display) print_ec(&in[0].ec_used);
    const long long stride = hx_multi_out_stride(b, CH);
    std::vector<float> pcm((size_t) S * CH * 1152 * nch), tmp(2304);
    const std::vector<unsigned char> zero_frame(2304 * 4, 0);
    std::vector<unsigned char> out((size_t) S * stride);
    std::vector<int> nb(S), stats((size_t) S * CH * 2);
    std::vector<std::vector<unsigned char>> stream(S);
    std::vector<std::vector<unsigned>> fr(S), by(S);    // per input frame: frames / bytes out so far
    for (size_t c0 = 0; c0 < total; c0 += CH) {
        for (int i = 0; i < S; i++) {
            const size_t calls = ncalls(in[i]);
            for (int k = 0; k < CH; k++) {
                // past the end the single-file loop feeds frames of zero BYTES: silence, except for
                // 8-bit unsigned input where a zero byte is full-scale negative
                if (c0 + k < calls) frame_to_float(in[i], in[i].data.data() + (c0 + k) * in[i].frame_in, tmp.data());
                else frame_to_float(in[i], zero_frame.data(), tmp.data());
                memcpy(&pcm[((size_t) i * CH + k) * 1152 * nch], tmp.data(), sizeof(float) * 1152 * nch);
            }
        }
        if (hx_multi_encode_f32_host_stats(b, pcm.data(), CH, out.data(), stride, nb.data(), stats.data()) != 0) {
            fprintf(stderr, "\n ENCODE FAIL: %s\n", hx_last_error());
            hx_multi_destroy(b);
            return 1;
        }
        for (int i = 0; i < S; i++) {
            stream[i].insert(stream[i].end(), out.begin() + (size_t) i * stride, out.begin() + (size_t) i * stride + nb[i]);
            for (int k = 0; k < CH; k++) { fr[i].push_back((unsigned) stats[((size_t) i * CH + k) * 2]); by[i].push_back((unsigned) stats[((size_t) i * CH + k) * 2 + 1]); }
        }
    }
    if (hx_multi_status(b) != 0) fprintf(stderr, "\n WARNING: kernel status %d\n", hx_multi_status(b));
    hx_multi_destroy(b);
    // per file: what the single-file loop would have written
    int rc = 0;
    for (int i = 0; i < S; i++) {
        const size_t calls = ncalls(in[i]);
        Tagger tg;
        tg.begin(in[i], opt.xing_flag);
        for (size_t u = 0; u < calls; u++) tg.after_call(fr[i][u], by[i][u]);
        size_t u = calls;                               // drain calls: while (get_frames() < frames_expected) encode silence
        const size_t expected = calls * (in[i].ec_used.samprate < 32000 ? 2 : 1);   // MPEG-2: two frames per call
        while (u < fr[i].size() && fr[i][u - 1] < expected) u++;
        if (fr[i][u - 1] < expected) { fprintf(stderr, "\n %s: drain did not complete\n", files[2 * i]); rc = 1; }
        const unsigned frames = fr[i][u - 1], nbytes = by[i][u - 1];
        tg.bytes(stream[i].data(), (int) nbytes);
        const uint64_t out_bytes = (uint64_t) tg.head_bytes + nbytes;
        if (opt.xing_flag) tg.finish(in[i], frames, out_bytes);
        FILE *o = fopen(files[2 * i + 1], "wb");
        if (!o) { fprintf(stderr, "\n CANNOT CREATE OUTPUT FILE %s\n", files[2 * i + 1]); rc = 1; continue; }
        fwrite(tg.tag.data(), 1, tg.head_bytes, o);
        fwrite(stream[i].data(), 1, nbytes, o);
        fclose(o);
        fprintf(stderr, "\n %s: %u frames, %llu bytes", files[2 * i + 1], frames, (unsigned long long) out_bytes);
    }
    fprintf(stderr, "\n");
    return rc;
}

}  // namespace

int main(int argc, char **argv)
{
    Options opt;
    hx_default_control(&opt.ec);        // the reference CLI's defaults (tomp3.cpp:357-384)
    opt.ec.bitrate = -1;
    std::vector<const char *> files;
    bool batch = false;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || a[1] == '\0') { files.push_back(a); continue; }
        if (!strcmp(a, "-batch")) { batch = true; continue; }
        HX_E_CONTROL &ec = opt.ec;
        const char c = (char) (a[1] | 0x20), c2 = (char) (a[2] | 0x20);
        switch (c) {
        case 'h': if (c2 == 'f') ec.hf_flag = 1 | atoi(a + 3); else { usage(); return 0; } break;
        case 'e': if (c2 == 'c') opt.ec_display = 1; break;            // -EC: print the settings in use (tomp3.cpp:412-416)
        case 'g': opt.ngpus = atoi(a + 2); break;                      // -Gn: GPUs for -batch (default: all)
        case 'd': case 'p': case 'z': break;                            // -D progress display off, -P / -Z reserved (tomp3.cpp:438-441,470-472,508-511): nothing to do here
        case 'q': ec.quick = atoi(a + 2); break;
        case 'u': ec.cpu_select = atoi(a + 2); break;
        case 'x': opt.xing_flag = atoi(a + 2); if (opt.xing_flag == 2) opt.xing_flag = 3; break;
        case 'b': ec.bitrate = atoi(a + 2); break;
        case 'c': ec.cr_bit = atoi(a + 2); break;
        case 'o': ec.original = atoi(a + 2); break;
        case 'm': ec.mode = atoi(a + 2); break;
        case 'n': ec.nsbstereo = atoi(a + 2); break;
        case 's': if (c2 == 'b' && (a[3] | 0x20) == 't') ec.short_block_threshold = atoi(a + 4); else ec.filter_select = atoi(a + 2); break;
        case 'f': ec.freq_limit = atoi(a + 2); break;
        case 't': if (c2 == 'x') ec.test1 = atoi(a + 3); else ec.vbr_delta_mnr = atoi(a + 2); break;
        case 'i': if (c2 == 'l') opt.ignore_length = 1; else ec.chan_add_f0 = atoi(a + 2); break;
        case 'j': ec.chan_add_f1 = atoi(a + 2); break;
        case 'v': ec.vbr_flag = 1; ec.vbr_mnr = atoi(a + 2); break;
        case 'l': ec.vbr_br_limit = atoi(a + 2); break;
        case 'a': opt.mpeg_select = atoi(a + 2); if (opt.mpeg_select < 0) opt.mpeg_select = 0; break;
        case 'w': {                 // -W<file>: 21 per-band MNR offsets (tomp3.cpp:552-555, get_mnr_adjust :1203-1233)
            FILE *f = fopen(a + 2, "rt");
            if (!f) break;          // (the reference ignores a file it cannot open)
            for (int k = 0; k < 21; k++) ec.mnr_adjust[k] = 0;
            for (int k = 0, m = 0; k < 21 && fscanf(f, "%d", &m) == 1; k++) ec.mnr_adjust[k] = m;
            fclose(f);
            fprintf(stderr, "\nMNR adjust ");
            for (int k = 0; k < 21; k++) { ec.mnr_adjust[k] = std::max(-200, std::min(200, ec.mnr_adjust[k])); fprintf(stderr, " %d", ec.mnr_adjust[k]); }
            break;
        }
        default: break;             // unknown switches: ignored like the reference does
        }
    }
    opt.ec.vbr_flag = opt.ec.bitrate < 0 ? 1 : 0;
    if (batch) {
        if (files.empty() || (files.size() & 1)) { usage(); return 1; }
        return encode_batch(files, opt);
    }
    if (files.size() < 2) { usage(); return 1; }
    return encode_one_file(files[0], files[1], opt);
}
