// hmp3amd - file front end over libhmp3amd: `hmp3amd <input.wav|-> <output.mp3|-> [flags]`.
// Same command line, encode loop and output file as the reference CLI (SURVEY §8 f1/f2;
// reference test/tomp3.cpp:336-602 main, :645-1088 ff_encode): Xing/Info tag frame first, audio
// frames, four frames of silence behind the input, drain until every submitted frame is out, then
// the tag is completed in place.  Accepted here: RIFF/WAVE, mono or stereo, 8/16/24/32-bit PCM or 32-bit float,
// 32 / 44.1 / 48 kHz (what the GPU path encodes); everything else fails like an unsupported file.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hmp3_amd.h"

namespace {

struct WavInfo { int channels = 0, rate = 0, bits = 0, type = 0; uint64_t data_bytes = 0; };

unsigned rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned) p[3] << 24); }
unsigned rd16(const unsigned char *p) { return p[0] | (p[1] << 8); }

bool read_exact(FILE *f, void *dst, size_t n) { return fread(dst, 1, n, f) == n; }
bool skip_bytes(FILE *f, uint64_t n)
{
    unsigned char tmp[4096];
    while (n) { size_t k = n < sizeof(tmp) ? (size_t) n : sizeof(tmp); if (fread(tmp, 1, k, f) != k) return false; n -= k; }
    return true;
}

// RIFF walk up to the start of the "data" chunk (works on pipes: no seeking)
bool wav_header(FILE *f, WavInfo *w)
{
    unsigned char h[12];
    if (!read_exact(f, h, 12) || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) return false;
    bool have_fmt = false;
    for (;;) {
        unsigned char c[8];
        if (!read_exact(f, c, 8)) return false;
        const unsigned n = rd32(c + 4);
        if (!memcmp(c, "fmt ", 4)) {
            std::vector<unsigned char> b(n + (n & 1));
            if (n < 16 || !read_exact(f, b.data(), b.size())) return false;
            w->type = (int) rd16(&b[0]);
            w->channels = (int) rd16(&b[2]);
            w->rate = (int) rd32(&b[4]);
            w->bits = (int) rd16(&b[14]);
            if (w->type == 0xFFFE && n >= 26) w->type = (int) rd16(&b[24]);    // WAVE_FORMAT_EXTENSIBLE: sub-format
            have_fmt = true;
        } else if (!memcmp(c, "data", 4)) {
            if (!have_fmt) return false;
            w->data_bytes = n;
            return true;
        } else if (!skip_bytes(f, (uint64_t) n + (n & 1))) return false;
    }
}

void usage()
{
    fprintf(stderr,
            "\n hmp3amd <input.wav|-> <output.mp3|-> [-Bn] [-Vn] [-Mn] [-Fn] [-HFn] [-SBTn] [-Sn] [-Xn] [-Cn] [-On] [-Ln] [-Tn] [-TXn] [-IL]"
            "\n   -Bn  kbps per channel (CBR)      -Vn  VBR quality 0..150 (default 50)"
            "\n   -Mn  0 stereo, 1 joint stereo    -Fn  low-pass Hz      -HFn high-frequency mode"
            "\n   -SBTn short-block threshold      -S1  DC blocker       -Xn  0 no tag, 1 Xing, 2/3 + TOC, default + info"
            "\n   -IL  ignore the length field of the WAV header\n");
}

}  // namespace

int main(int argc, char **argv)
{
    HX_E_CONTROL ec;
    hx_default_control(&ec);        // the reference CLI's defaults (tomp3.cpp:357-384)
    ec.bitrate = -1;
    int xing_flag = 3 | 0x40, ignore_length = 0;
    const char *fin = nullptr, *fout = nullptr;
    int k = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || a[1] == '\0') { if (k == 0) fin = a; if (k == 1) fout = a; k++; continue; }
        const char c = (char) (a[1] | 0x20), c2 = (char) (a[2] | 0x20);
        switch (c) {
        case 'h': if (c2 == 'f') ec.hf_flag = 1 | atoi(a + 3); else { usage(); return 0; } break;
        case 'q': ec.quick = atoi(a + 2); break;
        case 'u': ec.cpu_select = atoi(a + 2); break;
        case 'x': xing_flag = atoi(a + 2); if (xing_flag == 2) xing_flag = 3; break;
        case 'b': ec.bitrate = atoi(a + 2); break;
        case 'c': ec.cr_bit = atoi(a + 2); break;
        case 'o': ec.original = atoi(a + 2); break;
        case 'm': ec.mode = atoi(a + 2); break;
        case 'n': ec.nsbstereo = atoi(a + 2); break;
        case 's': if (c2 == 'b' && (a[3] | 0x20) == 't') ec.short_block_threshold = atoi(a + 4); else ec.filter_select = atoi(a + 2); break;
        case 'f': ec.freq_limit = atoi(a + 2); break;
        case 't': if (c2 == 'x') ec.test1 = atoi(a + 3); else ec.vbr_delta_mnr = atoi(a + 2); break;
        case 'i': if (c2 == 'l') ignore_length = 1; else ec.chan_add_f0 = atoi(a + 2); break;
        case 'j': ec.chan_add_f1 = atoi(a + 2); break;
        case 'v': ec.vbr_flag = 1; ec.vbr_mnr = atoi(a + 2); break;
        case 'l': ec.vbr_br_limit = atoi(a + 2); break;
        default: break;             // -D -EC -P -Z -A -W: display / reserved switches, no effect on the stream
        }
    }
    if (!fin || !fout) { usage(); return 1; }
    ec.vbr_flag = ec.bitrate < 0 ? 1 : 0;

    FILE *in = strcmp(fin, "-") ? fopen(fin, "rb") : stdin;
    if (!in) { fprintf(stderr, "\n CANNOT_OPEN_INPUT_FILE\n"); return 1; }
    if (in == stdin) ignore_length = 1;
    WavInfo wi;
    if (!wav_header(in, &wi)) { fprintf(stderr, "\n UNRECOGNIZED PCM FILE TYPE\n"); return 1; }
    uint64_t indatasize = ignore_length ? UINT64_MAX : wi.data_bytes;
    if (indatasize == 0) { fprintf(stderr, "\n INPUT FILE CONTAINS NO AUDIO\n"); return 1; }
    fprintf(stderr, "\n pcm file:  channels = %d  bits = %d,  rate = %d  type = %d", wi.channels, wi.bits, wi.rate, wi.type);
    const bool is_float = wi.type == 3;
    if ((wi.channels != 1 && wi.channels != 2) || !((wi.type == 1 && (wi.bits == 8 || wi.bits == 16 || wi.bits == 24 || wi.bits == 32)) || (is_float && wi.bits == 32)) ||
        (wi.rate != 32000 && wi.rate != 44100 && wi.rate != 48000)) {
        fprintf(stderr, "\n UNSUPPORTED PCM FILE TYPE\n This build encodes mono or stereo 8/16/24/32-bit PCM or 32-bit float input at 32 / 44.1 / 48 kHz.\n");
        return 1;
    }
    if (ec.mode < 0) ec.mode = 0;
    const int mono_convert = ec.mode == 3;      // -M3: encode one channel (tomp3.cpp:562-563)
    if (wi.channels == 1) ec.mode = 3;
    else if (ec.mode == 3) ec.mode = 1;
    ec.samprate = wi.rate;

    hx_enc *enc = hx_enc_create(0);
    const int frame_in = enc ? hx_enc_MP3_audio_encode_init(enc, &ec, wi.bits, is_float, 0, mono_convert) : 0;
    if (!frame_in) { fprintf(stderr, "\n ENCODER INIT FAIL: %s\n", hx_last_error()); return 1; }
    FILE *out = strcmp(fout, "-") ? fopen(fout, "w+b") : stdout;
    if (!out) { fprintf(stderr, "\n CANNOT CREATE OUTPUT FILE\n"); return 1; }
    char info[128];
    hx_enc_info_string(enc, info);
    fprintf(stderr, "\n %s\n", info);
    hx_enc_info_ec(enc, &ec);       // the settings actually in use

    // ---- tag frame first (tomp3.cpp:871-896) ----
    int head_flags = 0, head_bytes = 0, vbr_scale = -1;
    if (xing_flag) head_flags = 1 | 2 | 8;
    if (xing_flag & 2) head_flags |= 4 | 0x40;
    hx_xing *xg = hx_xing_create();
    std::vector<unsigned char> tag(2048, 0);
    uint64_t out_bytes = 0;
    if (xing_flag) {
        HX_MPEG_HEAD head;
        hx_enc_info_head(enc, &head);
        if (ec.vbr_flag) vbr_scale = ec.vbr_mnr;
        head_bytes = hx_xing_header(xg, ec.samprate, head.mode, ec.cr_bit, ec.original, head_flags, 0, 0, vbr_scale, nullptr,
                                    tag.data(), nullptr, nullptr, ec.bitrate * wi.channels);
        if (fwrite(tag.data(), 1, head_bytes, out) != (size_t) head_bytes) { fprintf(stderr, "\n FILE WRITE ERROR\n"); return 1; }
        out_bytes += head_bytes;
    }

    // ---- encode: the input, then four frames of silence, whole frames only (tomp3.cpp:906-1003) ----
    // The reference refills a 256-frame buffer and appends the silence when it meets the end of the
    // data; feeding whole frames of (data ++ 4 frames of zero bytes) is the same sequence of calls.
    std::vector<unsigned char> audio, bs(128 * 1024);
    {
        std::vector<unsigned char> chunk(1 << 20);
        while (audio.size() < indatasize) {
            size_t want = chunk.size();
            if (audio.size() + want > indatasize) want = (size_t) (indatasize - audio.size());
            const size_t got = fread(chunk.data(), 1, want, in);
            audio.insert(audio.end(), chunk.begin(), chunk.begin() + got);
            if (got < want) break;
        }
    }
    const uint64_t audio_bytes = audio.size();
    audio.resize(audio.size() + 4 * (size_t) frame_in, 0);
    std::vector<unsigned char> pcm(frame_in);
    unsigned frames_expected = 0, crc = 0;
    int toc_counter = 0;
    for (size_t off = 0; off + frame_in <= audio.size(); off += frame_in) {
        const HX_IN_OUT x = hx_enc_MP3_audio_encode(enc, audio.data() + off, bs.data());
        frames_expected++;
        if (x.out_bytes) {
            if (fwrite(bs.data(), 1, x.out_bytes, out) != (size_t) x.out_bytes) { fprintf(stderr, "\n FILE WRITE ERROR\n"); return 1; }
            crc = hx_xing_update_crc((unsigned short) crc, bs.data(), x.out_bytes);
            out_bytes += x.out_bytes;
        }
        if (head_flags & 4) {
            if (--toc_counter <= 0) {
                const HX_INT_PAIR fb = hx_enc_get_frames_bytes(enc);
                toc_counter = hx_xing_toc(xg, fb.a + 1, fb.b + head_bytes);
            }
        }
    }
    // ---- drain: silent frames until every submitted frame is out (tomp3.cpp:1020-1036) ----
    memset(pcm.data(), 0, frame_in);
    while (hx_enc_get_frames(enc) < frames_expected) {
        const HX_IN_OUT x = hx_enc_MP3_audio_encode(enc, pcm.data(), bs.data());
        if (fwrite(bs.data(), 1, x.out_bytes, out) != (size_t) x.out_bytes) { fprintf(stderr, "\n FILE WRITE ERROR\n"); return 1; }
        crc = hx_xing_update_crc((unsigned short) crc, bs.data(), x.out_bytes);
        out_bytes += x.out_bytes;
    }
    // ---- complete the tag in place (tomp3.cpp:1055-1072) ----
    const unsigned frames = hx_enc_get_frames(enc);
    if (xing_flag) {
        const uint64_t samples_audio = audio_bytes / (uint64_t) (wi.channels * (wi.bits / 8));
        hx_xing_update_info(xg, frames, (int) out_bytes, vbr_scale, nullptr, tag.data(), nullptr, nullptr, samples_audio,
                            (unsigned) out_bytes, (unsigned) ec.freq_limit, (unsigned) wi.rate, (unsigned) ec.samprate, (unsigned short) crc);
        if (out == stdout || fseek(out, 0, SEEK_SET) != 0) fprintf(stderr, "\n OUTPUT IS NOT SEEKABLE: TAG FRAME LEFT WITHOUT TOTALS");
        else fwrite(tag.data(), 1, head_bytes, out);
    }
    fprintf(stderr, "\n %u frames, %llu bytes, %.2f kbps\n", frames, (unsigned long long) out_bytes, hx_enc_get_bitrate_float(enc));
    if (out != stdout) fclose(out);
    if (in != stdin) fclose(in);
    hx_xing_destroy(xg);
    hx_enc_destroy(enc);
    return frames == 0 ? 1 : 0;
}
