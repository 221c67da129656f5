// hx_xhead.cpp - the Xing / Info / LAME tag frame that hmp3 writes first in a file
// (reference xhead.c:255-470 XingHeader, :486-690 XingHeaderUpdateInfo,
// :695-719 XingHeaderTOC, :132-182 BuildTOC, :188-231 MusicCRC).  Host-only code.
// Unlike the reference (file-static table, one stream per process) the seek-table state lives in
// a context object, so any number of files can be written concurrently.
#include <cstdint>
#include <cstring>
#include <new>

#include "../../include/hmp3_amd.h"

namespace {

const int kRates[6] = {22050, 24000, 16000, 44100, 48000, 32000};      // index >= 3: MPEG-1
const int kKbps[2][16] = {{0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 0},
                          {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 0}};
enum { F_FRAMES = 1, F_BYTES = 2, F_TOC = 4, F_SCALE = 8, F_RESA = 16, F_RESB = 32, F_INFO = 64 };
const int kPoints = 512;

void put_be32(unsigned char *p, unsigned v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
unsigned get_be32(const unsigned char *p) { return ((unsigned) p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

int side_bytes_of(int mpeg1, int mode) { return mpeg1 ? (mode == 3 ? 17 : 32) : (mode == 3 ? 9 : 17); }

unsigned short crc_step(unsigned short crc, unsigned char d)
{
    // reflected CRC-16, polynomial 0xA001 (= bit-reversed 0x8005), one byte
    unsigned short x = (unsigned short) ((crc ^ d) & 0xFF);
    for (int j = 0; j < 8; j++) x = (x & 1) ? (unsigned short) ((x >> 1) ^ 0xA001) : (unsigned short) (x >> 1);
    return (unsigned short) ((crc >> 8) ^ x);
}

}  // namespace

struct hx_xing {
    int pt[kPoints + 1][2];     // seek points (frames, bytes), decimated 2:1 whenever the table fills
    int npt, every;
};

extern "C" hx_xing *hx_xing_create(void)
{
    hx_xing *x = new (std::nothrow) hx_xing;
    if (x) { memset(x, 0, sizeof(*x)); x->every = 1; }
    return x;
}
extern "C" void hx_xing_destroy(hx_xing *x) { delete x; }

extern "C" unsigned short hx_xing_update_crc(unsigned short crc, const unsigned char *data, int len)
{
    for (int i = 0; i < len; i++) crc = crc_step(crc, data[i]);
    return crc;
}

extern "C" int hx_xing_bitrate_index(int mpeg1, int kbps)
{
    mpeg1 &= 1;
    for (int i = 1; i < 15; i++) if (kKbps[mpeg1][i] == kbps) return i;
    return 0;
}

extern "C" int hx_xing_header(hx_xing *x, int samprate, int h_mode, int cr_bit, int original_bit, int flags, int frames,
                              int bs_bytes, int vbr_scale, const unsigned char *toc, unsigned char *buf,
                              const unsigned char *buf20, const unsigned char *buf20b, int kbps)
{
    memset(x->pt, 0, sizeof(x->pt));
    x->npt = 0;
    x->every = 1;
    h_mode &= 3; cr_bit &= 1; original_bit &= 1; flags &= 127;
    int sri = 0;
    while (sri < 6 && kRates[sri] != samprate) sri++;
    if (sri >= 6) return 0;
    const int mpeg1 = sri >= 3;
    if (mpeg1) sri -= 3;
    const int side = side_bytes_of(mpeg1, h_mode);
    const int bri_cbr = hx_xing_bitrate_index(mpeg1, kbps);
    if (vbr_scale == -1 && kKbps[mpeg1][bri_cbr] < 64) flags &= ~F_TOC;     // CBR at low rates: no room for a TOC
    int need = 4 + side + 8;
    if (flags & F_FRAMES) need += 4;
    if (flags & F_BYTES) need += 4;
    if (flags & F_TOC) need += 100;
    if (flags & F_SCALE) need += 4;
    if (flags & F_RESA) need += 20;
    if (flags & F_RESB) need += 20;
    if (flags & F_INFO) need += 36;
    const int div = mpeg1 ? samprate : 2 * samprate;
    int bri, frame_bytes = 0;
    if (vbr_scale != -1) {      // VBR file: smallest frame that holds the tag
        for (bri = 1; bri < 15; bri++) { frame_bytes = 144000 * kKbps[mpeg1][bri] / div; if (frame_bytes >= need) break; }
        if (bri >= 15) return 0;
    } else {                    // CBR file: the tag frame has the stream's own bitrate
        if (bri_cbr >= 15) return 0;
        frame_bytes = 144000 * kKbps[mpeg1][bri_cbr] / div;
        if (frame_bytes < need) return 0;
        bri = bri_cbr;
    }
    unsigned char *p = buf;
    p[0] = 0xFF;
    p[1] = (unsigned char) (0xF3 | (mpeg1 << 3));
    p[2] = (unsigned char) ((bri << 4) | (sri << 2));
    p[3] = (unsigned char) ((h_mode << 6) | (cr_bit << 3) | (original_bit << 2));
    p += 4;
    memset(p, 0, side);
    p += side;
    memcpy(p, vbr_scale != -1 ? "Xing" : "Info", 4);
    p += 4;
    put_be32(p, (unsigned) flags); p += 4;
    if (flags & F_FRAMES) { put_be32(p, (unsigned) frames); p += 4; }
    if (flags & F_BYTES) { put_be32(p, (unsigned) bs_bytes); p += 4; }
    if (flags & F_TOC) { if (toc) memcpy(p, toc, 100); else memset(p, 0, 100); p += 100; }
    if (flags & F_SCALE) { put_be32(p, (unsigned) vbr_scale); p += 4; }
    if (flags & F_RESA) { if (buf20) memcpy(p, buf20, 20); else memset(p, 0, 20); p += 20; }
    if (flags & F_RESB) { if (buf20) memcpy(p, buf20b, 20); else memset(p, 0, 20); p += 20; }    // (sic: keyed on buf20)
    const int rest = frame_bytes - (int) (p - buf);
    if (rest > 0) memset(p, 0, rest);
    return frame_bytes;
}

extern "C" int hx_xing_toc(hx_xing *x, int frames, int bs_bytes)
{
    x->pt[x->npt][0] = frames;
    x->pt[x->npt][1] = bs_bytes;
    x->npt++;
    if (x->npt < kPoints) return x->every;
    for (int i = 0, k = 1; i < kPoints / 2; i++, k += 2) { x->pt[i][0] = x->pt[k][0]; x->pt[i][1] = x->pt[k][1]; }
    x->npt = kPoints / 2;
    x->every += x->every;
    return x->every;
}

// 100 seek bytes: byte i = 256 * (file offset at i % of the frames) / file size, linear between points
static void build_toc(hx_xing *x, int tot_frames, int tot_bytes, unsigned char *out)
{
    if (tot_frames <= 0 || tot_bytes <= 0) { memset(out, 0, 100); return; }
    x->pt[x->npt][0] = tot_frames;
    x->pt[x->npt][1] = tot_bytes;
    x->npt++;
    for (int i = 0; i < x->npt; i++) x->pt[i][0] *= 100;
    const double a = 256.0 / tot_bytes;
    int target = 0, f0 = 0, b0 = 0, k = 0;
    for (int i = 0; i < 100; i++) {
        while (x->pt[k][0] <= target) { f0 = x->pt[k][0]; b0 = x->pt[k][1]; k++; }
        const double b = b0 + ((double) (target - f0)) * ((double) (x->pt[k][1] - b0)) / ((double) (x->pt[k][0] - f0));
        int idx = (int) (a * b + 0.5);
        if (idx < 0) idx = 0;
        if (idx > 255) idx = 255;
        out[i] = (unsigned char) idx;
        target += tot_frames;
    }
}

extern "C" int hx_xing_update_info(hx_xing *x, unsigned frames, int bs_bytes, int vbr_scale, const unsigned char *toc,
                                   unsigned char *buf, const unsigned char *buf20, const unsigned char *buf20b,
                                   unsigned long long samples_audio, unsigned bytes_mp3, unsigned lowpass,
                                   unsigned in_samplerate, unsigned out_samplerate, unsigned short musiccrc)
{
    unsigned char *const start = buf;
    const int mpeg1 = (buf[1] >> 3) & 1, h_mode = (buf[3] >> 6) & 3;
    const int spf = mpeg1 ? 1152 : 576;
    if (in_samplerate == 0 || out_samplerate == 0) in_samplerate = out_samplerate = 1;
    const int pad_start = 1680;         // encoder delay in samples
    int pad_end = 0;
    if (samples_audio > 0) {
        const uint64_t audio_r = (uint64_t) ((double) samples_audio * ((double) out_samplerate / (double) in_samplerate) + 0.5);
        uint64_t mp3 = (uint64_t) frames * spf;
        if (mp3 - audio_r - pad_start >= 4096) {        // the 12-bit padding field would overflow: trim the frame count
            frames = (unsigned) ((audio_r + pad_start + 1152) / spf);
            mp3 = frames * spf;                          // (32-bit product, as in the reference)
        }
        pad_end = (int) ((long) (mp3 - audio_r)) - pad_start;
    }
    buf += 4 + side_bytes_of(mpeg1, h_mode);
    if (memcmp(buf, vbr_scale != -1 ? "Xing" : "Info", 4) != 0) return 0;
    buf += 4;
    const int flags = (int) get_be32(buf);
    buf += 4;
    if (flags & F_FRAMES) { put_be32(buf, frames); buf += 4; }
    if (flags & F_BYTES) { put_be32(buf, (unsigned) bs_bytes); buf += 4; }
    if (flags & F_TOC) { if (toc) memcpy(buf, toc, 100); else build_toc(x, (int) frames, bs_bytes, buf); buf += 100; }
    if (flags & F_SCALE) { put_be32(buf, (unsigned) vbr_scale); buf += 4; }
    if (flags & F_RESA) { if (buf20) memcpy(buf, buf20, 20); else memset(buf, 0, 20); buf += 20; }
    if (flags & F_RESB) { if (buf20) memcpy(buf, buf20b, 20); else memset(buf, 0, 20); buf += 20; }
    if ((flags & F_INFO) && samples_audio != 0) {
        memcpy(buf, "LAMEH5.24", 9); buf += 9;          // short version string the reference writes
        *buf++ = (unsigned char) ((0 << 4) | (vbr_scale != -1 ? 0 : 1));     // tag revision 0 | method: unknown / CBR
        *buf++ = (unsigned char) ((lowpass / 100) & 0xFF);
        put_be32(buf, 0); buf += 4;                     // ReplayGain peak
        put_be32(buf, 0); buf += 4;                     // ReplayGain radio / audiophile
        *buf++ = 0;                                     // encoding flags, ATH type
        *buf++ = 0;                                     // bitrate: unknown
        buf[0] = (unsigned char) ((pad_start >> 4) & 0xFF);
        buf[1] = (unsigned char) (((pad_start << 4) & 0xF0) | ((pad_end >> 8) & 0x0F));
        buf[2] = (unsigned char) (pad_end & 0xFF);
        buf += 3;
        unsigned char misc = 0x1C;
        if (in_samplerate == 44100) misc |= 0x40;
        else if (in_samplerate == 48000) misc |= 0x80;
        else if (in_samplerate > 48000) misc |= 0xC0;
        *buf++ = misc;
        *buf++ = 0;                                     // MP3 gain
        *buf++ = 0; *buf++ = 0;                         // preset, surround
        put_be32(buf, bytes_mp3); buf += 4;             // music length
        *buf++ = (unsigned char) (musiccrc >> 8); *buf++ = (unsigned char) musiccrc;
        unsigned short tagcrc = 0;
        for (const unsigned char *q = start; q < buf; q++) tagcrc = crc_step(tagcrc, *q);
        *buf++ = (unsigned char) (tagcrc >> 8); *buf++ = (unsigned char) tagcrc;
    }
    return 1;
}
