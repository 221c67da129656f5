"""File-level golden vectors for the CLI: WAVs synthesised by hmp3_amd/synth.py are
encoded by the REAL reference CLI (oracle/_ref/hmp3, built by `make -C oracle ref`) and the complete
.mp3 files (tag frame included) are committed under tests/golden/.  Run in the build container:
    python tests/golden/make_golden_cli.py
The WAVs themselves are not committed: tests regenerate them from the same seeds."""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hmp3_amd import synth  # noqa: E402

CASES = {
    # name: (seed, samples (deliberately not a multiple of 1152), sample rate, float WAV?, bursts, CLI flags)
    "cli_cbr128_s16_44k": (901, 150000, 44100, False, False, ["-B64"]),
    "cli_vbr75_f32_48k_hf": (902, 120011, 48000, True, True, ["-V75", "-HF2", "-F19000"]),
    "cli_cbr192_s16_32k_x1_dc": (903, 70001, 32000, False, True, ["-B96", "-X1", "-S1", "-M0"]),
    "cli_vbr50_s24_44k": (904, 60007, 44100, 24, False, []),
    "cli_cbr128_u8_44k": (905, 50003, 44100, 8, False, ["-B64"]),
    "cli_vbr50_s32_48k": (906, 50003, 48000, 32, False, ["-V50"]),
    # mono WAVs (mode 3): the left channel of the synthetic stream
    "cli_mono_cbr64_s16_44k": (907, 90001, 44100, False, True, ["-B64"]),
    "cli_mono_vbr60_f32_48k": (908, 60013, 48000, True, True, ["-V60"]),
    # stereo file encoded as mono (-M3: down-mix)
    "cli_downmix_vbr50_s16_44k": (909, 60001, 44100, False, True, ["-M3"]),
    "cli_downmix_cbr64_s24_48k": (910, 50021, 48000, 24, False, ["-M3", "-B64"]),
    # MPEG-2 LSF rates (one granule per frame, "Info"/"Xing" tag in the MPEG-2 layout)
    "cli_lsf_cbr64_s16_22k": (911, 70001, 22050, False, True, ["-B32"]),
    "cli_lsf_vbr50_f32_24k": (912, 60013, 24000, True, True, ["-V50"]),
    "cli_lsf_mono_cbr24_s16_16k": (913, 50021, 16000, False, True, ["-B24"]),
    "cli_lsf_downmix_vbr80_s24_22k": (914, 40009, 22050, 24, False, ["-M3", "-V80"]),
    # other containers / header variants the reference parses (pcmhpm.c:203-429)
    "cli_rifx_cbr64_s16_44k": (915, 30011, 44100, False, False, ["-B64"]),            # big-endian fields and samples
    "cli_rf64_vbr50_s16_48k": (916, 30013, 48000, False, True, []),                   # ds64 chunk, data size 0xFFFFFFFF
    "cli_w64_cbr64_s24_44k": (917, 30001, 44100, 24, False, ["-B64"]),                # Sony Wave64 GUID chunks
    "cli_ext_vbr60_s24_48k": (918, 30007, 48000, 24, False, ["-V60"]),                # WAVE_FORMAT_EXTENSIBLE + a LIST chunk
    "cli_rifx_cbr64_s24_44k": (931, 200003, 44100, 24, False, ["-B64"]),            # big-endian 24-bit, more than 1 MiB of samples (a streaming reader's pieces must not tear a sample)
    "cli_odd_cbr64_u8_mono_44k": (919, 30001, 44100, 8, False, ["-B64"]),             # odd data size: the pad byte counts
    # sample-rate conversion in front of the encoder (Csrc cases 1-4; -A picks the encode rate)
    "cli_src_11k_to_22k_s16": (920, 20001, 11025, False, True, ["-B32"]),                 # 1:2 up
    "cli_src_8k_to_16k_u8_mono": (921, 20003, 8000, 8, False, ["-B24"]),                  # 1:2 up, mono, 8-bit
    "cli_src_32k_to_44k_f32": (922, 30011, 32000, True, False, ["-A44100", "-V60"]),      # up by 441:320, linear
    "cli_src_48k_to_24k_s24": (923, 40001, 48000, 24, True, ["-A2", "-B32"]),             # 2:1 down, one filter bank
    "cli_src_44k_to_32k_s16": (924, 40003, 44100, False, True, ["-A32000", "-B64"]),      # 441:320 down, two stages
    "cli_src_44k_to_22k_downmix": (925, 40007, 44100, False, False, ["-A22050", "-M3", "-V50"]),   # down + down-mix
    # the reference appends its four calls' worth of silence only if that fits a limit that depends on the sample format
    # (tomp3.cpp:925): 32-bit samples converted down by more than 1.14, or 24-bit by more than 2.03, never get it, and the
    # tail of the input that is shorter than one call's need is dropped
    "cli_src_44k_to_16k_f32_nopad": (932, 20011, 44100, True, True, ["-A16000", "-B32"]),
    "cli_src_48k_to_22k_s24_nopad": (933, 30011, 48000, 24, False, ["-A22050", "-V60"]),
    "cli_src_24k_to_22k_s32_nopad": (934, 14301, 24000, 32, True, ["-A22050", "-B128", "-M1", "-N16"]),
    # the reference's first-generation allocator: intensity stereo (MPEG-2 below 48 kbps total; MPEG-1 on request, -N), dual channel (-M2)
    "cli_is_lsf_cbr32_s16_22k": (926, 50003, 22050, False, True, ["-B16"]),
    "cli_is_lsf_cbr16_f32_16k": (927, 40001, 16000, True, False, ["-B8"]),
    "cli_is_n8_cbr128_s16_44k": (928, 60007, 44100, False, True, ["-B64", "-N8"]),
    "cli_dual_cbr128_s24_44k": (929, 50021, 44100, 24, True, ["-B64", "-M2"]),
    "cli_dual_lsf_cbr48_s16_24k": (930, 40009, 24000, False, False, ["-B24", "-M2"]),
}
CONTAINER = {"cli_rifx_cbr64_s16_44k": "rifx", "cli_rifx_cbr64_s24_44k": "rifx", "cli_rf64_vbr50_s16_48k": "rf64", "cli_w64_cbr64_s24_44k": "w64",
             "cli_ext_vbr60_s24_48k": "ext"}
MONO = {"cli_odd_cbr64_u8_mono_44k", "cli_src_8k_to_16k_u8_mono", "cli_mono_cbr64_s16_44k", "cli_mono_vbr60_f32_48k", "cli_lsf_mono_cbr24_s16_16k"}


def write_container(path, kind, fmt_tag, nch, sr, bits, data):
    """the same sample bytes in one of the other containers the reference reads"""
    be = kind == "rifx"
    e = ">" if be else "<"
    bps = bits // 8
    if be and bps > 1:
        data = b"".join(data[i:i + bps][::-1] for i in range(0, len(data), bps))
    fmt = struct.pack(e + "HHIIHH", fmt_tag, nch, sr, sr * nch * bps, nch * bps, bits)
    if kind == "ext":
        guid = struct.pack("<H", fmt_tag) + bytes([0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xaa, 0x00, 0x38, 0x9b, 0x71])
        fmt = struct.pack("<HHIIHH", 0xFFFE, nch, sr, sr * nch * bps, nch * bps, bits) + struct.pack("<HHI", 22, bits, 3) + guid
    pad = b"\0" if len(data) & 1 else b""
    with open(path, "wb") as f:
        if kind == "w64":
            tail = bytes([0xF3, 0xAC, 0xD3, 0x11, 0x8C, 0xD1, 0x00, 0xC0, 0x4F, 0x8E, 0xDB, 0x8A])
            riff = bytes([0x72, 0x69, 0x66, 0x66, 0x2E, 0x91, 0xCF, 0x11, 0xA5, 0xD6, 0x28, 0xDB, 0x04, 0xC1, 0x00, 0x00])
            p8 = lambda b: b + b"\0" * (-len(b) % 8)
            body = b"wave" + tail + b"fmt " + tail + struct.pack("<Q", 24 + len(fmt)) + p8(fmt) + b"data" + tail + struct.pack("<Q", 24 + len(data)) + p8(data)
            f.write(riff + struct.pack("<Q", 24 + len(body)) + body)
        elif kind == "rf64":
            ds64 = b"ds64" + struct.pack("<IQQQI", 28, 0, len(data), len(data) // (nch * bps), 0)
            f.write(b"RF64" + struct.pack("<I", 0xFFFFFFFF) + b"WAVE" + ds64 + b"fmt " + struct.pack("<I", len(fmt)) + fmt)
            f.write(b"data" + struct.pack("<I", 0xFFFFFFFF) + data + pad)
        else:
            extra = b"LIST" + struct.pack(e + "I", 5) + b"INFOx\0" if kind == "ext" else b""     # an odd-sized chunk to skip
            total = 4 + 8 + len(fmt) + len(extra) + 8 + len(data) + len(pad)
            f.write((b"RIFX" if be else b"RIFF") + struct.pack(e + "I", total) + b"WAVE")
            f.write(b"fmt " + struct.pack(e + "I", len(fmt)) + fmt + extra)
            f.write(b"data" + struct.pack(e + "I", len(data)) + data + pad)


def write_wav(path, pcm_i16, sr, as_float, container=None):
    n = pcm_i16.shape[0]
    if container:       # integer PCM only
        bits = as_float if as_float in (8, 24, 32) else 16
        rng = np.random.default_rng(n)
        if bits == 16:
            data = pcm_i16.astype("<i2").tobytes()
        elif bits == 24:
            v = (pcm_i16.astype(np.int32) << 8) + rng.integers(0, 256, pcm_i16.shape)
            b = v.astype("<i4").tobytes()
            data = b"".join(b[i:i + 3] for i in range(0, len(b), 4))
        else:
            raise ValueError(bits)
        write_container(path, container, 1, pcm_i16.shape[1] if pcm_i16.ndim == 2 else 1, sr, bits, data)
        return n
    if pcm_i16.ndim == 1 and as_float == 8:     # mono 8-bit unsigned (an odd sample count leaves an odd data size)
        data = ((pcm_i16.astype(np.int32) >> 8) + 128).astype(np.uint8).tobytes()
        fmt = struct.pack("<HHIIHH", 1, 1, sr, sr, 1, 8)
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data) + (len(data) & 1)) + b"WAVE")
            f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
            f.write(b"data" + struct.pack("<I", len(data)) + data + (b"\x80" if len(data) & 1 else b""))
        return n
    if pcm_i16.ndim == 1:       # mono: 16-bit or float only
        data = (pcm_i16.astype(np.float32) / 32768.0).astype("<f4").tobytes() if as_float is True else pcm_i16.astype("<i2").tobytes()
        bps = 4 if as_float is True else 2
        fmt = struct.pack("<HHIIHH", 3 if as_float is True else 1, 1, sr, sr * bps, bps, 8 * bps)
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE")
            f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
            f.write(b"data" + struct.pack("<I", len(data)) + data)
        return n
    if as_float is True:
        data = (pcm_i16.astype(np.float32) / 32768.0).astype("<f4").tobytes()
        fmt = struct.pack("<HHIIHH", 3, 2, sr, sr * 8, 8, 32)
    elif as_float in (8, 24, 32):       # integer PCM of that width; low bits filled so they matter
        rng = np.random.default_rng(n)
        if as_float == 8:
            data = ((pcm_i16.astype(np.int32) >> 8) + 128).astype(np.uint8).tobytes()
        elif as_float == 24:
            v = (pcm_i16.astype(np.int32) << 8) + rng.integers(0, 256, pcm_i16.shape)
            b = v.astype("<i4").tobytes()
            data = b"".join(b[i:i + 3] for i in range(0, len(b), 4))
        else:
            data = ((pcm_i16.astype(np.int64) << 16) + rng.integers(0, 65536, pcm_i16.shape)).astype("<i4").tobytes()
        fmt = struct.pack("<HHIIHH", 1, 2, sr, sr * 2 * as_float // 8, 2 * as_float // 8, as_float)
    else:
        data = pcm_i16.astype("<i2").tobytes()
        fmt = struct.pack("<HHIIHH", 1, 2, sr, sr * 4, 4, 16)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"data" + struct.pack("<I", len(data)) + data)
    return n


def case_pcm(name):
    seed, nsamp, sr, as_float, bursts, flags = CASES[name]
    nfr = (nsamp + 1151) // 1152
    pcm = synth.stream_pcm(seed, nfr, sr=sr, rho=0.5, bursts=bursts)[:nsamp]
    return pcm[:, 0].copy() if name in MONO else pcm


# batch mode: these stereo inputs are encoded together by `hmp3amd -batch ... -V60 -HF2`; the reference encodes them one by one
BATCH_FLAGS = ["-V60", "-HF2"]
BATCH_INPUTS = ["cli_cbr128_s16_44k", "cli_vbr75_f32_48k_hf", "cli_cbr192_s16_32k_x1_dc", "cli_vbr50_s24_44k", "cli_cbr128_u8_44k", "cli_vbr50_s32_48k"]
# the same for MPEG-2 rates (a batch holds MPEG-1 or MPEG-2 streams, not both)
BATCH_LSF_FLAGS = ["-B40"]
BATCH_LSF_INPUTS = ["cli_lsf_cbr64_s16_22k", "cli_lsf_vbr50_f32_24k"]


if __name__ == "__main__":
    ref = os.path.join(ROOT, "oracle", "_ref", "hmp3")
    gold = os.path.join(ROOT, "tests", "golden")
    meta = {}
    for name, (seed, nsamp, sr, as_float, bursts, flags) in CASES.items():
        with tempfile.TemporaryDirectory() as d:
            wav, mp3 = os.path.join(d, "in.wav"), os.path.join(d, "out.mp3")
            write_wav(wav, case_pcm(name), sr, as_float, CONTAINER.get(name))
            subprocess.run([ref, wav, mp3] + flags, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            data = open(mp3, "rb").read()
        open(os.path.join(gold, name + ".mp3"), "wb").write(data)
        meta[name] = {"bytes": len(data), "flags": flags}
        print(name, len(data), "bytes")
    for name in BATCH_INPUTS:
        seed, nsamp, sr, as_float, bursts, flags = CASES[name]
        with tempfile.TemporaryDirectory() as d:
            wav, mp3 = os.path.join(d, "in.wav"), os.path.join(d, "out.mp3")
            write_wav(wav, case_pcm(name), sr, as_float)
            subprocess.run([ref, wav, mp3] + BATCH_FLAGS, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            data = open(mp3, "rb").read()
        open(os.path.join(gold, "batch_" + name[4:] + ".mp3"), "wb").write(data)
        meta["batch_" + name[4:]] = {"bytes": len(data), "flags": BATCH_FLAGS}
        print("batch_" + name[4:], len(data), "bytes")
    for name in BATCH_LSF_INPUTS:
        seed, nsamp, sr, as_float, bursts, flags = CASES[name]
        with tempfile.TemporaryDirectory() as d:
            wav, mp3 = os.path.join(d, "in.wav"), os.path.join(d, "out.mp3")
            write_wav(wav, case_pcm(name), sr, as_float)
            subprocess.run([ref, wav, mp3] + BATCH_LSF_FLAGS, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            data = open(mp3, "rb").read()
        open(os.path.join(gold, "batch_" + name[4:] + ".mp3"), "wb").write(data)
        meta["batch_" + name[4:]] = {"bytes": len(data), "flags": BATCH_LSF_FLAGS}
        print("batch_" + name[4:], len(data), "bytes")
    json.dump(meta, open(os.path.join(gold, "cli.json"), "w"), indent=1)
