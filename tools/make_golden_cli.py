"""File-level golden vectors for the CLI: WAVs synthesised by hmp3_amd/synth.py are
encoded by the REAL reference CLI (oracle/_ref/hmp3, built by `make -C oracle ref`) and the complete
.mp3 files (tag frame included) are committed under tests/golden/.  Run in the build container:
    python tools/make_golden_cli.py
The WAVs themselves are not committed: tests regenerate them from the same seeds."""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
}
MONO = {"cli_mono_cbr64_s16_44k", "cli_mono_vbr60_f32_48k", "cli_lsf_mono_cbr24_s16_16k"}


def write_wav(path, pcm_i16, sr, as_float):
    n = pcm_i16.shape[0]
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
            write_wav(wav, case_pcm(name), sr, as_float)
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
