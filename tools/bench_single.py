"""Frames per second of the one-stream CMp3Enc-shaped API (hx_enc_*): one 1152-sample block per call, host buffers,
one HIP-graph launch per call (HMP3AMD_ENC_GRAPH=0: plain calls - eight kernel launches and two PCIe copies).  python tools/bench_single.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hmp3_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pcm = synth.stream_pcm(1, 256)
e = api.Mp3Enc()
assert e.MP3_audio_encode_init(api.default_control(bitrate=64)) > 0
for f in range(64):
    e.MP3_audio_encode(pcm[(f % 256) * 1152:(f % 256 + 1) * 1152])
t0 = time.perf_counter()
nbytes = 0
for f in range(n):
    nbytes += len(e.MP3_audio_encode(pcm[(f % 256) * 1152:(f % 256 + 1) * 1152])[1])
dt = time.perf_counter() - t0
print("hx_enc_MP3_audio_encode: %d frames in %.3f s = %.0f frames/s (%.1f x real time at 44.1 kHz), %d bytes" % (n, dt, n / dt, n / dt * 1152 / 44100, nbytes))
