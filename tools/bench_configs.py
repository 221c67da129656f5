"""Throughput of the other BASELINE.json configurations (parity-test cases, not bench lines) on one
GPU: python tools/bench_configs.py.  Signals come from hmp3_amd/synth.py (16 distinct streams tiled
over the batch; bursts where short blocks are wanted); PCM resident in HBM; 4 timed steps of pipelined calls."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hmp3_amd import api, synth  # noqa: E402

CASES = [
    ("config2  1024 x 256  44.1k CBR-128 long blocks", 1024, 256, [dict(bitrate=64, short_block_threshold=99999)], 44100, False, 2),
    ("config3  4096 x 256  44.1k VBR-50, block switching, bursts", 4096, 256, [dict()], 44100, True, 2),
    ("config4  4096 x 128  48k -V100 -HF2 -F19000 (one GPU's share)", 4096, 128, [dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000)], 48000, True, 2),
    ("config5  4096 x 256  32/44.1/48k mixed, CBR-128 (one GPU's share)", 4096, 256,
     [dict(bitrate=64, samprate=32000), dict(bitrate=64), dict(bitrate=64, samprate=48000)], 44100, True, 2),
    ("mono     1024 x 256  44.1k CBR-64 mono, block switching", 1024, 256, [dict(bitrate=64, mode=3)], 44100, True, 1),
    ("mpeg2    1024 x 256  22.05k CBR-64 joint stereo, block switching", 1024, 256, [dict(bitrate=32, samprate=22050)], 22050, True, 2),
]


def run(name, S, F, kws, sr, bursts, nch):
    dev = torch.device("cuda", 0)
    uniq = np.stack([synth.stream_pcm(700 + i, F, sr=sr, rho=[0.7, 0.0, 1.0, 0.3][i % 4], bursts=bursts) for i in range(16)])
    if nch == 1:
        uniq = uniq[:, :, :1]
    pcm = torch.from_numpy(np.ascontiguousarray(uniq)).to(dev).repeat((S + 15) // 16, 1, 1)[:S].contiguous()
    ctl = [api.default_control(**kws[i % len(kws)]) for i in range(S)] if len(kws) > 1 else api.default_control(**kws[0])
    b = api.Batch(ctl, nstreams=S, max_frames=F)
    stride = b.out_stride(F)
    out = torch.empty((S, stride), dtype=torch.uint8, device=dev)
    nb = torch.zeros((S,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(1):
        b.submit_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
    b.wait(st)
    torch.cuda.synchronize()
    b.alloc_kernel_ms()
    t0 = time.perf_counter()
    steps = 4
    for _ in range(steps):
        b.submit_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)     # pipelined calls, as bench.py makes them
    b.wait(st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms, _ = b.alloc_kernel_ms()
    res = {"case": name, "frames_per_s": round(S * F * steps / dt), "ms_per_step": round(1e3 * dt / steps, 2), "k_alloc_ms": round(k_ms, 2),
           "x_realtime": round(S * F * steps / dt * 1152 / sr), "bytes_per_frame": round(float(nb.sum().item()) / (S * F), 1), "status": b.status()}
    b.close()
    del pcm, out
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    for c in CASES:
        print(json.dumps(run(*c)), flush=True)
