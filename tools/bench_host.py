"""PCIe-inclusive throughput of the headline workload with host buffers: plain synchronous calls against
pipelined ones (hx_batch_submit_s16_host, page-locked memory).  python tools/bench_host.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from hmp3_amd import api  # noqa: E402

S, F = 1024, 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
pcm = bench.synth_batch_gpu(torch, S, F, 44100, dev).cpu().pin_memory()
b = api.Batch(api.default_control(bitrate=64, short_block_threshold=99999), nstreams=S, max_frames=F)
stride = b.out_stride(F)
outs = [torch.zeros((S, stride), dtype=torch.uint8).pin_memory() for _ in range(2)]
nbs = [torch.zeros((S,), dtype=torch.int32).pin_memory() for _ in range(2)]
res = {}
for mode in ("plain", "pipelined"):
    for timed in (0, 1):
        n = steps if timed else 2
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if mode == "plain":
                lib = api.lib()
                lib.hx_batch_encode_s16_host(b.h, pcm.data_ptr(), F, outs[i & 1].data_ptr(), stride, nbs[i & 1].data_ptr())
            else:
                b.submit_host(pcm.data_ptr(), F, outs[i & 1].data_ptr(), stride, nbs[i & 1].data_ptr())
        if mode == "pipelined":
            b.wait_host()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[mode] = {"frames_per_s": round(S * F * steps / dt), "ms_per_step": round(1e3 * dt / steps, 2)}
res["bytes_per_step"] = {"pcm_in": int(pcm.numel() * 2), "out_buffer": int(S * stride)}
print(json.dumps(res))
