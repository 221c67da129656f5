"""A/B helper: the bench's worst-case pass only (config 2 signals with the correlation cycled over {0.7, 0, 1, 0.3}), K6 ms and frames/s.
HMP3AMD_LIB selects the library.  python tools/ab_worst.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hmp3_amd import api
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
w = bench.workload(2)
kw, sr = w["classes"][0]
S, F = w["S"], w["F"]
pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [bench.RHO_CYCLE[i % 4] for i in range(S)], False, dev)
b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
stride = b.out_stride(F)
outs = [torch.empty((S, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
nbs = [torch.zeros((S,), dtype=torch.int32, device=dev) for _ in range(2)]
st = torch.cuda.current_stream().cuda_stream
def go(n):
    for i in range(n):
        b.submit_device(pcm.data_ptr(), F, outs[i & 1].data_ptr(), stride, nbs[i & 1].data_ptr(), st)
    b.wait(st); torch.cuda.synchronize()
go(2); b.alloc_kernel_ms()
t0 = time.perf_counter(); go(steps); dt = time.perf_counter() - t0
ms, n = b.alloc_kernel_ms()
print("worst case: %.3f M frames/s  %.3f ms/step  K6 %.3f ms  status %d  bytes %d" % (S * F * steps / dt / 1e6, dt / steps * 1e3, ms, b.status(), int(nbs[(steps - 1) & 1].sum().item())))
