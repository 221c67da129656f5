"""Who is the tail?  Per-stream K6 durations of a 256-frame launch (plain calls), the slowest streams by id, and a histogram.
  python tools/k6_tail.py [S] [F] [cfg]        (AB_RHO=0.7,0,1,0.3 for the worst-case signal set)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hmp3_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
CFG = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
w = bench.workload(CFG)
kw, sr = w["classes"][0]
rho = [float(x) for x in os.environ["AB_RHO"].split(",")] if os.environ.get("AB_RHO") else w["rho"]
pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [rho[i % len(rho)] for i in range(S)], w["bursts"], dev)
st = torch.cuda.current_stream().cuda_stream
b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
stride = b.out_stride(F)
out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
ds = []
for c in range(5):
    b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
    torch.cuda.synchronize()
    if c >= 2:
        ds.append(b.debug_read("dur", np.uint32, S).astype(np.float64) / 1e5)
d = np.mean(ds, axis=0)
o = np.argsort(-d)
print("S=%d F=%d cfg %d rho %s: stream ms min %.3f mean %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f" % (S, F, CFG, rho, d.min(), d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max()))
print("slowest:", ", ".join("%d: %.2f" % (i, d[i]) for i in o[:16]))
h, e = np.histogram(d, bins=24)
for k in range(len(h)):
    print("  %6.2f - %6.2f ms  %5d %s" % (e[k], e[k + 1], h[k], "#" * int(60.0 * h[k] / h.max())))
bytes_ = nb.cpu().numpy()
print("bytes of the slowest:", bytes_[o[:8]].tolist(), " median stream:", int(np.median(bytes_)))
pk = pcm.abs().amax(dim=(1, 2)).cpu().numpy(); rms = pcm.float().pow(2).mean(dim=(1, 2)).sqrt().cpu().numpy()
print("rms of the slowest:", [int(x) for x in rms[o[:8]]], " median rms:", int(np.median(rms)), " min rms:", int(rms.min()))
print("rank correlation duration / rms: %.3f" % np.corrcoef(np.argsort(np.argsort(d)), np.argsort(np.argsort(rms)))[0, 1])
