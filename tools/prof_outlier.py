"""Phase profile of the slowest streams of the bench's config-n signal set against the batch mean (library built with
HX_EXTRA=-DHX_PROFILE).  python tools/prof_outlier.py [S] [F] [cfg] [calls]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hmp3_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
CFG = int(sys.argv[3]) if len(sys.argv) > 3 else 2
CALLS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
w = bench.workload(CFG)
kw, sr = w["classes"][0]
rho = w["rho"]
pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [rho[i % len(rho)] for i in range(S)], w["bursts"], dev)
st = torch.cuda.current_stream().cuda_stream
b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
b.debug_enable(True)
stride = b.out_stride(F)
out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
acc = np.zeros((S, 64))
for c in range(CALLS):
    b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
    torch.cuda.synchronize()
    if c >= 1:
        acc += b.debug_read("prof", np.uint64, S * 64).reshape(S, 64).astype(np.float64)
acc /= (CALLS - 1) * F
names = {1: "startup", 3: "seek_actual", 5: "scale_factors", 6: "big_lucky", 8: "count_bits", 9: "increase_bits", 10: "decrease_bits", 11: "inverse_sf2", 12: "bitallo total",
         13: "hand-over", 17: "flush+side", 20: "#seek sweeps", 21: "#lucky passes", 22: "#count_bits", 31: "kernel total"}
tot = acc[:, 31]
order = np.argsort(-tot)
print("per-stream ticks/frame: min %.0f mean %.0f p99 %.0f max %.0f" % (tot.min(), tot.mean(), np.percentile(tot, 99), tot.max()))
print("%-16s %10s %10s %10s %10s %10s" % ("phase", "mean", "slowest", "2nd", "3rd", "p50 stream"))
mid = order[S // 2]
for k in sorted(names):
    print("%-16s %10.1f %10.1f %10.1f %10.1f %10.1f" % (names[k], acc[:, k].mean(), acc[order[0], k], acc[order[1], k], acc[order[2], k], acc[mid, k]))
print("slowest stream ids:", order[:8].tolist(), " their totals:", [int(x) for x in tot[order[:8]]])
