"""Per-stream phase profile of k_alloc on the bench's own signal set (1024 distinct streams): where the slowest streams
spend their time.  Library built with HX_EXTRA=-DHX_PROFILE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hmp3_amd import api
S, F = 1024, int(os.environ.get("PF", "256"))
dev = torch.device("cuda:0")
CFG = int(os.environ.get("PF_CFG", "2"))         # 2: CBR-128 long blocks; 3: VBR-50 with block switching (1024 of its streams)
w = bench.workload(CFG)
kw, sr = w["classes"][0]
rho_list = [float(os.environ["PF_RHO"])] if "PF_RHO" in os.environ else w["rho"]     # PF_RHO=1.0: the loud near-mono case
pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [rho_list[i % len(rho_list)] for i in range(S)], w["bursts"], dev)
st = torch.cuda.current_stream().cuda_stream
b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
b.debug_enable(True)
stride = b.out_stride(F)
out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
for c in range(3):
    b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
torch.cuda.synchronize()
prof = b.debug_read("prof", np.uint64, S * 64).reshape(S, 64).astype(np.float64)
names = {0: "join wait", 1: "startup", 2: "seek_initial", 3: "seek_actual", 4: "trade_dual", 5: "scale_factors", 6: "big_lucky", 7: "do_quant", 8: "quant+count", 9: "increase_bits",
         10: "decrease_bits", 11: "inverse_sf2", 12: "bitallo total", 13: "hand-over", 17: "placement", 20: "#sweeps", 21: "#lucky passes", 22: "#counts", 31: "kernel total",
         23: "lucky: setup", 24: "lucky: terms", 25: "lucky: sums", 26: "lucky: replay", 27: "sweep: publish", 28: "sweep: lines", 29: "sweep: sums",
         36: "cnt: ballots", 37: "cnt: j2/j3", 38: "cnt: regions", 39: "cnt: pairs", 40: "cnt: quads", 41: "cnt: reduce", 42: "q+c: post", 43: "q+c: quant", 44: "q+c: join", 30: "seek: join wait", 45: "#sweeps helper", 14: "frame budget", 15: "gr: pre", 16: "gr: fetch post", 18: "retire", 19: "pack_sf", 46: "pl: granule tail", 47: "pl: sizes", 48: "pl: slot", 49: "pl: head+gr copy"}
order = np.argsort(prof[:, 31])
top = order[-10:]
print("streams by kernel total (k cycles per frame): min %.0f  mean %.0f  p95 %.0f  p99 %.0f  max %.0f" % tuple(x / F / 1e3 for x in (
    prof[:, 31].min(), prof[:, 31].mean(), np.percentile(prof[:, 31], 95), np.percentile(prof[:, 31], 99), prof[:, 31].max())))
print("%-16s %10s %10s   slowest streams: %s" % ("per frame", "mean", "slowest10", " ".join(str(int(s)) for s in top[::-1])))
for k in sorted(names):
    unit = 1.0 if k >= 20 and k <= 22 else 1.0
    print("%-16s %10.0f %10.0f   %s" % (names[k], prof[:, k].mean() / F, prof[top, k].mean() / F, " ".join("%6.0f" % (prof[s, k] / F) for s in top[::-1][:6])))
