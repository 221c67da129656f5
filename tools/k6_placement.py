"""Where do the workgroups of a K6 launch land?  Pipelined submits at config 2; per launch the (XCC, SE, SH, CU) of workgroups
0 .. 23 of the launch order (= the streams that ran longest in the previous call) - are they the same CUs every launch?
  python tools/k6_placement.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hmp3_amd import api
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
w = bench.workload(2)
kw, sr = w["classes"][0]
S, F = w["S"], w["F"]
pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [0.7] * S, False, dev)
st = torch.cuda.current_stream().cuda_stream
b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
stride = b.out_stride(F)
outs = [torch.empty((S, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
nbs = [torch.zeros((S,), dtype=torch.int32, device=dev) for _ in range(2)]
def fmt(v):
    return "x%d.se%d.sh%d.cu%02d" % (v >> 16, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 15)
hist = []
for mode in ("plain", "submit"):
    for c in range(steps):
        f = b.encode_device if mode == "plain" else b.submit_device
        f(pcm.data_ptr(), F, outs[c & 1].data_ptr(), stride, nbs[c & 1].data_ptr(), st)
        if mode == "submit":
            b.wait(st)          # (drains the pipeline: placement of this launch only)
        torch.cuda.synchronize()
        pl = b.debug_read("place", np.uint32, S)
        hist.append(pl.copy())
        print("%-6s launch %d: wg 0..11 ->" % (mode, c), " ".join(fmt(int(v)) for v in pl[:12]), flush=True)
h = np.array(hist)
same = (h[1:] == h[:1]).all(axis=0)
print("workgroups whose CU is the same in every launch: %d of %d; of the first 16: %d; of the first 64: %d" % (same.sum(), S, same[:16].sum(), same[:64].sum()))
cus = {}
for i, v in enumerate(h[-1]):
    cus.setdefault(int(v) & 0xFFFFFF00 | 0, []).append(i)
print("distinct (xcc, se, sh, cu) in the last launch: %d; workgroups per CU: min %d max %d" % (len(cus), min(len(x) for x in cus.values()), max(len(x) for x in cus.values())))
k = [sorted(v) for v in cus.values() if 0 in v][0]
print("the CU of workgroup 0 also runs workgroups", k)
