import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from hmp3_amd import api
S, F = 1024, 256
dev = torch.device("cuda:0")
w = bench.workload(3)
kw, sr = w["classes"][0]
pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [w["rho"][i % len(w["rho"])] for i in range(S)], w["bursts"], dev)
st = torch.cuda.current_stream().cuda_stream
b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
b.debug_enable(True)
stride = b.out_stride(F)
out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
for c in range(2):
    b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
torch.cuda.synchronize()
prof = b.debug_read("prof", np.uint64, S * 64).reshape(S, 64).astype(np.float64)
bt = b.debug_read("bt", np.uint8, S * 2 * F).reshape(S, 2 * F)
nshort = (bt == 2).sum(axis=1)
tot = prof[:, 31]
print("short granules per stream: mean %.1f of %d (%.1f %%), min %d max %d" % (nshort.mean(), 2 * F, 100 * nshort.mean() / (2 * F), nshort.min(), nshort.max()))
A = np.stack([np.ones(S), nshort], axis=1)
coef, *_ = np.linalg.lstsq(A, tot, rcond=None)
print("stream cycles = %.0f + %.0f per short granule; mean long granule %.0f cycles" % (coef[0], coef[1], coef[0] / (2 * F)))
print("=> a short granule costs %.0f cycles more than a long one; share of the stream's time spent in short granules: %.1f %%" % (coef[1], 100 * (coef[1] + coef[0] / (2 * F)) * nshort.mean() / tot.mean()))
