"""Phase profile of k_alloc under the submit path's overlap (front end of the next call and packing of the previous one
in its tail) next to plain calls.  Library built with HX_EXTRA=-DHX_PROFILE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hmp3_amd import api, synth
S, F, calls = 1024, int(os.environ.get("PF", "64")), 6
pcm = synth.batch_pcm(S, F, unique=16)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
d_pcm = torch.from_numpy(pcm).to(dev)
names = {0: "join wait (fetch)", 1: "startup", 3: "seek_actual", 6: "big_lucky", 7: "do_quant", 8: "count_bits", 11: "inverse_sf2", 12: "bitallo total",
         13: "hand-over", 17: "placement", 31: "kernel total"}
res = {}
for mode in ("plain", "submit"):
    b = api.Batch(api.default_control(bitrate=64, short_block_threshold=99999), nstreams=S, max_frames=F)
    b.debug_enable(True)
    stride = b.out_stride(F)
    outs = [torch.zeros((S, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
    nbs = [torch.zeros((S,), dtype=torch.int32, device=dev) for _ in range(2)]
    for c in range(calls):
        f = b.encode_device if mode == "plain" else b.submit_device
        f(d_pcm.data_ptr(), F, outs[c & 1].data_ptr(), stride, nbs[c & 1].data_ptr(), st)
    b.wait(st)
    torch.cuda.synchronize()
    prof = b.debug_read("prof", np.uint64, S * 64).reshape(S, 64).astype(np.float64)
    res[mode] = prof
    b.close()
for m in res:       # wall-clock (100 MHz) start and end of every stream's workgroup in the last launch
    t0, t1 = res[m][:, 40], res[m][:, 41]
    z = t0.min()
    print("%-7s starts: %.3f .. %.3f ms after the first; ends: earliest %.3f, mean %.3f, last %.3f ms; mean duration %.3f ms" % (
        m, 0.0, (t0.max() - z) / 1e5, (t1.min() - z) / 1e5, (t1.mean() - z) / 1e5, (t1.max() - z) / 1e5, (t1 - t0).mean() / 1e5))
print("%-20s %12s %12s   (mean cycles per frame; slowest 5%% of the streams in brackets)" % ("", "plain", "submit"))
slow = {m: np.argsort(res[m][:, 31])[-S // 20:] for m in res}
for k in sorted(names):
    print("%-20s %8.0f (%6.0f) %8.0f (%6.0f)" % (names[k], res["plain"][:, k].mean() / F, res["plain"][slow["plain"], k].mean() / F,
                                                  res["submit"][:, k].mean() / F, res["submit"][slow["submit"], k].mean() / F))
