"""Phase profile of k_alloc under the submit path's overlap (front end of the next call and packing of the previous one
in its tail) next to plain calls.  Library built with HX_EXTRA=-DHX_PROFILE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hmp3_amd import api, synth
S, F, calls = 1024, 64, 6
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
print("%-20s %12s %12s   (mean cycles per frame; slowest 5%% of the streams in brackets)" % ("", "plain", "submit"))
slow = {m: np.argsort(res[m][:, 31])[-S // 20:] for m in res}
for k in sorted(names):
    print("%-20s %8.0f (%6.0f) %8.0f (%6.0f)" % (names[k], res["plain"][:, k].mean() / F, res["plain"][slow["plain"], k].mean() / F,
                                                  res["submit"][:, k].mean() / F, res["submit"][slow["submit"], k].mean() / F))
