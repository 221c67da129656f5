"""Phase profile of k_alloc (library must be built with HX_EXTRA=-DHX_PROFILE hmp3_amd/build.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmp3_amd import api, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
import json
kw = json.loads(os.environ["PROF_KW"]) if "PROF_KW" in os.environ else dict(bitrate=64, short_block_threshold=99999)
pcm = synth.batch_pcm(S, F, unique=16, bursts=bool(int(os.environ.get("PROF_BURSTS", "0"))), rho=float(os.environ.get("PROF_RHO", "0.7")))
b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
b.debug_enable(True)
b.encode_host(pcm)
b.encode_host(pcm)
prof = b.debug_read("prof", np.uint64, S * 64).reshape(S, 64).astype(np.float64)
names = {0: "load xr", 1: "startup", 2: "seek_initial", 3: "seek_actual", 4: "trade_dual", 5: "scale_factors", 6: "big_lucky",
         7: "do_quant", 8: "count_bits", 9: "increase_bits", 10: "decrease_bits", 11: "inverse_sf2", 12: "bitallo total",
         13: "hand-over: records + lines", 14: "frame setup", 15: "compute_mask", 16: "next granule fetch issue", 17: "flush+side", 18: "emit", 19: "hand-over: scalefactor sizing", 30: "sweep: join wait", 20: "#seek sweeps", 21: "#lucky passes", 22: "#count_bits", 23: "lucky: setup", 24: "lucky: terms", 25: "lucky: sums", 26: "lucky: replay", 27: "sweep: publish", 28: "sweep: lines", 29: "sweep: sums", 32: "sf: scfsi", 33: "sf: maxima", 34: "sf: compress", 35: "sf: sum", 36: "cnt: ballots", 37: "cnt: j2/j3", 38: "cnt: regions", 39: "cnt: pairs", 40: "cnt: quads", 41: "cnt: reduce", 42: "huff: build pairs", 43: "huff: place pairs", 44: "huff: build quads", 45: "huff: place quads", 31: "kernel total"}
tot = prof[:, 31].mean()
print("mean cycles per frame (clock64 ticks), S=%d F=%d" % (S, F))
for k in sorted(names):
    v = prof[:, k].mean() / F
    print("  %-16s %10.0f  %5.1f%%" % (names[k], v, 100 * prof[:, k].mean() / tot))
print("  kernel ticks/frame %.0f" % (tot / F))
ms, n = b.alloc_kernel_ms()
print("k_alloc ms %.3f (%d calls)" % (ms, n))
t = prof[:, 31] / F
print("per-stream kernel ticks/frame: min %.0f  mean %.0f  p95 %.0f  max %.0f" % (t.min(), t.mean(), np.percentile(t, 95), t.max()))
