"""Per-stream time distribution of k_alloc on the bench workload (HX_PROFILE build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmp3_amd import api
import bench
S, F = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
pcm = bench.synth_batch_gpu(torch, S, F, 44100, dev)
b = api.Batch(api.default_control(bitrate=64, short_block_threshold=99999), nstreams=S, max_frames=F)
b.debug_enable(True)
host = pcm.cpu().numpy()
b.encode_host(host)
b.encode_host(host)
prof = b.debug_read("prof", np.uint64, S * 64).reshape(S, 64).astype(np.float64)
t = prof[:, 31] / (2 * F)
print("per-stream ticks/frame: min %.0f mean %.0f p50 %.0f p95 %.0f p99 %.0f max %.0f  max/mean %.3f" % (t.min(), t.mean(), np.median(t), np.percentile(t, 95), np.percentile(t, 99), t.max(), t.max() / t.mean()))
names = {2: "seek_initial", 3: "seek_actual", 6: "big_lucky", 8: "count_bits", 9: "increase_bits", 10: "decrease_bits", 1: "startup", 13: "pack_huff", 20: "#sweeps", 21: "#lucky", 22: "#count_bits"}
order = np.argsort(t)
for label, idx in (("slowest 16", order[-16:]), ("median 16", order[S // 2 - 8:S // 2 + 8]), ("fastest 16", order[:16])):
    print(label, " ".join("%s=%.0f" % (names[k], prof[idx, k].mean() / (2 * F)) for k in names))
full = {0: "load xr", 1: "startup", 2: "seek_initial", 3: "seek_actual", 4: "trade_dual", 5: "scale_factors", 6: "big_lucky", 7: "do_quant", 8: "count_bits", 9: "increase_bits", 10: "decrease_bits", 11: "inverse_sf2", 13: "pack_huff", 14: "frame setup", 15: "compute_mask", 16: "pack_sf", 17: "flush+side", 18: "emit"}
mean = prof.mean(axis=0) / (2 * F)
for s_ in order[-6:][::-1]:
    row = prof[s_] / (2 * F)
    print("stream %4d total %.0f (+%.0f): " % (s_, row[31], row[31] - mean[31]) + " ".join("%s%+.0f" % (full[k], row[k] - mean[k]) for k in full if abs(row[k] - mean[k]) > 300))
hist, edges = np.histogram(t, bins=12)
print("histogram:", " ".join("%.0f:%d" % (edges[i], hist[i]) for i in range(12)))
ms, n = b.alloc_kernel_ms()
print("k_alloc ms %.3f (%d calls)" % (ms, n))
