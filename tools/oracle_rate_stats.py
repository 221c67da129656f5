"""Rate-loop path statistics of single streams on the CPU oracle (oracle/hxo_alloc.c: hxo_rate_stats): how often a granule
enters increase_bits / decrease_bits and how many quantise-and-count passes it takes - what makes a stream the tail of a
resident-set launch.   python tools/oracle_rate_stats.py <pcm.npz from tools/dump_pcm.py> [passes over the signal]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
d = np.load(sys.argv[1])
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = dict(bitrate=64, short_block_threshold=99999)
lib = O.lib()
stats = (C.c_longlong * 8).in_dll(lib, "hxo_rate_stats")
names = ["granules", "increase entered", "increase iterations", "increase step-backs", "decrease entered", "decrease rounds", "limit rounds", "quantise+count passes"]
for k, sid in enumerate(d["ids"]):
    pcm = d["pcm"][k]
    F = pcm.shape[0] // 1152
    enc = O.OracleEncoder(O.default_control(**kw))
    per_frame = []
    for st in range(passes):
        for f in range(F):
            before = list(stats)
            enc.encode_s16(pcm[f * 1152:(f + 1) * 1152])
            if st == passes - 1:
                per_frame.append([stats[i] - before[i] for i in range(8)])
    a = np.array(per_frame, dtype=np.float64)
    tot = a.sum(axis=0)
    print("stream %4d, last of %d passes over %d frames: per granule: increase entered %.3f (iterations %.2f each, step-back %.2f each), decrease entered %.3f (rounds %.2f each), passes %.2f" %
          (sid, passes, F, tot[1] / tot[0], tot[2] / max(tot[1], 1), tot[3] / max(tot[1], 1), tot[4] / tot[0], tot[5] / max(tot[4], 1), tot[7] / tot[0]))
    # where in the signal: passes per frame in blocks of 32 frames
    blk = a[:, 7].reshape(-1, 32).sum(axis=1) / 64.0
    print("    passes per granule by 32-frame block:", " ".join("%.2f" % x for x in blk))
    it = a[:, 2][a[:, 1] > 0] / a[:, 1][a[:, 1] > 0]
    if len(it):
        print("    increase iterations per entering granule (frame means): histogram 1..10:", np.histogram(it, bins=np.arange(0.5, 11.5))[0].tolist())
