"""One encode call of the bench workload for a rocprofv3 --pmc pass: python3 tools/pmc_run.py [S] [F] [unique]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmp3_amd import api, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
U = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pcm = synth.batch_pcm(S, F, unique=U)
b = api.Batch(api.default_control(bitrate=64, short_block_threshold=99999), nstreams=S, max_frames=F)
b.encode_host(pcm)
b.encode_host(pcm)
ms, n = b.alloc_kernel_ms()
print("k_alloc ms %.3f (%d calls) S=%d F=%d unique=%d" % (ms, n, S, F, U))
