"""K6 time against streams in flight per CU, for both builds of the stream walk: plain calls (no overlap with the front end), the
bench's config-2 signal set, S streams x F frames.  HMP3AMD_K6 is set per batch.  python tools/r4_occ.py [F] [cfg]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hmp3_amd import api
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CFG = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
w = bench.workload(CFG)
kw, sr = w["classes"][0]
SMAX = 2048
rho = w["rho"]
pcm_all = bench.synth_batch_gpu(torch, np, SMAX, F, [sr] * SMAX, [rho[i % len(rho)] for i in range(SMAX)], w["bursts"], dev)
st = torch.cuda.current_stream().cuda_stream
print("config %d, F = %d; K6 ms per call (mean of 3 after 2 warm-up calls)" % (CFG, F))
for var in ("fat", "slim"):
    os.environ["HMP3AMD_K6"] = var
    for S in (256, 512, 768, 1024, 1280, 1536, 2048):
        b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
        stride = b.out_stride(F)
        out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
        pcm = pcm_all[:S].contiguous()
        for c in range(2):
            b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
        torch.cuda.synchronize()
        b.alloc_kernel_ms()
        for c in range(3):
            b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
        torch.cuda.synchronize()
        ms, n = b.alloc_kernel_ms()
        print("%-5s S=%5d resident=%5d variant=%d  K6 %.3f ms  -> %.1f k frames/s per launch-ms, streams/ms %.1f" % (var, S, b.resident_streams(), b.k6_variant(), ms, S * F / ms, S / ms), flush=True)
        assert b.status() == 0
        b.close()
        del out, nb
