"""The stream walk (K6) alone on the chip - plain calls, nothing beside it - at a list of batch sizes (= streams per CU) for
both MPEG-1 builds: launch time and per-stream durations.  Under `rocprofv3 --pmc ... --kernel-trace` the dispatches of one
(kernel, grid size) are the same workload, so tools/k6_contention.py can read the counters per occupancy level.
  python tools/k6_alone.py [F] [cfg] [fat:256,512,768,1024;slim:256,1024,1536] [calls]
  python tools/k6_alone.py --quarters [S] [cfg]      one 256-frame call against four 64-frame calls over the same signal"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hmp3_amd import api

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def stats(d):
    return "min %.3f mean %.3f p50 %.3f p99 %.3f max %.3f" % (d.min(), d.mean(), np.percentile(d, 50), np.percentile(d, 99), d.max())


def make(cfg, S, F):
    w = bench.workload(cfg)
    kw, sr = w["classes"][0]
    rho = [float(x) for x in os.environ["AB_RHO"].split(",")] if os.environ.get("AB_RHO") else w["rho"]
    pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [rho[i % len(rho)] for i in range(S)], w["bursts"], dev)
    return kw, pcm


def call(b, pcm, F, out, nb, stride):
    b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
    torch.cuda.synchronize()
    ms, n = b.alloc_kernel_ms()
    return ms, b.debug_read("dur", np.uint32, b.n).astype(np.float64) / 1e5


if len(sys.argv) > 1 and sys.argv[1] == "--quarters":
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    CFG = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    kw, pcm = make(CFG, S, 256)
    q = [pcm[:, k * 64 * 1152:(k + 1) * 64 * 1152].contiguous() for k in range(4)]
    for var in (["fat", "slim"] if S <= 1024 else ["slim"]):
        os.environ["HMP3AMD_K6"] = var
        for mode in ("1x256", "4x64", "1x256", "4x64"):
            b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=256)
            stride = b.out_stride(256)
            out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
            # two passes over the signal as warm-up (the encoder state is carried: the third pass is steady state)
            for _ in range(2):
                call(b, pcm, 256, out, nb, stride)
            if mode == "1x256":
                ms, d = call(b, pcm, 256, out, nb, stride)
                print("%-4s S=%d cfg %d one 256-frame launch: K6 %.3f ms | stream ms %s" % (var, S, CFG, ms, stats(d)), flush=True)
            else:
                tot, dsum, parts = 0.0, np.zeros(S), []
                for k in range(4):
                    ms, d = call(b, q[k], 64, out, nb, stride)
                    tot += ms; dsum += d; parts.append(ms)
                print("%-4s S=%d cfg %d four 64-frame launches over the same frames: K6 %s = %.3f ms | per-stream sums %s" % (var, S, CFG, " + ".join("%.3f" % x for x in parts), tot, stats(dsum)), flush=True)
            assert b.status() == 0
            b.close()
    sys.exit(0)

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CFG = int(sys.argv[2]) if len(sys.argv) > 2 else 2
plan = sys.argv[3] if len(sys.argv) > 3 else "fat:256,512,768,1024;slim:256,512,1024,1536"
CALLS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
SMAX = max(int(x) for part in plan.split(";") for x in part.split(":")[1].split(","))
kw, pcm_all = make(CFG, SMAX, F)
print("config %d, F = %d; K6 alone, plain calls (mean of %d after 2 warm-up calls)" % (CFG, F, CALLS))
for part in plan.split(";"):
    var, sizes = part.split(":")
    os.environ["HMP3AMD_K6"] = var
    for S in (int(x) for x in sizes.split(",")):
        b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
        stride = b.out_stride(F)
        out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
        pcm = pcm_all[:S].contiguous()
        for c in range(2):
            call(b, pcm, F, out, nb, stride)
        mss, ds = [], []
        for c in range(CALLS):
            ms, d = call(b, pcm, F, out, nb, stride)
            mss.append(ms); ds.append(d)
        d = np.mean(ds, axis=0)
        print("%-4s S=%5d (%.1f streams per CU, resident %d, variant %d): K6 %.3f ms | stream ms %s | streams/ms %.1f" %
              (var, S, S / 256.0, b.resident_streams(), b.k6_variant(), np.mean(mss), stats(d), S / np.mean(mss)), flush=True)
        assert b.status() == 0
        b.close()
        del out, nb
