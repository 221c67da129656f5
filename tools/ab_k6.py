"""A/B of the stream walk alone (plain calls: no front end or packing beside it): for each library variant, K6 ms per call and
the per-stream durations (100 MHz ticks -> ms) at the bench's config-n signal set.
  python tools/ab_k6.py base,c2 [S] [F] [cfg] [rounds]     (libraries hmp3_amd/libhmp3amd_<name>.so; one child process per run)"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import bench
    from hmp3_amd import api
    S, F, CFG = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda:0")
    w = bench.workload(CFG)
    kw, sr = w["classes"][0]
    rho = [float(x) for x in os.environ["AB_RHO"].split(",")] if os.environ.get("AB_RHO") else w["rho"]
    pcm = bench.synth_batch_gpu(torch, np, S, F, [sr] * S, [rho[i % len(rho)] for i in range(S)], w["bursts"], dev)
    st = torch.cuda.current_stream().cuda_stream
    b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    stride = b.out_stride(F)
    out = torch.empty((S, stride), dtype=torch.uint8, device=dev); nb = torch.zeros((S,), dtype=torch.int32, device=dev)
    for c in range(2):
        b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
    torch.cuda.synchronize(); b.alloc_kernel_ms()
    durs = []
    for c in range(4):
        b.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nb.data_ptr(), st)
        torch.cuda.synchronize()
        durs.append(b.debug_read("dur", np.uint32, S).astype(np.float64) / 1e5)
    ms, n = b.alloc_kernel_ms()
    d = np.mean(durs, axis=0)
    extra = ""
    try:
        extra = " strict %d big %d" % (int(b.debug_read("strict_sums", np.int32, 1)[0]), int(b.debug_read("big_sweeps", np.int32, 1)[0]))
    except Exception:
        pass
    print("K6 %.3f ms | stream ms: min %.3f mean %.3f p99 %.3f max %.3f | status %d bytes %d%s" % (ms, d.min(), d.mean(), np.percentile(d, 99), d.max(), b.status(), int(nb.sum().item()), extra))
    sys.exit(0)
libs = sys.argv[1].split(",")
S = sys.argv[2] if len(sys.argv) > 2 else "1024"; F = sys.argv[3] if len(sys.argv) > 3 else "64"; CFG = sys.argv[4] if len(sys.argv) > 4 else "2"
for r in range(int(sys.argv[5]) if len(sys.argv) > 5 else 2):
    for v in libs:
        env = dict(os.environ, HMP3AMD_LIB=os.path.join(ROOT, "hmp3_amd", "libhmp3amd_%s.so" % v))
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", S, F, CFG], env=env, capture_output=True, text=True)
        print("%-8s S=%s F=%s cfg %s: %s" % (v, S, F, CFG, (o.stdout.strip().splitlines() or [o.stderr[-300:]])[-1]), flush=True)
