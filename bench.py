#!/usr/bin/env python3
"""bench.py - throughput of the batched MP3 encode hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): 1024 independent stereo 44.1 kHz streams x 256 frames per
step, CBR 128 kbps, long blocks only (short_block_threshold = 99999), joint stereo; int16 PCM
resident in HBM when the timed region starts.  A step = one pass of the whole pipeline
(polyphase -> MDCT -> psy -> allocator/quantiser/Huffman -> bitstream + reservoir) over the batch;
stream state is carried from step to step, so K steps encode K x 256 consecutive frames of every
stream.  Streams shard over ranks with no collective (weak scaling: 1024 streams per GPU).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_IN_PER_FRAME = 1152 * 2 * 2          # int16 stereo
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8 TB/s spec


def pmc_traffic(kernel, S, F):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE, separate runs, gfx950 correction applied as MI355X_MICROARCH.md prescribes); None
    when no pass exists for this workload.  Counters cannot be read from inside the timed run."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_counters_bench_1024x256.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d["workload"]["streams"] != S or d["workload"]["frames_per_step"] != F:
            return None
        return int(d["kernels"][kernel]["hbm_bytes_corrected"])
    except (OSError, KeyError, ValueError):
        return None


def synth_batch_gpu(torch, nstreams, nframes, sr, dev, first_stream=0):
    """Same signal family as hmp3_amd/synth.py (tones with AM + high-passed random walk,
    R = 0.7 L + 0.3 R'), synthesised on the GPU; per-stream tone parameters from PCG64 with
    seed 0x484D5033 + stream index, noise from torch's generator."""
    n = nframes * 1152
    f = np.empty((nstreams, 2, 12)); a = np.empty_like(f); fm = np.empty_like(f); ph = np.empty_like(f)
    for i in range(nstreams):
        rng = np.random.Generator(np.random.PCG64(0x484D5033 + first_stream + i))
        for c in range(2):
            f[i, c] = rng.uniform(60.0, 9000.0, 12); a[i, c] = rng.uniform(0.02, 0.15, 12)
            fm[i, c] = rng.uniform(0.1, 2.0, 12); ph[i, c] = rng.uniform(0.0, 2 * np.pi, 12)
    out = torch.empty((nstreams, n, 2), dtype=torch.int16, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0x484D5033 + first_stream)
    t = torch.arange(n, dtype=torch.float64, device=dev) / sr
    chunk = 64
    for s0 in range(0, nstreams, chunk):
        s1 = min(nstreams, s0 + chunk)
        x = torch.zeros((s1 - s0, 2, n), dtype=torch.float64, device=dev)
        tf = torch.tensor(f[s0:s1], device=dev); ta = torch.tensor(a[s0:s1], device=dev)
        tfm = torch.tensor(fm[s0:s1], device=dev); tph = torch.tensor(ph[s0:s1], device=dev)
        for k in range(12):
            x += ta[:, :, k:k + 1] * (0.6 + 0.4 * torch.sin(2 * np.pi * tfm[:, :, k:k + 1] * t)) * \
                torch.sin(2 * np.pi * tf[:, :, k:k + 1] * t + tph[:, :, k:k + 1])
        w = torch.cumsum(torch.randn((s1 - s0, 2, n), dtype=torch.float64, device=dev, generator=g), dim=2)
        w = w - torch.nn.functional.avg_pool1d(w, 65, stride=1, padding=32)
        w = w / (w.abs().amax(dim=2, keepdim=True) + 1e-9)
        x = x + 0.1 * w
        x[:, 1] = 0.7 * x[:, 0] + 0.3 * x[:, 1]
        x = x * (0.95 / x.abs().amax(dim=(1, 2), keepdim=True))
        out[s0:s1] = torch.round(x * 32767.0).to(torch.int16).permute(0, 2, 1)
    return out


def _cpu_worker(args):
    """one process: encode `nframes` frames of a synthetic stream with the CPU checker"""
    kind, nframes, seed = args
    from oracle import oracle as O
    from hmp3_amd import synth
    base = synth.stream_pcm(seed, 256)
    reps = (nframes + 255) // 256
    pcm = np.ascontiguousarray(np.tile(base, (reps, 1))[: nframes * 1152])
    ec = O.default_control(bitrate=64, short_block_threshold=99999)
    if kind == "reference":
        r = O.ref()
        h = r.ref_new()
        r.ref_init_s16(h, C.byref(ec))
        out = (C.c_ubyte * (1 << 20))()
        t0 = time.perf_counter()
        r.ref_encode_stream_s16(h, pcm.ctypes.data, nframes, out, len(out))
        dt = time.perf_counter() - t0
        r.ref_free(h)
    else:
        l = O.lib()
        enc = O.OracleEncoder(ec)
        out = (C.c_ubyte * 16384)()
        t0 = time.perf_counter()
        for f in range(nframes):
            l.hxo_encode_frame_s16(enc.h, pcm[f * 1152:].ctypes.data, out)
        dt = time.perf_counter() - t0
    return nframes, dt


def cpu_baseline():
    """the reference built under oracle/_ref (kind "reference") or the oracle restatement
    (kind "port"), one process per host core, on a bounded sample of the same workload"""
    try:
        from oracle import oracle as O
        kind = "reference" if O.ref() is not None else "port"
        if kind == "port":
            O.lib()
    except Exception as e:      # no checker available on this box
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "none", "sample": "unavailable: %s" % e}
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 32))
    per = 24576 if kind == "reference" else 8192      # ~2-3 s of work per core
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(kind, per, i) for i in range(cores)])
    wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": round(frames / busy, 1), "unit": "frames/s", "cores": cores, "kind": kind,
            "sample": "%d processes x %d frames of 44.1 kHz stereo CBR-128 long-block streams (%.1f s wall incl. spawn)" % (cores, per, wall),
            "per_core": round(frames / busy / cores, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=1024, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=256, help="frames per stream per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="plain hx_batch_encode_s16_device calls instead of submit / wait")
    ap.add_argument("--gate", type=int, default=-1, help="hx_batch_set_gate percent (library default 90)")
    args = ap.parse_args()

    import torch
    from hmp3_amd import api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the encoder has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")     # RCCL; used for the barrier and the max-over-ranks time only

    S, F = args.streams, args.frames
    sr = 44100
    ec = api.default_control(bitrate=64, short_block_threshold=99999)
    batch = api.Batch(ec, nstreams=S, max_frames=F, device=local)
    if args.gate >= 0:
        batch.set_gate(args.gate)
    pcm = synth_batch_gpu(torch, S, F, sr, dev, first_stream=rank * S)
    stride = batch.out_stride(F)
    out = torch.empty((S, stride), dtype=torch.uint8, device=dev)
    nbytes = torch.zeros((S,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        # hx_batch_submit_s16_device: the front-end kernels of step n+1 run in the tail of step n's allocator kernel
        # (its slowest streams), on the SIMDs the finished streams have left; all of every step's work completes
        # inside the timed region (hx_batch_wait + synchronize in barrier()).  --no-pipeline: plain calls.
        if not args.no_pipeline:
            batch.submit_device(pcm.data_ptr(), F, out.data_ptr(), stride, nbytes.data_ptr(), stream)
        else:
            batch.encode_device(pcm.data_ptr(), F, out.data_ptr(), stride, nbytes.data_ptr(), stream)

    def barrier():
        batch.wait(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    batch.alloc_kernel_ms()             # drop warm-up timings
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    status = batch.status()
    k_ms, k_calls = batch.alloc_kernel_ms()
    out_total = int(nbytes.sum().item())

    if rank == 0:
        frames = S * F * args.steps * world
        fps = frames / dt
        out_per_frame = out_total / float(S * F)
        alg_bytes = (BYTES_IN_PER_FRAME + out_per_frame) * S * F        # per launch of the dominant kernel
        ach = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
        traffic = pmc_traffic("k_alloc", S, F)
        res = {
            "metric": "batched stereo 44.1kHz frames/sec (whole node), CBR-128",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch=%d stereo 44.1kHz streams x %d frames per step per GPU, CBR-128, long blocks only, int16 PCM resident in HBM" % (S, F),
                       "streams_per_gpu": S, "frames_per_step": F, "parallelism": "streams sharded over %d GPU(s), no collective" % world},
            "x_realtime_per_gpu": round(fps / world * 1152.0 / sr, 1),
            "kernel_status": status,
            "bitstream_bytes_per_frame": round(out_per_frame, 2),
            "roofline": {"bound": "hbm", "kernel": "k_alloc", "achieved": round(ach, 3) if ach else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6) if ach else None, "traffic": traffic,
                         "kernel_ms": round(k_ms, 3), "launches": k_calls,
                         "algorithmic_bytes_per_frame": round(BYTES_IN_PER_FRAME + out_per_frame, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    batch.close()


if __name__ == "__main__":
    main()
