#!/usr/bin/env python3
"""bench.py - throughput of the batched MP3 encode hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

--gpus N > 1 without a torch.distributed environment starts N single-GPU worker processes itself
(one rank per GPU, before anything in this process touches the GPU); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks it is given
are used as they are.  A world size that differs from --gpus is an error.

Workloads (BASELINE.json configs; per GPU, weak scaling):
  2 (default)  1024 streams x 256 frames, 44.1 kHz stereo, CBR-128, long blocks only (SBT 99999)
  3            4096 x 256, 44.1 kHz, VBR -V50, block switching, bursts in the signal
  4            4096 x 128, 48 kHz, VBR -V100 -HF2 -F19000 (one GPU's share of 32768 streams)
  5            4096 x 256, sample rate {32, 44.1, 48 kHz}[stream mod 3], CBR-128, inter-channel
               correlation {0, 0.3, 0.7, 1}[stream mod 4] (one GPU's share of 8 x 4096)
int16 PCM resident in HBM when the timed region starts.  A step = one pass of the whole pipeline
(polyphase -> MDCT -> psy -> allocator/quantiser/Huffman -> bitstream + reservoir) over the batch;
stream state is carried from step to step, so K steps encode K x F consecutive frames of every
stream.  Streams shard over ranks with no collective; RCCL carries the barrier and the
max-over-ranks time only.  After the timed region rank 0 copies --verify streams (PCM and the last
step's bitstream) to the host and compares them byte for byte with the CPU oracle run over the
same W + K steps.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8 TB/s spec
# vector issue peak: 256 CUs x 4 SIMD-32, one wave64 vector instruction per 2 cycles and SIMD, 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0
RHO_CYCLE = [0.7, 0.0, 1.0, 0.3]


def workload(cfg):
    """-> dict(name, S, F, classes = [(control kwargs, sample rate)], rho list, bursts)"""
    if cfg == 2:
        return dict(name="batch=%d stereo 44.1kHz streams x %d frames per step per GPU, CBR-128, long blocks only, int16 PCM resident in HBM",
                    S=1024, F=256, classes=[(dict(bitrate=64, short_block_threshold=99999), 44100)], rho=[0.7], bursts=False)
    if cfg == 3:
        return dict(name="batch=%d stereo 44.1kHz streams x %d frames per step per GPU, VBR -V50, psy model + block switching (bursts), int16 PCM resident in HBM",
                    S=4096, F=256, classes=[(dict(), 44100)], rho=[0.7], bursts=True)
    if cfg == 4:
        return dict(name="batch=%d stereo 48kHz streams x %d frames per step per GPU (one GPU's share of 32768), VBR -V100 -HF2 -F19000, int16 PCM resident in HBM",
                    S=4096, F=128, classes=[(dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), 48000)], rho=[0.7], bursts=True)
    if cfg == 5:
        return dict(name="batch=%d stereo streams x %d frames per step per GPU (one GPU's share of 8 x 4096), 32/44.1/48kHz by stream mod 3, CBR-128, "
                         "inter-channel correlation {0.7,0,1,0.3} by stream mod 4, block switching, int16 PCM resident in HBM",
                    S=4096, F=256, classes=[(dict(bitrate=64, samprate=32000), 32000), (dict(bitrate=64), 44100), (dict(bitrate=64, samprate=48000), 48000)],
                    rho=RHO_CYCLE, bursts=True)
    raise SystemExit("unknown --config %r" % cfg)


def pmc_profile(cfg, S, F):
    """the committed rocprofv3 --pmc passes for this workload (profiles/), or None; the file carries the build id
    (hx_build_id) of the library the counters were collected with"""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters_*.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
            w = d["workload"]
            if w["streams"] == S and w["frames_per_step"] == F and int(w.get("config", 2)) == cfg:
                best = (d, os.path.relpath(path, ROOT))     # the latest round's file wins
        except (OSError, KeyError, ValueError):
            pass
    return best


def synth_batch_gpu(torch, np, nstreams, nframes, srs, rhos, bursts, dev, first_stream=0):
    """Same signal family as hmp3_amd/synth.py (tones with AM + high-passed random walk, R = rho L +
    (1 - rho) R', optional decaying white bursts every 0.7 s), synthesised on the GPU; per-stream tone
    parameters from PCG64 with seed 0x484D5033 + global stream index, noise from torch's generator.
    srs / rhos: per-stream sample rate and inter-channel correlation."""
    n = nframes * 1152
    out = torch.empty((nstreams, n, 2), dtype=torch.int16, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0x484D5033 + first_stream)
    chunk = 64
    idx_all = np.arange(nstreams)
    for sr in sorted(set(srs)):
        sel = idx_all[np.asarray(srs) == sr]
        t = torch.arange(n, dtype=torch.float64, device=dev) / sr
        for c0 in range(0, len(sel), chunk):
            ids = sel[c0:c0 + chunk]
            m = len(ids)
            f = np.empty((m, 2, 12)); a = np.empty_like(f); fm = np.empty_like(f); ph = np.empty_like(f)
            for k, i in enumerate(ids):
                rng = np.random.Generator(np.random.PCG64(0x484D5033 + first_stream + int(i)))
                for c in range(2):
                    f[k, c] = rng.uniform(60.0, 9000.0, 12); a[k, c] = rng.uniform(0.02, 0.15, 12)
                    fm[k, c] = rng.uniform(0.1, 2.0, 12); ph[k, c] = rng.uniform(0.0, 2 * np.pi, 12)
            x = torch.zeros((m, 2, n), dtype=torch.float64, device=dev)
            tf = torch.tensor(f, device=dev); ta = torch.tensor(a, device=dev)
            tfm = torch.tensor(fm, device=dev); tph = torch.tensor(ph, device=dev)
            for k in range(12):
                x += ta[:, :, k:k + 1] * (0.6 + 0.4 * torch.sin(2 * np.pi * tfm[:, :, k:k + 1] * t)) * \
                    torch.sin(2 * np.pi * tf[:, :, k:k + 1] * t + tph[:, :, k:k + 1])
            w = torch.cumsum(torch.randn((m, 2, n), dtype=torch.float64, device=dev, generator=g), dim=2)
            w = w - torch.nn.functional.avg_pool1d(w, 65, stride=1, padding=32)
            w = w / (w.abs().amax(dim=2, keepdim=True) + 1e-9)
            x = x + 0.1 * w
            if bursts:
                period = int(0.7 * sr)
                env = torch.exp(-torch.arange(2000, dtype=torch.float64, device=dev) / 200.0)
                for s0 in range(period // 2, n - 2000, period):
                    x[:, :, s0:s0 + 2000] += 0.5 * env * torch.randn((m, 2, 2000), dtype=torch.float64, device=dev, generator=g)
            rho = torch.tensor([rhos[int(i)] for i in ids], dtype=torch.float64, device=dev).view(m, 1)
            x[:, 1] = rho * x[:, 0] + (1.0 - rho) * x[:, 1]
            x = x * (0.95 / x.abs().amax(dim=(1, 2), keepdim=True))
            out[torch.as_tensor(ids, device=dev)] = torch.round(x * 32767.0).to(torch.int16).permute(0, 2, 1)
    return out


# ---- CPU side: the checker (oracle/) as verifier and as the reported CPU baseline -------------

def _verify_worker(args):
    """one process: the oracle's bitstream for the last `nlast` calls of `steps` passes over pcm"""
    kw, pcm, steps, nlast = args
    from oracle import oracle as O
    import numpy as np
    enc = O.OracleEncoder(O.default_control(**kw))
    F = pcm.shape[0] // 1152
    tail = []
    for st in range(steps):
        for f in range(F):
            b = enc.encode_s16(pcm[f * 1152:(f + 1) * 1152])
            if st * F + f >= steps * F - nlast:
                tail.append(b)
    return b"".join(tail)


def verify_streams(torch, np, wl, kws, pcm, out, nbytes, ids, steps_total):
    """byte-compare the last step's output of streams `ids` with the oracle; returns (n_ok, first bad id)"""
    import multiprocessing as mp
    F = pcm.shape[1] // 1152
    host_nb = nbytes.cpu().numpy()
    ok, bad = 0, None
    # (--verify all: whole batches go through here, in blocks of 256 streams so that the host copies stay a few hundred MB)
    with mp.get_context("spawn").Pool(max(1, min(len(ids), usable_cpus(), 32))) as pool:
        for c0 in range(0, len(ids), 256):
            blk = ids[c0:c0 + 256]
            host_pcm = pcm[torch.as_tensor(blk, device=pcm.device)].cpu().numpy()
            host_out = out[torch.as_tensor(blk, device=out.device)].cpu().numpy()
            want = pool.map(_verify_worker, [(kws[i], host_pcm[k], steps_total, F) for k, i in enumerate(blk)], chunksize=1)
            for k, i in enumerate(blk):
                got = host_out[k, :host_nb[i]].tobytes()
                if got == want[k]:
                    ok += 1
                elif bad is None:
                    n = min(len(got), len(want[k]))
                    diff = next((j for j in range(n) if got[j] != want[k][j]), n)
                    bad = {"stream": int(i), "bytes": len(got), "oracle_bytes": len(want[k]), "first_difference_at": diff}
    return ok, bad


def _cpu_worker(args):
    """one process: encode `nframes` frames of a synthetic stream with the CPU checker"""
    kind, nframes, seed, kw, sr, bursts = args
    import numpy as np
    from oracle import oracle as O
    from hmp3_amd import synth
    base = synth.stream_pcm(seed, 256, sr=sr, bursts=bursts)
    reps = (nframes + 255) // 256
    pcm = np.ascontiguousarray(np.tile(base, (reps, 1))[: nframes * 1152])
    ec = O.default_control(**kw)
    if kind == "reference":
        r = O.ref()
        h = r.ref_new()
        r.ref_init_s16(h, C.byref(ec))
        out = (C.c_ubyte * (4 << 20))()
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


def usable_cpus():
    """CPUs this process may run on: the affinity mask, cut down to a cgroup CPU quota if one is set"""
    try:
        n = len(os.sched_getaffinity(0)) or 1
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]       # cgroup v2
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = max(1, min(n, (quota + period // 2) // period))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(wl):
    """the reference built under oracle/_ref (kind "reference") or the oracle restatement (kind "port"), one process
    and one stream per core, on a bounded sample of the same workload.  Timed with every usable host CPU and, where the
    box has more than 32, with 32 as well (hardware threads of a shared host do not all deliver a core's worth: on the
    round-2 box 32 processes encoded faster than 256): the faster run is the value, both are listed."""
    try:
        from oracle import oracle as O
        kind = "reference" if O.ref() is not None else "port"
        if kind == "port":
            O.lib()
    except Exception as e:      # no checker available on this box
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "none", "sample": "unavailable: %s" % e}
    import multiprocessing as mp
    host = os.cpu_count() or 1
    usable = usable_cpus()
    ncls = len(wl["classes"])
    runs = []
    for cores in sorted({usable, min(usable, 32)}, reverse=True):
        per = (24576 if kind == "reference" else 8192) // (4 if cores > 32 else 1)     # ~2-3 s of work per process
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(kind, per, i, wl["classes"][i % ncls][0], wl["classes"][i % ncls][1], wl["bursts"]) for i in range(cores)])
        wall = time.perf_counter() - t0
        frames = sum(r[0] for r in res)
        busy = max(r[1] for r in res)
        runs.append({"cores": cores, "value": round(frames / busy, 1), "per_core": round(frames / busy / cores, 1), "frames_per_process": per, "wall_s": round(wall, 1)})
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "frames/s", "cores": best["cores"], "host_cpu_count": host, "usable_cpus": usable, "kind": kind,
            "sample": "%d processes x %d frames of this workload's stream classes, one stream each; rate = frames / the slowest process's encode time" % (best["cores"], best["frames_per_process"]),
            "per_core": best["per_core"], "runs": runs}


# ---- launcher: --gpus N without a torch.distributed environment --------------------------------

def launch_ranks(n, argv):
    """start n single-GPU worker processes of this script (rank i on GPU i) and wait for them.
    Nothing in this process has touched the GPU (torch is not even imported)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                r = p.poll()
                if r is None:
                    continue
                procs.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in procs:         # a rank failed: the others would wait for it in the barrier
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; the first one's front end and the last one's packing have nothing to hide behind, so few steps understate the steady rate")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[n-1]")
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the config's)")
    ap.add_argument("--frames", type=int, default=0, help="frames per stream per step (default: the config's)")
    ap.add_argument("--verify", default="16", help="streams checked against the CPU oracle after the timed region: a number (0 = off) or 'all' (every stream of every rank's block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-worst-case", action="store_true", help="skip the second timed pass on the correlation-cycled signal set (config 2, one GPU)")
    ap.add_argument("--host-fed", type=int, default=-1, help="1 / 0: also time (or not) the workload fed from page-locked host memory through the pipelined host-buffer calls; default: config 2 only")
    ap.add_argument("--other-configs", type=int, default=-1, help="1 / 0: also run the other BASELINE configs at full width for 4 timed steps each and append them as other_configs (one GPU: 3, 4, 5; several: 4 and 5, the configs BASELINE defines on 8 GPUs), each with its host-fed rate; default: with config 2 at its own size")
    ap.add_argument("--other-size", default="", help="S,F: run the other_configs (and their host-fed passes) at this reduced size per GPU instead of their full width, also next to --streams / --frames (the multi-GPU pre-flight test: everything of an 8-rank run except the devices)")
    ap.add_argument("--strong-scaling", type=int, default=-1, help="1 / 0: with --gpus N > 1 also time config 2's 1024 streams split N ways (strong scaling, SURVEY 8e); default: with config 2 at its own size")
    ap.add_argument("--no-pipeline", action="store_true", help="plain hx_batch_encode_s16_device calls instead of submit / wait")
    ap.add_argument("--gate", type=int, default=-1, help="hx_batch_set_gate percent (library default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process group of the barrier / max-time (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank on GPU 0 (tests on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous, sharding and reporting only, no encode (CPU test of the multi-rank path)")
    args = ap.parse_args()
    try:
        args.verify = (1 << 30) if str(args.verify).lower() == "all" else int(args.verify)      # (capped at the batch's streams where it is used)
    except ValueError:
        raise SystemExit("--verify takes a number or 'all'")
    if args.other_configs < 0:
        args.other_configs = 1 if (args.config == 2 or args.other_size) else 0
    if args.strong_scaling < 0:
        args.strong_scaling = 1 if (args.config == 2 and args.gpus > 1 and not args.streams and not args.frames) else 0
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # what the line will carry beside `value` (also printed by --dry-run, so that the multi-rank CPU tests can check the plan)
    other_size = tuple(int(x) for x in args.other_size.split(",")) if args.other_size else ()
    if other_size and (len(other_size) != 2 or min(other_size) < 1):
        raise SystemExit("--other-size takes S,F")
    plan_other = [c for c in ((3, 4, 5) if args.gpus == 1 else (4, 5)) if c != args.config] if (args.other_configs and ((not args.streams and not args.frames) or other_size)) else []
    plan = {"other_configs": plan_other, "host_fed_configs": ([args.config] if (args.host_fed == 1 or (args.host_fed < 0 and args.config == 2)) else []) + (plan_other if args.host_fed != 0 else []),
            "strong_scaling": bool(args.strong_scaling and args.gpus > 1), "worst_case": bool(args.config == 2 and args.gpus == 1 and not args.no_worst_case)}
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))

    import numpy as np
    import torch
    from hmp3_amd import shard
    def config_setup(cfg, S=0, F=0):
        """one rank's share of a BASELINE configuration: weak scaling, S streams per GPU, a contiguous block per rank; the
        stream classes and correlations cycle over the global stream index"""
        wl = workload(cfg)
        S = S or wl["S"]
        F = F or wl["F"]
        first, last = shard.shard_range(S * world, world, rank)
        assert last - first == S
        ncls = len(wl["classes"])
        return dict(cfg=cfg, wl=wl, S=S, F=F, first=first, ncls=ncls,
                    kws=[wl["classes"][(first + i) % ncls][0] for i in range(S)],
                    srs=[wl["classes"][(first + i) % ncls][1] for i in range(S)],
                    rhos=[wl["rho"][(first + i) % len(wl["rho"])] for i in range(S)])

    main_c = config_setup(args.config, args.streams, args.frames)
    wl, S, F, first, ncls, kws, srs, rhos = (main_c[k] for k in ("wl", "S", "F", "first", "ncls", "kws", "srs", "rhos"))
    last = first + S
    wl0, S0, F0, kws0, ncls0 = wl, S, F, kws, ncls

    dist = None
    if world > 1:
        import torch.distributed as dist
    if args.dry_run:
        if dist is not None:
            dist.init_process_group(args.backend)
            t = torch.tensor([1.0 + rank], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seen = torch.zeros(world, dtype=torch.int64); seen[rank] = last - first
            dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        else:
            t = torch.tensor([1.0]); seen = torch.tensor([S])
        # every rank's block as it would encode it: global stream range, and the stream classes / correlations at both ends
        # (they cycle over the GLOBAL stream index, so they continue across the rank boundaries)
        cls_of = [wl["classes"].index((kws[i], srs[i])) for i in list(range(min(6, S))) + list(range(max(S - 6, 0), S))]
        mine = {"rank": rank, "first": first, "count": S, "cls_head": cls_of[:min(6, S)], "cls_tail": cls_of[-min(6, S):],
                "rho_head": rhos[:6], "rho_tail": rhos[-6:], "cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
        blocks = [mine]
        if dist is not None:
            blocks = [None] * world
            dist.all_gather_object(blocks, mine)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int((seen > 0).sum()), "streams_total": int(seen.sum()),
                              "max_time_token": float(t.item()), "blocks": blocks, "nclasses": ncls, "rho_cycle": wl["rho"], "plan": plan,
                              "strong_scaling_blocks": [list(shard.shard_range(workload(2)["S"], world, r)) for r in range(world)] if plan["strong_scaling"] else None,
                              "config": {"workload": wl["name"] % (S, F), "baseline_config": args.config}}), flush=True)
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    from hmp3_amd import api
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the encoder has no CPU path")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # Host placement: this rank's thread - and with it the page-locked buffers it allocates (first touch) and the copies it
    # submits - on the NUMA node its GPU hangs on.  At 8 GPUs x 49 GB/s of PCM the host-fed path is a host-memory-bandwidth
    # problem; it is decided by which socket the buffers are on.  Best effort (no sysfs entry / one node: nothing changes).
    cpus_before = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    try:
        placement = {"rank": rank, "device": local, "numa_node": api.device_numa_node(local), "cpus_bound": api.bind_thread_to_device(local)}
    except AttributeError:      # an older build of the library (HMP3AMD_LIB)
        placement = {"rank": rank, "device": local, "numa_node": None, "cpus_bound": 0}
    placement["cpus_allowed"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    if dist is not None:
        dist.init_process_group(args.backend)     # RCCL; used for the barrier and the max-over-ranks time only

    def run(pcm, verify_n, c=None, steps=None, warmup=None):
        """warm-up + timed steps over one signal set with a fresh batch; returns measurements
        (c: another configuration's set-up from config_setup(), default = the line's own)"""
        S, F, wl, kws, ncls = (c["S"], c["F"], c["wl"], c["kws"], c["ncls"]) if c else (S0, F0, wl0, kws0, ncls0)
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        ctl = api.default_control(**kws[0]) if ncls == 1 else [api.default_control(**k) for k in kws]
        batch = api.Batch(ctl, nstreams=S, max_frames=F, device=local)
        if args.gate >= 0:
            batch.set_gate(args.gate)
        stride = batch.out_stride(F)
        # two sets of output buffers, taken in turn like a client that keeps every call's output would: the packing of
        # step n overlaps the allocator launch of step n + 1 (with one set the library runs them one after the other)
        outs = [torch.empty((S, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
        nbs = [torch.zeros((S,), dtype=torch.int32, device=dev) for _ in range(2)]
        turn = [0]
        stream = torch.cuda.current_stream().cuda_stream

        def step():
            # hx_batch_submit_s16_device: the front-end kernels of step n+1 run in the tail of step n's allocator kernel
            # (its slowest streams); all of every step's work completes inside the timed region (hx_batch_wait +
            # synchronize in barrier()).  --no-pipeline: plain calls.
            out, nbytes = outs[turn[0] & 1], nbs[turn[0] & 1]
            turn[0] += 1
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

        for _ in range(warmup):
            step()
        barrier()
        batch.alloc_kernel_ms()             # drop warm-up timings
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        out, nbytes = outs[(turn[0] - 1) & 1], nbs[(turn[0] - 1) & 1]       # the last step's
        m = {"dt": dt, "status": batch.status(), "out_total": int(nbytes.sum().item())}
        m["k_ms"], m["k_calls"] = batch.alloc_kernel_ms()
        try:
            # the tail the step is made of: every stream's own time inside the last timed step's allocator launch
            # (device wall clock, 100 MHz ticks -> ms); a launch lasts as long as its slowest stream
            d = batch.debug_read("dur", np.uint32, S).astype(np.float64) / 1e5
            m["stream_ms"] = {"min": round(float(d.min()), 3), "mean": round(float(d.mean()), 3), "p50": round(float(np.percentile(d, 50)), 3),
                              "p99": round(float(np.percentile(d, 99)), 3), "max": round(float(d.max()), 3), "streams": int(S), "frames": int(F),
                              # the three slowest streams (index in this rank's block, ms): at a resident-set batch the launch IS the first of them
                              "slowest": [[int(i), round(float(d[i]), 3)] for i in np.argsort(-d)[:3]]}
        except Exception:           # an older build of the library (HMP3AMD_LIB)
            m["stream_ms"] = None
        try:
            m["k6"] = {"kernel": ("k_alloc_slim" if batch.k6_variant() == 1 else None), "resident_streams": batch.resident_streams()}
        except AttributeError:      # an older build of the library (HMP3AMD_LIB)
            m["k6"] = None
        try:
            m["gate_timeouts"] = batch.gate_timeouts()
        except AttributeError:      # an older build of the library (HMP3AMD_LIB)
            m["gate_timeouts"] = None
        if verify_n > 0:            # every rank checks streams of its own block
            rs = np.random.RandomState(12345 + (c["cfg"] if c else args.config) + 1000 * rank)
            ids = sorted(rs.choice(S, size=min(verify_n, S), replace=False).tolist())
            ok, bad = verify_streams(torch, np, wl, kws, pcm, out, nbytes, ids, warmup + steps)
            m["verified"], m["verify_bad"], m["verify_n"] = ok, bad, len(ids)
        batch.close()
        del out, nbytes, outs, nbs
        return m

    def run_host_fed(pcm, c=None, steps_h=None):
        """The same workload fed from host memory: page-locked PCM in, bitstream out, through the pipelined host-buffer
        entry points (hx_batch_submit_s16_host / hx_batch_wait_host: the PCM of call n+1 and the bitstream of call n-1
        cross PCIe while call n is encoded).  Reported next to `value`, never as it.  (c: another configuration's set-up)"""
        S, F, kws, ncls = (c["S"], c["F"], c["kws"], c["ncls"]) if c else (S0, F0, kws0, ncls0)
        steps_h = steps_h or max(2, min(args.steps, 8))
        ctl = api.default_control(**kws[0]) if ncls == 1 else [api.default_control(**k) for k in kws]
        batch = api.Batch(ctl, nstreams=S, max_frames=F, device=local)
        stride = batch.out_stride(F)
        hp = pcm.cpu().pin_memory()
        outs = [torch.empty((S, stride), dtype=torch.uint8).pin_memory() for _ in range(2)]
        nbs = [torch.zeros((S,), dtype=torch.int32).pin_memory() for _ in range(2)]

        def go(n):
            for i in range(n):
                batch.submit_host(hp.data_ptr(), F, outs[i & 1].data_ptr(), stride, nbs[i & 1].data_ptr())
            batch.wait_host()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        go(2)
        t0 = time.perf_counter()
        go(steps_h)
        dt = time.perf_counter() - t0
        if c is None:
            placement["host_fed_ms_per_step"] = round(dt / steps_h * 1e3, 3)       # this rank's own time (the line's value uses the slowest rank's)
        else:
            placement.setdefault("host_fed_ms_per_step_other", {})[str(c["cfg"])] = round(dt / steps_h * 1e3, 3)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        st = batch.status()
        used = int(nbs[(steps_h - 1) & 1].sum().item())
        batch.close()
        del outs, nbs
        return {"value": round(S * F * steps_h * world / dt, 1), "unit": "frames/s", "steps": steps_h, "ms_per_step": round(dt / steps_h * 1e3, 3),
                "h2d_GBps": round(hp.numel() * 2 * steps_h / dt / 1e9, 2), "d2h_GBps": round(S * stride * steps_h / dt / 1e9, 2),
                "bytes_per_step": {"pcm_in": int(hp.numel() * 2), "out_buffer": int(S * stride), "bitstream_used": used}, "kernel_status": st,
                "what": "per GPU: int16 PCM in page-locked host memory -> hx_batch_submit_s16_host -> bitstream in page-locked host memory "
                        "(whole output buffer copied back), PCIe both ways inside the timed region"}

    pcm = synth_batch_gpu(torch, np, S, F, srs, rhos, wl["bursts"], dev, first_stream=first)
    vper = args.verify if world == 1 else (max(2, -(-args.verify // world)) if args.verify > 0 else 0)
    m = run(pcm, vper)
    host_fed = None
    if args.host_fed == 1 or (args.host_fed < 0 and args.config == 2):
        host_fed = run_host_fed(pcm)
    worst = None
    if args.config == 2 and world == 1 and not args.no_worst_case:
        # the same config on the least friendly signal mix of the family: correlation cycled over {0.7, 0, 1, 0.3}
        del pcm
        pcm = synth_batch_gpu(torch, np, S, F, srs, [RHO_CYCLE[i % 4] for i in range(S)], wl["bursts"], dev, first_stream=first)
        worst = run(pcm, 0)
    del pcm
    # ---- the other BASELINE configurations at full width, a few steps each: reported beside `value`, never as it ----
    others = []
    for cfg in plan["other_configs"]:
        c = config_setup(cfg, *other_size)
        pcm = synth_batch_gpu(torch, np, c["S"], c["F"], c["srs"], c["rhos"], c["wl"]["bursts"], dev, first_stream=c["first"])
        mo = run(pcm, (args.verify if args.verify >= (1 << 30) else min(args.verify, 8)) if world == 1 else (2 if args.verify > 0 else 0), c=c, steps=4, warmup=2)
        # the same fed from page-locked host memory: the deployable rate of the configuration (PCIe both ways in the timed region)
        mo["host_fed"] = run_host_fed(pcm, c=c, steps_h=4) if cfg in plan["host_fed_configs"] else None
        del pcm
        torch.cuda.empty_cache()
        if dist is not None:        # streams verified: every rank's, summed
            tv = torch.tensor([mo.get("verified", 0), mo.get("verify_n", 0), mo["status"] & 0x7FFFFFFF if mo["status"] >= 0 else 0x40000000],
                              dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
            lst = [torch.zeros_like(tv) for _ in range(world)]
            dist.all_gather(lst, tv)
            mo["verified_all"], mo["verify_n_all"] = int(sum(int(x[0]) for x in lst)), int(sum(int(x[1]) for x in lst))
            mo["status_all"] = 0
            for x in lst:
                mo["status_all"] |= int(x[2])
        others.append((c, mo))
    # ---- strong scaling ("for transparency"): config 2's 1024 streams split over the ranks ----
    strong = None
    if plan["strong_scaling"]:
        wl2 = workload(2)
        f2, l2 = shard.shard_range(wl2["S"], world, rank)
        cs = dict(cfg=2, wl=wl2, S=l2 - f2, F=(other_size[1] if other_size else wl2["F"]), first=f2, ncls=1, kws=[wl2["classes"][0][0]] * (l2 - f2),
                  srs=[wl2["classes"][0][1]] * (l2 - f2), rhos=[wl2["rho"][0]] * (l2 - f2))
        pcm = synth_batch_gpu(torch, np, cs["S"], cs["F"], cs["srs"], cs["rhos"], False, dev, first_stream=f2)
        ms_ = run(pcm, 2 if args.verify > 0 else 0, c=cs, steps=8, warmup=2)
        del pcm
        torch.cuda.empty_cache()
        strong = (cs, ms_)

    # ---- every rank's health in the line: status word (OR), gate time-outs (sum), streams verified (sum) ----
    # status bit 0..2 are failures (hmp3_amd.h); a gate time-out costs overlap only and is counted separately
    mine_bad = (m["status"] != 0) or (m.get("verified", 0) != m.get("verify_n", 0)) or (host_fed is not None and host_fed["kernel_status"] != 0) \
        or (worst is not None and worst["status"] != 0) \
        or any(mo["status"] != 0 or mo.get("verified", 0) != mo.get("verify_n", 0) or (mo.get("host_fed") is not None and mo["host_fed"]["kernel_status"] != 0) for _, mo in others) \
        or (strong is not None and (strong[1]["status"] != 0 or strong[1].get("verified", 0) != strong[1].get("verify_n", 0)))
    if os.environ.get("HMP3AMD_BENCH_FAULT_RANK") == str(rank):      # test hook: this rank reports a failure (tests/test_gpu_runtime.py)
        mine_bad = True
    vals = [m["status"] & 0x7FFFFFFF if m["status"] >= 0 else 0x40000000, m["gate_timeouts"] or 0, m.get("verified", 0), m.get("verify_n", 0), 1 if mine_bad else 0, 1]
    per_rank = [vals]
    k_ms_all = [m["k_ms"]]
    if dist is not None:
        gdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor(vals + [int(round(m["k_ms"] * 1000))], dtype=torch.int64, device=gdev)
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        per_rank = [x.cpu().tolist()[:6] for x in lst]
        k_ms_all = [x.cpu().tolist()[6] / 1000.0 for x in lst]
    placements = [placement]
    if dist is not None:
        pl = [None] * world
        dist.all_gather_object(pl, placement)
        placements = pl
    status_or = 0
    for v in per_rank:
        status_or |= v[0]
    gate_sum = sum(v[1] for v in per_rank)
    ver_ok, ver_n = sum(v[2] for v in per_rank), sum(v[3] for v in per_rank)
    ranks_failed = [r for r, v in enumerate(per_rank) if v[4]]
    ranks_seen = sum(v[5] for v in per_rank)
    failed = bool(ranks_failed) or ranks_seen != world
    if m.get("verify_bad") is not None and rank != 0:
        print("bench.py rank %d: %s" % (rank, json.dumps(m["verify_bad"])), file=sys.stderr, flush=True)

    if rank == 0:
        frames = S * F * args.steps * world
        fps = frames / m["dt"]
        k_ms, k_calls = m["k_ms"], m["k_calls"]
        out_per_frame = m["out_total"] / float(S * F)
        bytes_in = 1152 * 2 * 2                                          # int16 stereo
        alg_bytes = (bytes_in + out_per_frame) * S * F                   # per launch of the dominant kernel
        ach = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
        prof = pmc_profile(args.config, S, F)
        lib_id = api.build_id()
        stale = prof is not None and prof[0].get("build_id") != lib_id
        kernel = "k_alloc" if srs[0] >= 32000 else "k_alloc_lsf"
        if m.get("k6") and m["k6"]["kernel"]:
            kernel = m["k6"]["kernel"]          # the low-footprint build of the stream walk (picked by batch size)
        traffic = None
        valu = None
        if prof is not None and not stale:
            kc = prof[0]["kernels"].get(kernel, {})
            if "hbm_bytes_corrected" in kc:
                traffic = int(kc["hbm_bytes_corrected"])
            if "SQ_INSTS_VALU" in kc and k_ms > 0:
                # vector instructions the kernel issues per launch (committed SQ counters of this build and workload) over its
                # live-measured duration, against the chip's vector issue rate
                gi = kc["SQ_INSTS_VALU"] / (k_ms * 1e-3) / 1e9
                valu = {"bound": "valu_issue", "kernel": kernel, "achieved": round(gi, 2), "peak": VALU_PEAK_GINST, "unit": "Ginst/s (wave64 vector instructions)",
                        "frac": round(gi / VALU_PEAK_GINST, 4), "insts_per_launch": int(kc["SQ_INSTS_VALU"]),
                        # waves of this kernel resident per SIMD while the launch is full: 2 per stream (master + helper)
                        "waves_per_simd": round(2.0 * min(S, m["k6"]["resident_streams"]) / (4.0 * torch.cuda.get_device_properties(local).multi_processor_count), 2) if m.get("k6") else None,
                        "waves_launched": int(kc.get("SQ_WAVES", 0)),
                        "lds_bank_conflict_frac": round(kc["SQ_LDS_BANK_CONFLICT"] / kc["SQ_LDS_IDX_ACTIVE"], 4) if kc.get("SQ_LDS_IDX_ACTIVE") else None,
                        "valu_active_over_wave_cycles": round(kc["SQ_ACTIVE_INST_VALU"] / kc["SQ_WAVE_CYCLES"], 4) if kc.get("SQ_WAVE_CYCLES") else None,
                        "counters": prof[1]}
        sr0 = srs[0]
        rt = sum(1152.0 / sr for sr in srs) / S                           # mean seconds of audio per frame
        res = {
            "metric": "batched stereo 44.1kHz frames/sec (whole node), CBR-128" if args.config == 2 else "batched stereo frames/sec (whole node), BASELINE config %d" % args.config,
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(m["dt"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"] % (S, F), "baseline_config": args.config,
                       "streams_per_gpu": S, "frames_per_step": F, "parallelism": "streams sharded over %d GPU(s), no collective" % world},
            "x_realtime_per_gpu": round(fps / world * rt, 1),
            "ranks_seen": ranks_seen, "ranks_failed": ranks_failed,
            "kernel_status": status_or, "gate_timeouts": gate_sum,
            "bitstream_bytes_per_frame": round(out_per_frame, 2),
            "build_id": lib_id,
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 3) if ach else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6) if ach else None, "traffic": traffic,
                         "kernel_ms": round(k_ms, 3), "launches": k_calls,
                         "algorithmic_bytes_per_frame": round(bytes_in + out_per_frame, 1)},
        }
        if world > 1:
            res["roofline"]["kernel_ms_per_rank"] = [round(x, 3) for x in k_ms_all]
        if stale:
            # the committed counter profile was collected with another build of the kernels: not quoted
            res["roofline"]["stale_profile"] = True
            res["roofline"]["stale_profile_file"] = prof[1]
        if valu is not None:
            res["roofline_issue"] = valu
        # per-stream time inside the last timed step's allocator launch (ms): the launch - and at config 2 the step - ends with "max"
        res["stream_ms"] = m.get("stream_ms")
        if ver_n:
            res["verified_streams"] = ver_ok
            res["verify"] = {"checked": ver_n, "identical": ver_ok, "first_mismatch": m.get("verify_bad"),
                             "what": "last step's bitstream of randomly chosen streams of every rank's block vs the CPU oracle run over the same %d steps" % (args.warmup + args.steps)}
        if host_fed is not None:
            res["host_fed"] = host_fed
        if worst is not None:
            res["worst_case_value"] = round(S * F * args.steps / worst["dt"], 1)
            res["worst_case"] = {"signal": "inter-channel correlation cycled over {0.7, 0, 1, 0.3} by stream", "ms_per_step": round(worst["dt"] / args.steps * 1e3, 3),
                                 "kernel_ms": round(worst["k_ms"], 3), "kernel_status": worst["status"], "stream_ms": worst.get("stream_ms")}
        if m.get("k6"):
            res["roofline"]["resident_streams"] = m["k6"]["resident_streams"]
        try:        # which C library the checker on this box runs on (matters for first-generation-allocator streams only: hx_libm32.h)
            res["host_libm"] = api.libm_report(1000)
        except AttributeError:
            pass
        if others:
            oc = []
            for c, mo in others:
                fr = c["S"] * c["F"]
                opf = mo["out_total"] / float(fr)
                ach_o = (bytes_in + opf) * fr / (mo["k_ms"] * 1e-3) / 1e9 if mo["k_ms"] > 0 else None
                oc.append({"baseline_config": c["cfg"], "workload": c["wl"]["name"] % (c["S"], c["F"]), "value": round(fr * 4 * world / mo["dt"], 1), "unit": "frames/s",
                           "steps": 4, "warmup": 2, "ms_per_step": round(mo["dt"] / 4 * 1e3, 3), "kernel_ms": round(mo["k_ms"], 3),
                           "kernel_build": (mo["k6"]["kernel"] if mo.get("k6") and mo["k6"]["kernel"] else "k_alloc"),
                           "resident_streams": mo["k6"]["resident_streams"] if mo.get("k6") else None,
                           "roofline_frac": round(ach_o / HBM_PEAK_GBS, 6) if ach_o else None, "stream_ms": mo.get("stream_ms"),
                           "verified_streams": mo.get("verified_all", mo.get("verified")), "verify_checked": mo.get("verify_n_all", mo.get("verify_n")),
                           "kernel_status": mo.get("status_all", mo["status"]), "host_fed": mo.get("host_fed")})
            res["other_configs"] = oc
        if strong is not None:
            cs, ms_ = strong
            res["strong_scaling"] = {"baseline_config": 2, "streams_total": workload(2)["S"], "streams_this_rank": cs["S"], "frames_per_step": cs["F"], "steps": 8, "warmup": 2,
                                     "value": round(workload(2)["S"] * cs["F"] * 8 / ms_["dt"], 1), "unit": "frames/s", "ms_per_step": round(ms_["dt"] / 8 * 1e3, 3),
                                     "kernel_ms_rank0": round(ms_["k_ms"], 3), "kernel_status_rank0": ms_["status"],
                                     "what": "config 2's 1024 streams split over the %d ranks (total work fixed); the headline value is weak scaling (1024 streams per GPU)" % world}
        res["host_placement"] = placements      # per rank: device, its NUMA node, CPUs the rank was bound to, its own host-fed step time
        if not args.no_cpu_baseline:
            if cpus_before:
                os.sched_setaffinity(0, cpus_before)        # the CPU baseline runs on every usable core, not on this GPU's socket only
            # the reference CPU encoder on this box's host cores, in the same run (rank 0, after the timed regions; the
            # other ranks wait in the barrier below)
            res["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
