"""Multi-process path of bench.py / multi-GPU runs on CPU: world_size 2, gloo.  Streams are
sharded over ranks with no data-path collective; only the timing uses a max-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hmp3_amd import shard


def test_shard_ranges_partition_the_streams():
    for n in (1, 2, 7, 1024, 32768, 1000):
        for w in (1, 2, 3, 4, 8):
            seen = []
            for r in range(w):
                a, b = shard.shard_range(n, w, r)
                seen += list(range(a, b))
            assert seen == list(range(n))
            if n >= w:
                sizes = [shard.shard_range(n, w, r)[1] - shard.shard_range(n, w, r)[0] for r in range(w)]
                assert max(sizes) - min(sizes) <= 1
    assert shard.owner_of(5, 8, 2) == 1


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from hmp3_amd import synth
    n, nfr = 6, 6
    a, b = shard.shard_range(n, world, rank)
    sizes = torch.zeros(n, dtype=torch.int64)
    for s in range(a, b):       # each rank encodes only its own streams (oracle stands in for the GPU path)
        enc = O.OracleEncoder(O.default_control(bitrate=64, short_block_threshold=99999))
        sizes[s] = len(O.encode_stream(enc, synth.stream_pcm(s, nfr)))
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # bench.py: max over ranks of the timed region
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM)    # test-only gather of the results
    if rank == 0:
        ret["tmax"] = float(t.item())
        ret["sizes"] = sizes.tolist()
    dist.destroy_process_group()


def test_two_ranks_gloo_cover_all_streams_once():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["tmax"] == 1.5
    assert all(v > 0 for v in ret["sizes"]) and len(ret["sizes"]) == 6


def _run_bench(args, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` starts two ranks by itself (no torchrun); here without a GPU:
    rendezvous, sharding and the reporting path only (--dry-run), process group on gloo"""
    rc, line, err = _run_bench(["--gpus", "2", "--dry-run", "--backend", "gloo"])
    assert rc == 0, err
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["streams_total"] == 2 * 1024
    assert line["max_time_token"] == 2.0            # max over ranks of (1 + rank)
    # what an N-rank line carries beside `value`: the configs BASELINE defines on 8 GPUs, their host-fed rates, strong scaling
    assert line["plan"] == {"other_configs": [4, 5], "host_fed_configs": [2, 4, 5], "strong_scaling": True, "worst_case": False}
    assert line["strong_scaling_blocks"] == [[0, 512], [512, 1024]]


def test_one_gpu_line_plans_every_other_config_with_its_host_fed_rate():
    rc, line, err = _run_bench(["--dry-run"])
    assert rc == 0, err
    assert line["plan"] == {"other_configs": [3, 4, 5], "host_fed_configs": [2, 3, 4, 5], "strong_scaling": False, "worst_case": True}
    rc, line, err = _run_bench(["--dry-run", "--config", "3"])
    assert rc == 0 and line["plan"]["other_configs"] == [] and line["plan"]["host_fed_configs"] == []


def test_eight_rank_config_2_line_plans_configs_4_and_5_and_the_strong_scaling_split():
    """the first 8-GPU run of the driver (`bench.py --gpus 8`, config 2): weak-scaling headline, plus other_configs 4 and 5 (each
    rank its 4096-stream share), their host-fed rates, and config 2's 1024 streams split eight ways"""
    rc, line, err = _run_bench(["--gpus", "8", "--dry-run", "--backend", "gloo"], timeout=900)
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["streams_total"] == 8 * 1024
    assert line["plan"] == {"other_configs": [4, 5], "host_fed_configs": [2, 4, 5], "strong_scaling": True, "worst_case": False}
    assert line["strong_scaling_blocks"] == [[128 * r, 128 * (r + 1)] for r in range(8)]


def test_bench_rejects_world_size_that_differs_from_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


import pytest


@pytest.mark.parametrize("cfg", [4, 5])
def test_eight_ranks_take_eight_disjoint_blocks_and_the_class_cycles_run_on_across_them(cfg):
    """world 8 on gloo, dry run: BASELINE config 4 (32768 streams = 8 x 4096) and config 5 (8 x 4096, three sample rates by
    stream mod 3, correlation by stream mod 4).  Every rank reports its block; the blocks tile the global stream range in
    rank order, and a stream's class and correlation follow from its GLOBAL index, so the cycles continue across the rank
    boundaries (4096 is not a multiple of 3: a rank's first stream is not class 0)."""
    rc, line, err = _run_bench(["--gpus", "8", "--dry-run", "--backend", "gloo", "--config", str(cfg)], timeout=900)
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["streams_total"] == 8 * 4096
    assert line["max_time_token"] == 8.0            # max over ranks of (1 + rank)
    blocks = sorted(line["blocks"], key=lambda b: b["rank"])
    assert [b["rank"] for b in blocks] == list(range(8))
    ncls, rho = line["nclasses"], line["rho_cycle"]
    assert ncls == (3 if cfg == 5 else 1) and len(rho) == (4 if cfg == 5 else 1)
    nxt = 0
    for b in blocks:
        assert b["first"] == nxt and b["count"] == 4096         # contiguous, disjoint, in rank order
        for k in range(6):
            g = b["first"] + k
            assert b["cls_head"][k] == g % ncls and b["rho_head"][k] == rho[g % len(rho)]
            g = b["first"] + b["count"] - 6 + k
            assert b["cls_tail"][k] == g % ncls and b["rho_tail"][k] == rho[g % len(rho)]
        nxt += b["count"]
    assert nxt == 32768
    if cfg == 5:
        assert [b["cls_head"][0] for b in blocks] == [(4096 * r) % 3 for r in range(8)]      # 0, 1, 2, 0, ...: not reset per rank
