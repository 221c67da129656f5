"""Host runtime around the kernels, on a real MI355X: multi-device dispatcher, checkpoint blob checks,
argument validation of the host-buffer entry points, gate accounting, the bench's own multi-rank launch,
and the streaming route of the command line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from hmp3_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
RHOS = [0.7, 0.0, 1.0, 0.3]


def api():
    from hmp3_amd import api as a
    return a


def oracle_bytes(kw, pcm, nfr):
    enc = O.OracleEncoder(O.default_control(**kw))
    return b"".join(enc.encode_s16(pcm[f * 1152:(f + 1) * 1152]) for f in range(nfr))


def test_multi_device_dispatcher_blocks_streams_and_matches_oracle():
    """hx_multi_*: 7 streams in 3 contiguous blocks (3 + 2 + 2), one host thread per block.  On a one-GPU box the
    three blocks are three batches on device 0 - the host-side dispatch is the same."""
    a = api()
    kws = [dict(bitrate=64), dict(vbr_mnr=60), dict(bitrate=96, short_block_threshold=99999)]
    S, F = 7, 10
    pcm = np.stack([synth.stream_pcm(3100 + i, F, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    ctl = [a.default_control(**kws[i % 3]) for i in range(S)]
    m = a.Multi(ctl, max_frames=F, devices=[0, 0, 0])
    assert m.ndevices() == 3
    assert [m.shard(k)[1:] for k in range(3)] == [(0, 3), (3, 2), (5, 2)]
    first = m.encode_host(np.ascontiguousarray(pcm[:, :4 * 1152]))
    second = m.encode_host(np.ascontiguousarray(pcm[:, 4 * 1152:]))     # 6 frames: state carried per block
    assert m.status() == 0
    for s in range(S):
        assert first[s] + second[s] == oracle_bytes(kws[s % 3], pcm[s], F), s
    m.close()
    # more devices asked for than streams: one stream per block, the rest unused
    m = a.Multi(a.default_control(bitrate=64), nstreams=2, max_frames=4, devices=[0, 0, 0, 0])
    assert m.ndevices() == 2
    m.close()
    assert a.lib().hx_device_count() >= 1


def test_checkpoint_blob_is_refused_under_another_configuration_or_size():
    a = api()
    b1 = a.Batch(a.default_control(bitrate=64), nstreams=1, max_frames=4)
    b1.encode_host(synth.stream_pcm(1, 4)[None])
    blob = b1.get_stream_state(0)
    b2 = a.Batch(a.default_control(bitrate=96), nstreams=1, max_frames=4)       # other bitrate: other frame sizes / budgets
    with pytest.raises(RuntimeError, match="different configuration"):
        b2.set_stream_state(0, blob)
    with pytest.raises(ValueError):
        b1.set_stream_state(0, blob[:-8])
    bad = bytearray(blob); bad[0] ^= 0xFF
    with pytest.raises(RuntimeError, match="not a stream-state blob"):
        b1.set_stream_state(0, bytes(bad))
    b1.set_stream_state(0, blob)        # the genuine blob still goes in
    b1.close(); b2.close()


def test_host_entry_points_check_arguments_before_touching_the_device():
    a = api()
    L = a.lib()
    b = a.Batch(a.default_control(bitrate=64), nstreams=2, max_frames=4)
    pcm = np.zeros((2, 4 * 1152, 2), np.int16)
    out = np.zeros((2, b.out_stride(4)), np.uint8)
    nb = np.zeros(2, np.int32)
    stride = b.out_stride(4)
    for fn in (L.hx_batch_encode_s16_host, L.hx_batch_submit_s16_host):
        assert fn(b.h, pcm.ctypes.data, 0, out.ctypes.data, stride, nb.ctypes.data) != 0        # nframes <= 0
        assert fn(b.h, pcm.ctypes.data, -3, out.ctypes.data, stride, nb.ctypes.data) != 0
        assert fn(b.h, pcm.ctypes.data, 5, out.ctypes.data, stride, nb.ctypes.data) != 0        # > max_frames
        assert fn(b.h, None, 4, out.ctypes.data, stride, nb.ctypes.data) != 0
        assert fn(b.h, pcm.ctypes.data, 4, None, stride, nb.ctypes.data) != 0
        assert fn(b.h, pcm.ctypes.data, 4, out.ctypes.data, stride, None) != 0
        assert fn(b.h, pcm.ctypes.data, 4, out.ctypes.data, 64, nb.ctypes.data) != 0           # out_stride too small
        assert "out_stride" in a.last_error()
    assert L.hx_batch_out_stride(None, 4) == 0 and L.hx_batch_status(None) == -1
    # the batch is untouched: a good call gives the oracle's bytes
    p = np.stack([synth.stream_pcm(41, 4), synth.stream_pcm(42, 4)])
    got = b.encode_host(p)
    for s in range(2):
        assert got[s] == oracle_bytes(dict(bitrate=64), p[s], 4)
    assert b.status() == 0 and b.gate_timeouts() == 0
    b.close()
    with pytest.raises(RuntimeError, match="32 Mi"):
        a.Batch(a.default_control(bitrate=64), nstreams=1 << 20, max_frames=64)


def _bench(args, timeout=900, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_two_ranks_encode_disjoint_stream_blocks():
    """bench.py --gpus 2 launches two ranks itself; on this one-GPU box both use GPU 0 (--share-gpu) and the
    process group runs on gloo.  Each rank encodes its own block of streams and rank 0 verifies against the oracle."""
    rc, line, err = _bench(["--gpus", "2", "--share-gpu", "--backend", "gloo", "--streams", "24", "--frames", "8", "--steps", "2", "--warmup", "1",
                            "--verify", "6", "--no-cpu-baseline", "--no-worst-case"])
    assert rc == 0, (line, err[-2000:])
    assert line["n_gpus"] == 2 and line["kernel_status"] == 0 and line["ranks_seen"] == 2 and line["ranks_failed"] == []
    # every rank checks streams of its own block (6 asked for over two ranks: 3 each)
    assert line["verified_streams"] == line["verify"]["checked"] == 6 and line["verify"]["first_mismatch"] is None
    assert line["value"] > 0 and line["roofline"]["kernel_ms"] > 0


@pytest.mark.one_k6_build
def test_eight_rank_preflight_on_one_gpu():
    """Everything of the first real 8-GPU run except the devices (the driver has not had an 8-GPU node yet): eight
    processes started by bench.py itself, rendezvous on 127.0.0.1, NUMA binding, page-locked buffers and the host-fed passes,
    per-rank verification against the oracle, the shares of configs 4 and 5 (the configs BASELINE defines on 8 GPUs) with
    stream classes / correlations cycling across the rank boundaries, the strong-scaling split of config 2's 1024 streams
    (128 per rank), the max-over-ranks clock, exit codes - on one GPU (--share-gpu) over gloo, at 128 streams x 32 frames per rank."""
    import time
    t0 = time.time()
    rc, line, err = _bench(["--gpus", "8", "--share-gpu", "--backend", "gloo", "--streams", "128", "--frames", "32", "--steps", "2", "--warmup", "1",
                            "--verify", "16", "--no-cpu-baseline", "--no-worst-case", "--other-size", "128,32", "--strong-scaling", "1"], timeout=600)
    wall = time.time() - t0
    assert rc == 0, (line, err[-2000:])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["ranks_failed"] == [] and line["kernel_status"] == 0
    assert line["verify"]["checked"] == 16 and line["verify"]["identical"] == 16          # two streams of every rank's own block
    pl = line["host_placement"]
    assert len(pl) == 8 and sorted(p["rank"] for p in pl) == list(range(8)) and all(p["device"] == 0 for p in pl)
    assert all("host_fed_ms_per_step" in p and set(p.get("host_fed_ms_per_step_other", {})) == {"4", "5"} for p in pl), pl
    oc = {o["baseline_config"]: o for o in line["other_configs"]}
    assert set(oc) == {4, 5}
    for c in (4, 5):
        assert oc[c]["kernel_status"] == 0 and oc[c]["verified_streams"] == oc[c]["verify_checked"] == 16, oc[c]
        assert oc[c]["host_fed"]["kernel_status"] == 0 and oc[c]["host_fed"]["value"] > 0
    ss = line["strong_scaling"]
    assert ss["streams_total"] == 1024 and ss["streams_this_rank"] == 128 and ss["kernel_status_rank0"] == 0 and ss["value"] > 0
    assert line["host_fed"]["kernel_status"] == 0 and len(line["roofline"]["kernel_ms_per_rank"]) == 8
    assert wall < 240, "the pre-flight took %.0f s" % wall
    print("eight-rank pre-flight: %.0f s" % wall)


def test_bench_line_shows_a_failing_rank_and_exits_non_zero():
    """a failure on a rank other than 0 must not be invisible: the line names the rank, the exit code is non-zero"""
    rc, line, err = _bench(["--gpus", "2", "--share-gpu", "--backend", "gloo", "--streams", "8", "--frames", "4", "--steps", "1", "--warmup", "1",
                            "--verify", "2", "--no-cpu-baseline", "--no-worst-case", "--host-fed", "0"], extra_env={"HMP3AMD_BENCH_FAULT_RANK": "1"})
    assert rc != 0 and line is not None
    assert line["ranks_seen"] == 2 and line["ranks_failed"] == [1]


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_bench_configs_verify_against_oracle_at_reduced_size(cfg):
    rc, line, err = _bench(["--config", str(cfg), "--streams", "48", "--frames", "12", "--steps", "2", "--warmup", "1", "--verify", "12",
                            "--no-cpu-baseline", "--no-worst-case"])
    assert rc == 0, (line, err[-2000:])
    assert line["config"]["baseline_config"] == cfg and line["kernel_status"] == 0
    assert line["verified_streams"] == 12, line["verify"]


@pytest.mark.parametrize("name", ["cli_cbr128_s16_44k", "cli_vbr50_s24_44k", "cli_lsf_vbr50_f32_24k", "cli_rifx_cbr64_s16_44k", "cli_rifx_cbr64_s24_44k"])
def test_cli_streams_from_a_pipe_with_bounded_buffers(name, tmp_path):
    """`hmp3amd - out.mp3`: the input arrives on a pipe and is encoded frame by frame through a sliding window
    (memory independent of the input length); the file equals the one the reference CLI wrote from the WAV."""
    sys.path.insert(0, GOLD)
    import make_golden_cli as M
    seed, nsamp, sr, as_float, bursts, flags = M.CASES[name]
    wav, mp3 = str(tmp_path / "in.wav"), str(tmp_path / "out.mp3")
    M.write_wav(wav, M.case_pcm(name), sr, as_float, M.CONTAINER.get(name))
    cli = os.path.join(ROOT, "hmp3_amd", "hmp3amd")
    with open(wav, "rb") as f:
        r = subprocess.run([cli, "-", mp3] + flags + ["-EC"], stdin=f, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert b"ec->samprate" in r.stderr            # -EC prints the settings in use
    assert open(mp3, "rb").read() == open(os.path.join(GOLD, name + ".mp3"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("lpt", ["3", "0"])
def test_longest_first_workgroup_order_changes_nothing_in_the_output(monkeypatch, lpt):
    """Batches with more streams than the chip has CUs start their allocator workgroups by the previous call's stream
    durations, longest first (hx_cabi.hip k_order); forced here on a small batch (HMP3AMD_LPT=3), three calls so that the
    order is a real permutation, against the oracle - and the identity order (HMP3AMD_LPT=0) the same"""
    import numpy as np
    from hmp3_amd import api, synth
    from oracle import oracle as O
    monkeypatch.setenv("HMP3AMD_LPT", lpt)
    S, F, calls = 24, 12, 3
    rhos = [0.7, 0.0, 1.0, 0.3]
    pcm = np.stack([synth.stream_pcm(7700 + i, F * calls, rho=rhos[i % 4], bursts=(i % 3 == 0)) for i in range(S)])
    kw = dict(bitrate=64)
    b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    got = [b"" for _ in range(S)]
    for c in range(calls):
        out = b.encode_host(np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152]))
        for s in range(S):
            got[s] += out[s]
    assert b.status() == 0
    for s in range(S):
        enc = O.OracleEncoder(O.default_control(**kw))
        want = b"".join(enc.encode_s16(pcm[s, f * 1152:(f + 1) * 1152]) for f in range(F * calls))
        assert got[s] == want, "stream %d" % s
    b.close()


@pytest.mark.gpu
def test_submits_into_one_output_buffer_are_run_one_after_the_other():
    """The packing of a device-buffer submit is deferred behind the next submit's allocator launch; a caller that hands
    every submit the same output buffers must still get the last call's bytes right (the library then orders the
    packing in front of the next allocator launch), and a stream checkpoint taken with a packing job pending is complete"""
    import numpy as np
    import torch
    from hmp3_amd import api, synth
    S, F, calls = 40, 10, 5
    kw = dict(vbr_mnr=70)
    pcm = np.stack([synth.stream_pcm(8800 + i, F * calls, rho=[0.7, 0.0, 1.0, 0.3][i % 4], bursts=True) for i in range(S)])
    b0 = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    want = [b0.encode_host(np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152])) for c in range(calls)]
    b0.close()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    d_pcm = [torch.from_numpy(np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152])).to(dev) for c in range(calls)]
    b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    stride = b.out_stride(F)
    d_out = torch.zeros((S, stride), dtype=torch.uint8, device=dev)
    d_nb = torch.zeros((S,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for c in range(calls - 1):
        b.submit_device(d_pcm[c].data_ptr(), F, d_out.data_ptr(), stride, d_nb.data_ptr(), st)
    blob = b.get_stream_state(3)            # with the fourth call's packing still pending
    b.submit_device(d_pcm[calls - 1].data_ptr(), F, d_out.data_ptr(), stride, d_nb.data_ptr(), st)
    b.wait(st)
    torch.cuda.synchronize()
    assert b.status() == 0
    o, n = d_out.cpu().numpy(), d_nb.cpu().numpy()
    for s in range(S):
        assert o[s, :n[s]].tobytes() == want[calls - 1][s], "stream %d" % s
    # the checkpoint continues in another batch with the last call's bytes
    b2 = api.Batch(api.default_control(**kw), nstreams=1, max_frames=F)
    b2.set_stream_state(0, blob)
    got = b2.encode_host(np.ascontiguousarray(pcm[3:4, (calls - 1) * F * 1152:calls * F * 1152]))
    assert got[0] == want[calls - 1][3]
    b.close(); b2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw,nch", [(dict(bitrate=32, samprate=22050), 2), (dict(bitrate=64, mode=3), 1), (dict(bitrate=64, mode=2), 2),
                                    (dict(bitrate=8, samprate=16000), 2), (dict(vbr_mnr=40, samprate=24000), 2), (dict(vbr_mnr=90, hf_flag=3, samprate=48000), 2)],
                         ids=["lsf_cbr64", "mono_cbr64", "dual_channel", "lsf_intensity", "lsf_vbr", "vbr_hf2_48k"])
def test_submit_path_for_every_stream_type(kw, nch):
    """device-buffer submits back to back (front end, allocator launch and packing of neighbouring calls overlapped) against
    plain calls: MPEG-2 frame-per-granule streams, mono, both allocators, the HF modes"""
    import numpy as np
    import torch
    from hmp3_amd import api, synth
    S, F, calls = 48, 8, 5
    sr = kw.get("samprate", 44100)
    pcm = np.stack([synth.stream_pcm(9900 + i, F * calls, sr=sr, rho=[0.7, 0.0, 1.0, 0.3][i % 4], bursts=True) for i in range(S)])
    if nch == 1:
        pcm = np.ascontiguousarray(pcm[:, :, 0])
    cut = lambda c: np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152])
    b0 = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    want = [b0.encode_host(cut(c)) for c in range(calls)]
    assert b0.status() == 0
    b0.close()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    d_pcm = [torch.from_numpy(cut(c)).to(dev) for c in range(calls)]
    b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    stride = b.out_stride(F)
    d_out = [torch.zeros((S, stride), dtype=torch.uint8, device=dev) for _ in range(calls)]
    d_nb = [torch.zeros((S,), dtype=torch.int32, device=dev) for _ in range(calls)]
    torch.cuda.synchronize()
    for c in range(calls):      # back to back: the packing of call c goes out behind the allocator launch of call c + 1
        b.submit_device(d_pcm[c].data_ptr(), F, d_out[c].data_ptr(), stride, d_nb[c].data_ptr(), st)
    b.wait(st)
    torch.cuda.synchronize()
    got = []
    for c in range(calls):
        o, n = d_out[c].cpu().numpy(), d_nb[c].cpu().numpy()
        got.append([o[s, :n[s]].tobytes() for s in range(S)])
    assert b.status() == 0
    for c in range(calls):
        for s in range(S):
            assert got[c][s] == want[c][s], "call %d stream %d" % (c, s)
    b.close()


@pytest.mark.parametrize("args", [["120", "501"], ["--submit", "80", "502"], ["--a1", "60", "503"]], ids=["host_calls", "overlapped_submits", "first_generation_allocator"])
def test_fuzz_parity_slice(args):
    """fixed-seed slices of tools/fuzz_parity.py (the sweep that found the round-2 bit-writer defect): random controls x
    random and extreme signals in random-sized calls, GPU against the oracle"""
    if "--a1" in args:
        from conftest import skip_unless_host_libm_is_the_restated_one
        skip_unless_host_libm_is_the_restated_one()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_fuzz_mixed_class_batches_slice():
    """ten batches of tools/fuzz_mixed.py: every stream of a batch with its own control"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_mixed.py"), "10", "504"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_host_submit_behind_device_submits_sees_their_carried_frames():
    """A host-buffer submit right behind device-buffer submits on a small batch: the earlier submit's packing (which saves
    the incomplete frames' images into the stream state) goes out on the packing stream, and the host submit's own
    packing must wait for it (a missing wait let k_pack_pre read stale images)."""
    import torch
    a = api()
    S, F, calls = 6, 4, 6
    kw = dict(bitrate=64)
    pcm = np.stack([synth.stream_pcm(9900 + i, F * calls, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for trial in range(8):
        b = a.Batch(a.default_control(**kw), nstreams=S, max_frames=F)
        stride = b.out_stride(F)
        got = [b"" for _ in range(S)]
        keep = []
        host = []
        for c in range(calls):
            blk = np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152])
            if c % 2 == 0:      # device-buffer submit
                d_pcm = torch.from_numpy(blk).to(dev)
                d_out = torch.zeros((S, stride), dtype=torch.uint8, device=dev); d_nb = torch.zeros((S,), dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                b.submit_device(d_pcm.data_ptr(), F, d_out.data_ptr(), stride, d_nb.data_ptr(), st)
                keep.append((c, d_pcm, d_out, d_nb))
            else:               # host-buffer submit straight behind it
                h_pcm = torch.from_numpy(blk).pin_memory()
                h_out = torch.zeros((S, stride), dtype=torch.uint8).pin_memory(); h_nb = torch.zeros((S,), dtype=torch.int32).pin_memory()
                b.submit_host(h_pcm.data_ptr(), F, h_out.data_ptr(), stride, h_nb.data_ptr())
                host.append((c, h_pcm, h_out, h_nb))
        b.wait(st); b.wait_host(); torch.cuda.synchronize()
        assert b.status() == 0
        parts = {}
        for c, _, o, nb in keep + host:
            o, nb = o.cpu().numpy(), nb.cpu().numpy()
            parts[c] = [o[s, :nb[s]].tobytes() for s in range(S)]
        for s in range(S):
            assert b"".join(parts[c][s] for c in range(calls)) == oracle_bytes(kw, pcm[s], F * calls), (trial, s)
        b.close()


def test_cli_mnr_adjust_file_switch(tmp_path):
    """-W<file> (test/tomp3.cpp:552-555, get_mnr_adjust): 21 per-band offsets, clamped to +-200 and echoed; the reference
    encoder never reads them (bitallo3.cpp:441 is commented out), so the file equals the one written without the switch"""
    sys.path.insert(0, GOLD)
    import make_golden_cli as M
    name = "cli_cbr128_s16_44k"
    seed, nsamp, sr, as_float, bursts, flags = M.CASES[name]
    wav, mp3, adj = str(tmp_path / "in.wav"), str(tmp_path / "out.mp3"), str(tmp_path / "adj.txt")
    M.write_wav(wav, M.case_pcm(name), sr, as_float, M.CONTAINER.get(name))
    open(adj, "w").write("10 -300 250 7\n")
    r = subprocess.run([os.path.join(ROOT, "hmp3_amd", "hmp3amd"), wav, mp3] + flags + ["-W" + adj], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert b"MNR adjust  10 -200 200 7 0 0" in r.stderr
    assert open(mp3, "rb").read() == open(os.path.join(GOLD, name + ".mp3"), "rb").read()


def _ndev():
    return api().lib().hx_device_count()


@pytest.mark.skipif("_ndev() < 2", reason="needs two physical GPUs")
def test_multi_device_dispatcher_on_two_physical_gpus():
    """the same dispatcher over two real devices (skipped on a one-GPU box): blocks on device 0 and 1, bytes against the oracle"""
    a = api()
    kw = dict(bitrate=64)
    S, F = 9, 12
    pcm = np.stack([synth.stream_pcm(4100 + i, F, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    m = a.Multi(a.default_control(**kw), nstreams=S, max_frames=F, devices=[0, 1])
    assert m.ndevices() == 2 and [m.shard(k)[0] for k in range(2)] == [0, 1]
    got = m.encode_host(pcm)
    assert m.status() == 0
    for s in range(S):
        assert got[s] == oracle_bytes(kw, pcm[s], F), s
    m.close()


@pytest.mark.skipif("_ndev() < 2", reason="needs two physical GPUs")
def test_bench_two_ranks_on_two_physical_gpus():
    """bench.py --gpus 2 as the driver runs it at N = 2: one rank per device over RCCL, every rank verified, health gathered"""
    rc, line, err = _bench(["--gpus", "2", "--streams", "64", "--frames", "16", "--steps", "3", "--warmup", "1", "--verify", "8",
                            "--no-cpu-baseline", "--no-worst-case", "--host-fed", "1"])
    assert rc == 0, (line, err[-2000:])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["ranks_failed"] == [] and line["kernel_status"] == 0
    assert line["verified_streams"] == line["verify"]["checked"] == 8 and line["host_fed"]["value"] > 0
    assert len(line["roofline"]["kernel_ms_per_rank"]) == 2


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hmp3")), reason="oracle/_ref/hmp3 (the reference's CLI, prebuilt) not present")
def test_fuzz_cli_slice_against_the_reference_binary():
    """a slice of tools/fuzz_cli.py: random WAVs (rates incl. the converter's, sample formats, containers) x random flags,
    hmp3amd against the reference's own command line, whole files.  (The sweep found the reference's conditional
    end-of-input padding: 32-bit samples converted down by more than 1.14 never get the four calls of silence.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cli.py"), "60", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_stream_walk_build_is_chosen_by_batch_size_and_both_give_the_same_bytes(monkeypatch):
    """more streams than the chip holds at once -> k_alloc_slim (six streams per CU), otherwise k_alloc (four);
    HMP3AMD_K6 overrides.  The same PCM through both builds: identical bytes, identical to the oracle."""
    a = api()
    kw = dict(bitrate=64)
    monkeypatch.delenv("HMP3AMD_K6", raising=False)
    small = a.Batch(a.default_control(**kw), nstreams=64, max_frames=4)
    assert small.k6_variant() == 0 and small.resident_streams() % 4 == 0
    cus = small.resident_streams() // 4
    small.close()
    S, F = 4 * cus + 8, 4
    pcm = np.stack([synth.stream_pcm(900 + (i % 24), F, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    big = a.Batch(a.default_control(**kw), nstreams=S, max_frames=F)
    assert big.k6_variant() == 1 and big.resident_streams() == 6 * cus
    got_slim = big.encode_host(pcm)
    assert big.status() == 0
    big.close()
    monkeypatch.setenv("HMP3AMD_K6", "fat")
    fat = a.Batch(a.default_control(**kw), nstreams=S, max_frames=F)
    assert fat.k6_variant() == 0 and fat.resident_streams() == 4 * cus
    got_fat = fat.encode_host(pcm)
    assert fat.status() == 0
    fat.close()
    assert got_fat == got_slim
    for s_ in range(24):
        assert got_slim[s_] == oracle_bytes(kw, pcm[s_], F), s_
    # MPEG-2 batches have one kernel: the switch is ignored
    monkeypatch.setenv("HMP3AMD_K6", "slim")
    lsf = a.Batch(a.default_control(bitrate=32, samprate=22050), nstreams=8, max_frames=4)
    assert lsf.k6_variant() == 0
    lsf.close()


def test_soak_slice_six_thousand_frames_in_calls_of_1_to_64():
    """a slice of tools/soak.py in the suite: CBR-128 and VBR-50 (block switching), 2 streams each (one with bursts), 6000
    frames = 2.6 minutes of audio per stream through calls of 1 .. 64 frames on one batch object, against the oracle"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "6000", "2", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert out.count(" ok") == 2 and "6000 frames x 2 streams" in out


def test_host_placement_helpers_are_best_effort_and_harmless(k6_build):
    """hx_device_numa_node / hx_bind_thread_to_device: the node of device 0 (or -1), and binding the calling thread to its
    CPUs never enlarges the thread's CPU set and leaves it usable (bench.py does this in every rank before it allocates
    page-locked buffers); an encode afterwards still matches the oracle"""
    a = api()
    before = os.sched_getaffinity(0)
    try:
        node = a.device_numa_node(0)
        n = a.bind_thread_to_device(0)
        after = os.sched_getaffinity(0)
        assert node >= -1 and n >= 0
        assert after <= before and len(after) >= 1
        assert n in (0, len(after))
        assert a.device_numa_node(99) == -1 and a.bind_thread_to_device(99) == 0      # no such device: nothing happens
        kw = dict(bitrate=64)
        pcm = np.stack([synth.stream_pcm(70 + i, 6, rho=RHOS[i % 4]) for i in range(4)])
        b = a.Batch(a.default_control(**kw), nstreams=4, max_frames=6)
        got = b.encode_host(pcm)
        assert b.status() == 0 and all(got[s] == oracle_bytes(kw, pcm[s], 6) for s in range(4))
        b.close()
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.gpu
def test_k6_build_switch_takes_only_its_two_values(monkeypatch):
    from hmp3_amd import api
    monkeypatch.setenv("HMP3AMD_K6", "Slim")
    with pytest.raises(Exception):
        api.Batch(api.default_control(bitrate=64), nstreams=2, max_frames=2)
