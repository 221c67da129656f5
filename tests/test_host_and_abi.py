"""Host-side logic of the product and the C ABI surface.  CPU only: no compute call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hmp3_amd", "libhmp3amd.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="hmp3_amd/libhmp3amd.so not built (hmp3_amd/build.sh)")

TABLES = ["psy_w", "psy_cnt", "psy_off", "psy_nsum", "psy_npart", "win", "csa",
          "look_gain", "look_34igain", "look_ix43", "look_log_cbwmb", "nBand_l", "startBand_l", "nsf", "taperNT",
          "head", "ec", "scalars"]
CONFIGS = [dict(bitrate=64), dict(), dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000),
           dict(samprate=32000, bitrate=64), dict(bitrate=96, mode=0), dict(bitrate=160), dict(vbr_mnr=120, quick=0),
           dict(bitrate=64, mode=3), dict(mode=3, samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), dict(bitrate=48, mode=3, samprate=32000),
           # MPEG-2 LSF rates
           dict(bitrate=32, samprate=22050), dict(bitrate=24, samprate=16000), dict(bitrate=80, samprate=24000, mode=0),
           dict(samprate=22050), dict(samprate=16000, vbr_mnr=0), dict(samprate=24000, vbr_mnr=150, vbr_br_limit=64),
           dict(bitrate=8, samprate=16000, mode=3), dict(samprate=24000, mode=3, vbr_mnr=90)]


def test_library_exports_every_declared_symbol():
    from hmp3_amd import api
    hdr = open(os.path.join(ROOT, "include", "hmp3_amd.h")).read()
    declared = set(re.findall(r"\b(hx_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = C.CDLL(LIB)
    for name in sorted(declared):
        assert hasattr(lib, name), "include/hmp3_amd.h declares %s but the library does not export it" % name
    assert set(api.EXPORTS) <= declared


@pytest.mark.parametrize("kw", CONFIGS, ids=[str(i) for i in range(len(CONFIGS))])
def test_host_tables_equal_oracle_tables(kw):
    from hmp3_amd import api
    l = O.lib()
    l.hxo_debug_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong]
    l.hxo_debug_table.restype = C.c_longlong
    eo = O.OracleEncoder(O.default_control(**kw))
    eg = api.default_control(**kw)
    for n in TABLES:
        a = np.zeros(16384, np.uint8); b = np.zeros(16384, np.uint8)
        na = l.hxo_debug_table(eo.h, n.encode(), a.ctypes.data, a.nbytes)
        nb = api.lib().hx_debug_host_table(C.byref(eg), n.encode(), b.ctypes.data, b.nbytes)
        assert na == nb and na > 0, n
        assert np.array_equal(a[:na], b[:nb]), "table %s differs from the oracle's" % n


def test_transform_constants_equal_the_oracles_in_their_own_layout():
    """the product keeps the DCT / MDCT constants in its own layout (hx_types.h): heap-ordered twiddles of the
    32-point DCT, and the 9-point cosine transform's rows split by the half they act on"""
    from hmp3_amd import api
    l = O.lib()
    l.hxo_debug_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong]
    l.hxo_debug_table.restype = C.c_longlong
    eo = O.OracleEncoder(O.default_control(bitrate=64))
    eg = api.default_control(bitrate=64)

    def orc(name, n):
        a = np.zeros(n, np.float32)
        assert l.hxo_debug_table(eo.h, name.encode(), a.ctypes.data, a.nbytes) == a.nbytes, name
        return a.view(np.uint32)

    def prod(name, n):
        a = np.zeros(n, np.float32)
        assert api.lib().hx_debug_host_table(C.byref(eg), name.encode(), a.ctypes.data, a.nbytes) == a.nbytes, name
        return a.view(np.uint32)
    tw, coef = prod("dct_tw", 32), orc("dct_coef", 31)
    for N in (2, 4, 8, 16, 32):
        assert np.array_equal(tw[N // 2:N], coef[32 - N:32 - N + N // 2]), N     # the oracle stores the levels largest first
    assert np.array_equal(prod("mdct_pre18", 18), orc("m18_w", 18))
    assert np.array_equal(prod("mdct_odd18", 9), orc("m18_w2", 9))
    c = orc("m18_c", 36).reshape(9, 4)
    assert np.array_equal(prod("dct9_even", 12).reshape(3, 4), c[[2, 4, 8]])
    assert np.array_equal(prod("dct9_odd", 12).reshape(3, 4), c[[1, 5, 7]])
    assert prod("dct9_k3", 1)[0] == c[3, 0]
    assert np.array_equal(prod("mdct_pre6", 6), orc("m6_v", 6))
    assert np.array_equal(prod("mdct_odd6", 3), orc("m6_v2", 3))
    assert prod("dct3_k", 1)[0] == orc("m6_c87", 1)[0]


def test_resolve_rejects_what_reference_rejects_and_out_of_scope():
    from hmp3_amd import api
    buf = np.zeros(64, np.uint8)

    def ok(**kw):
        return api.lib().hx_debug_host_table(C.byref(api.default_control(**kw)), b"scalars", buf.ctypes.data, 64) > 0
    assert ok(bitrate=64) and ok() and ok(bitrate=48)
    assert not ok(bitrate=40)               # mp3enc.cpp:346-351
    assert not ok(bitrate=64, layer=2)      # mp3enc.cpp:388
    assert ok(bitrate=64, mode=3)           # mono
    assert ok(bitrate=64, mode=2)           # dual channel (first-generation allocator)
    assert not ok(bitrate=40, mode=2)       # ... under the same CBR floor
    assert ok(bitrate=32, samprate=22050)       # MPEG-2 LSF
    assert ok(bitrate=8, samprate=16000)        # 16 kbps joint stereo: intensity stereo (first-generation allocator)
    assert ok(bitrate=64, nsbstereo=8)          # intensity stereo on request at an MPEG-1 rate
    assert not ok(bitrate=64, samprate=5000)    # nearest entry of the rate table is a reserved index


def test_no_gpu_means_loud_failure_not_fallback():
    """on a box without a GPU the create call must fail (no silent CPU path)"""
    import torch
    from hmp3_amd import api
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        api.Batch(api.default_control(bitrate=64), nstreams=2, max_frames=2)
    e = api.Mp3Enc()
    assert e.L3_audio_encode_init(api.default_control(bitrate=64)) == 0


def test_libm32_restatement_equals_this_machines_libm(tmp_path):
    """hmp3_amd/csrc/hx_libm32.h (what the first-generation allocator's kernels use for the reference's log10f / logf
    calls) against the libm the oracle and the reference link with, on every 997th positive normal float
    (tools/check_libm32.c without -DSTEP checks all of them)"""
    import subprocess
    exe = str(tmp_path / "check_libm32")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-DSTEP=997", "-o", exe, os.path.join(ROOT, "tools", "check_libm32.c"), "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_host_tables_have_the_structure_the_low_footprint_kernel_derives_them_from():
    """k_alloc_slim keeps 4 + 16 constants in place of the 128-entry gain tables and the 256-entry x^(3/4) exponent
    table (ldexp of the mantissa periods) and the mB tables as 16-bit values: the identity is checked on the host's
    own tables, for every BASELINE configuration's stream classes, here and in hx_batch_create"""
    from hmp3_amd import api
    for kw in (dict(bitrate=64), dict(), dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), dict(bitrate=64, samprate=32000),
               dict(bitrate=64, samprate=48000), dict(bitrate=160), dict(bitrate=64, mode=3)):
        assert api.lib().hx_debug_slim_tables_ok(C.byref(api.default_control(**kw))) == 1, kw
    assert api.lib().hx_debug_slim_tables_ok(C.byref(api.default_control(bitrate=40))) == -1      # rejected configuration
    # the same identities on the tables themselves, as the kernel evaluates them
    eg = api.default_control(bitrate=64)

    def tab(name, n, dt=np.float32):
        a = np.zeros(n, dt)
        assert api.lib().hx_debug_host_table(C.byref(eg), name.encode(), a.ctypes.data, a.nbytes) == a.nbytes
        return a
    gain, igain, pexp = tab("look_gain", 128), tab("look_34igain", 128), tab("pow34_exp", 256)
    k = np.arange(128) - 8
    assert np.array_equal(np.ldexp(gain[8 + (k & 3)], k >> 2).astype(np.float32).view(np.uint32), gain.view(np.uint32))
    assert np.array_equal(np.ldexp(igain[8 + (k & 15)], -3 * (k >> 4)).astype(np.float32).view(np.uint32), igain.view(np.uint32))
    e = 3 * (np.arange(1, 255) - 127)
    assert np.array_equal(np.ldexp(gain[8 + (e & 3)], e >> 2).astype(np.float32).view(np.uint32), pexp[1:255].view(np.uint32))
    assert pexp[0] == 0 and np.isinf(pexp[255])
    mblog = tab("mblog", 256, np.int32)
    assert mblog.min() + 38227 >= 0 and mblog.max() + 38227 <= 301


def test_a_thread_bound_to_one_node_can_be_bound_to_another():
    """hx_bind_thread_to_node / _device intersect a node's CPU list with the CPUs the PROCESS may use (captured when the library
    was loaded), not with the calling thread's current mask: a thread that was narrowed before - here to one CPU, as if bound to
    another device's node - is widened to the whole node again (round 4's advisor finding: the second bind did nothing)."""
    from hmp3_amd import api
    if not hasattr(os, "sched_getaffinity") or not os.path.exists("/sys/devices/system/node/node0/cpulist"):
        pytest.skip("no sysfs NUMA description on this host")
    api.lib()                                   # library loaded: the process mask is captured now
    before = os.sched_getaffinity(0)
    spec = open("/sys/devices/system/node/node0/cpulist").read().strip()
    node0 = set()
    for part in spec.split(","):
        a, _, z = part.partition("-")
        node0 |= set(range(int(a), int(z or a) + 1))
    want = node0 & before
    if len(want) < 2:
        pytest.skip("fewer than two usable CPUs on node 0")
    try:
        os.sched_setaffinity(0, {min(want)})
        n = api.bind_thread_to_node(0)
        assert n == len(want) and os.sched_getaffinity(0) == want
        assert api.bind_thread_to_node(-1) == 0 and os.sched_getaffinity(0) == want      # unknown node: left as it was
    finally:
        os.sched_setaffinity(0, before)


def test_a_process_narrowed_after_the_library_was_loaded_is_followed():
    """The CPU set the bind calls intersect a node's list with is captured when the library is loaded; a process that is narrowed
    afterwards (taskset -p, a launcher's sched_setaffinity) has hx_refresh_process_cpus to take it again, and a bind whose mask
    the kernel refuses retries with the refreshed set by itself (round 5's advisor finding).  Runs in a child process: the
    main thread's mask is what is re-read, and pytest's own must stay as it is."""
    import subprocess, sys, textwrap
    if not hasattr(os, "sched_getaffinity") or not os.path.exists("/sys/devices/system/node/node0/cpulist"):
        pytest.skip("no sysfs NUMA description on this host")
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        from hmp3_amd import api
        api.lib()                                   # loaded with the full mask
        full = os.sched_getaffinity(0)
        spec = open("/sys/devices/system/node/node0/cpulist").read().strip()
        node0 = set()
        for part in spec.split(","):
            a, _, z = part.partition("-")
            node0 |= set(range(int(a), int(z or a) + 1))
        want = sorted(node0 & full)
        if len(want) < 3:
            print("SKIP"); sys.exit(0)
        narrow = set(want[:2])
        os.sched_setaffinity(0, narrow)             # the main thread = the process, as a launcher would
        assert api.refresh_process_cpus() == 2
        os.sched_setaffinity(0, {want[0]})
        assert api.bind_thread_to_node(0) == 2 and os.sched_getaffinity(0) == narrow, os.sched_getaffinity(0)
        print("OK")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    if "SKIP" in r.stdout:
        pytest.skip("fewer than three usable CPUs on node 0")
    assert "OK" in r.stdout
