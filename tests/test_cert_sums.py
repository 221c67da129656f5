"""Certified band sums of the stream walk (hmp3_amd/csrc/hx_dev.h): the interval arithmetic that lets the kernels replace the
reference's strict left-to-right fp32 band sums (l3math.c:521-537) where their only consumer is mbLogC (l3math.c:228-242), and
the line-run tables the host cuts for it.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hmp3_amd", "libhmp3amd.so")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cert") / "cert_sums_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "cert_sums_check.c"), "-lm"])
    return exe


@pytest.mark.parametrize("seed", [1, 2])
def test_strict_sum_lies_inside_the_certified_interval(checker, seed):
    """>= 10^6 random and adversarial vectors of non-negative terms (n <= 192, runs of 2..10, wide dynamic range, subnormals,
    rounding-direction worst cases): the strict fp32 sum is inside [t (1 - c u), t (1 + c u)] of the kernels' tree sum t, a
    certified mbLogC bucket is the strict sum's, and the same for the quotient of two sums and for the four logarithms of the
    stereo metric (k_spec: two non-negative sums that start at 100 and a signed one, whose interval comes from the sum of the
    terms' magnitudes)"""
    r = subprocess.run([checker, "600000", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "outside 0" in r.stdout and "quotient outside 0" in r.stdout, r.stdout
    # the stereo metric's four certified buckets (signed sum included) on correlated channel pairs of every kind
    assert "metric bands 300000  wrong 0 " in r.stdout, r.stdout


RUN_CONFIGS = [dict(bitrate=64), dict(), dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), dict(samprate=32000, bitrate=64),
               dict(bitrate=96, mode=0), dict(bitrate=64, mode=3), dict(bitrate=32, samprate=22050), dict(bitrate=24, samprate=16000),
               dict(samprate=24000, vbr_mnr=150, vbr_br_limit=64), dict(bitrate=8, samprate=16000, mode=3), dict(bitrate=48, mode=3, samprate=32000),
               dict(bitrate=64, freq_limit=4000), dict(bitrate=64, nsb_limit=4, samprate=48000)]


@pytest.mark.skipif(not os.path.exists(LIB), reason="hmp3_amd/libhmp3amd.so not built (hmp3_amd/build.sh)")
@pytest.mark.parametrize("kw", RUN_CONFIGS, ids=[str(i) for i in range(len(RUN_CONFIGS))])
def test_line_runs_tile_the_measured_bands(kw):
    """HxParams::lane_run: every line of every band the gain search measures belongs to exactly one
    lane's run, a run stays inside one band, a band's lanes are neighbours (at most 16: the segmented scan crosses one row
    boundary at most), and the 'lanes back to the band's first lane' field counts them"""
    from hmp3_amd import api
    ec = api.default_control(**kw)

    def tab(name, dtype, count):
        a = np.zeros(count, dtype)
        n = api.lib().hx_debug_host_table(C.byref(ec), name.encode(), a.ctypes.data, a.nbytes)
        assert n == a.nbytes, name
        return a
    start, width, nsf = tab("startBand_l", np.int32, 24), tab("nBand_l", np.int32, 22), tab("nsf", np.int32, 2)
    nchan = int(tab("nchan", np.int32, 1)[0])
    W = int(tab("run_w", np.int32, 1)[0])
    run, last = tab("lane_run", np.uint16, 64), tab("band_last_lane", np.uint8, 24)
    nb = max(int(nsf[0]), int(nsf[1]) if nchan == 2 else 0)
    assert 2 <= W <= 10 and W % 2 == 0

    def check(run, lasts, bands, W, chbit):
        owner = {}
        used = 0
        while used < 64 and ((int(run[used]) >> 9) & 7):
            used += 1
        for l in range(used):
            r = int(run[l])
            st, cnt = (r & 511) * 2, ((r >> 9) & 7) * 2
            d, ch = ((r >> 12) & 7, r >> 15) if chbit else (r >> 12, 0)
            assert 0 < cnt <= W
            b = [k for k in range(22) if start[k] <= st < start[k] + width[k]][0]
            assert st + cnt <= start[b] + width[b], "a run leaves its band"
            assert (ch, b) in bands
            for j in range(st, st + cnt):
                assert (ch, j) not in owner
                owner[(ch, j)] = l
            # lanes back to the band's first lane
            first = l - d
            rf = int(run[first])
            assert (rf & 511) * 2 == start[b] and (not chbit or (rf >> 15) == ch) and d < 16
            for m in range(first, l + 1):
                assert start[b] <= (int(run[m]) & 511) * 2 < start[b] + width[b] and (not chbit or (int(run[m]) >> 15) == ch)
        for (ch, b) in bands:
            for j in range(start[b], start[b] + width[b]):
                assert (ch, j) in owner, "line %d of band %d has no lane" % (j, b)
            ll = int(lasts[ch][b])
            r = int(run[ll])
            assert (r & 511) * 2 + ((r >> 9) & 7) * 2 == start[b] + width[b] and (not chbit or (r >> 15) == ch), "band_last_lane"
        for l in range(used, 64):
            assert int(run[l]) == 0

    check(run, [last], [(0, b) for b in range(nb)], W, 0)


@pytest.mark.skipif(not os.path.exists(LIB), reason="hmp3_amd/libhmp3amd.so not built (hmp3_amd/build.sh)")
def test_the_tightest_rate_uses_all_64_lanes_and_no_run_reads_past_the_spectrum():
    """32 kHz with all 21 long bands measured is the layout's edge: runs of W = 10 lines take exactly 64 of the 64 lanes
    (hx_host.cpp band_runs).  Pinned so that a change of the band tables, of nsf or of the run rules that pushes it over shows
    up here as a named failure and not as hx_batch_create returning null; and for every supported rate the kernels' unconditional
    pair reads from a run's first line (sweep_load, isf2_run, msmetric_unit: W lines whatever the run's length) stay inside the
    channel's 576 lines."""
    from hmp3_amd import api

    def tab(ec, name, dtype, count):
        a = np.zeros(count, dtype)
        n = api.lib().hx_debug_host_table(C.byref(ec), name.encode(), a.ctypes.data, a.nbytes)
        assert n == a.nbytes, (name, api.last_error())
        return a
    ec = api.default_control(samprate=32000, bitrate=64)
    W = int(tab(ec, "run_w", np.int32, 1)[0])
    run = tab(ec, "lane_run", np.uint16, 64)
    used = sum(1 for r in run if (int(r) >> 9) & 7)
    assert (W, used) == (10, 64), "32 kHz: W = %d, %d lanes used (the layout's edge moved)" % (W, used)
    for kw in RUN_CONFIGS:
        ec = api.default_control(**kw)
        W = int(tab(ec, "run_w", np.int32, 1)[0])
        for r in tab(ec, "lane_run", np.uint16, 64):
            if (int(r) >> 9) & 7:
                assert (int(r) & 511) * 2 + W <= 576, kw
