"""Build checks that need no GPU: hipcc cross-compiles the kernels for gfx950 and the listings are read."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_flat_instruction_touches_lds_in_any_kernel():
    """the kernels' wave-local LDS hand-overs order DS instructions only (no s_waitcnt): a generic pointer into LDS in an
    out-of-line function would compile to FLAT accesses, which that order does not cover (tools/check_lds_flat.py)"""
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this host")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_flat.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok (8 translation units)" in r.stdout


def test_flat_check_compiles_with_the_flags_of_the_build_script():
    """tools/check_lds_flat.py restates the per-unit flags of hmp3_amd/build.sh: the optimisation level, scheduler strategy
    and MachineLICM switch of every translation unit must be the build's"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_flat.py"), "--print-flags"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    units = {l.split()[0]: l.split()[2:] for l in r.stdout.strip().splitlines()}
    sh = open(os.path.join(ROOT, "hmp3_amd", "build.sh")).read()
    assert '${HX_ALLOC_OPT:--O2}' in sh and 'ALLOC_SCHED="${HX_ALLOC_SCHED-$ILP}"' in sh and 'NOLICM="${HX_NOLICM--mllvm -disable-machine-licm}"' in sh
    for u in ("alloc", "alloc_slim", "alloc_lsf", "alloc1", "alloc1_lsf"):
        assert units[u] == ["-O2", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-disable-machine-licm"], u
        assert "for f in hx_alloc hx_alloc_slim hx_alloc_lsf hx_alloc1 hx_alloc1_lsf" in sh
    assert "-fno-slp-vectorize $NOLICM $HX_FRONT_EXTRA -DHX_FRONT_PART=1" in sh and units["front1"] == ["-O3", "-fno-slp-vectorize", "-DHX_FRONT_PART=1", "-mllvm", "-disable-machine-licm"]
    assert "-fno-slp-vectorize $ILP $HX_FRONT_EXTRA -DHX_FRONT_PART=2" in sh and units["front2"] == ["-O3", "-fno-slp-vectorize", "-DHX_FRONT_PART=2", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
    assert "${HX_OPT:--O3}" in sh and units["pack"] == ["-O3"]
