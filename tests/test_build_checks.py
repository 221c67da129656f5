"""Build checks that need no GPU: hipcc cross-compiles the kernels for gfx950 and the listings are read."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_flat_instruction_touches_lds_in_any_kernel():
    """the kernels' wave-local LDS hand-overs order DS instructions only (no s_waitcnt): a generic pointer into LDS in an
    out-of-line function would compile to FLAT accesses, which that order does not cover (tools/check_lds_flat.py)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_flat.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok (8 translation units)" in r.stdout
