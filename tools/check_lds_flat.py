"""Build check: no FLAT load / store in the front-end and packing kernels, none in the stream-walk kernels beyond the
functions listed below.  Those kernels hand data between lanes of a wave through LDS with a compiler-level ordering point
only (FE_WAVE_SYNC / WAVE_SYNC / SYNC: a wave's DS instructions execute in issue order); the ISA promises no such order
between DS and FLAT instructions, so an LDS access that compiles to FLAT (a generic pointer in an out-of-line function)
would make those hand-overs unsound.  Compiles the translation units to assembly with the product's flags (hipcc
cross-compiles without a GPU) and reads the listings.  Exit code 1 and a list on failure.
  python tools/check_lds_flat.py [--keep DIR]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hmp3_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", '-DHX_BUILD_ID="check"', "--cuda-device-only", "-S", "-w"]
ILP = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
NOLICM = ["-mllvm", "-disable-machine-licm"]
UNITS = [   # (name, source, flags) as hmp3_amd/build.sh compiles them
    ("front1", "hx_front.hip", ["-O3", "-fno-slp-vectorize", "-DHX_FRONT_PART=1"] + NOLICM),
    ("front2", "hx_front.hip", ["-O3", "-fno-slp-vectorize", "-DHX_FRONT_PART=2"] + ILP),
    ("pack", "hx_pack.hip", ["-O3"]),
    ("alloc", "hx_alloc.hip", ["-O2"] + ILP + NOLICM),
    ("alloc_slim", "hx_alloc_slim.hip", ["-O2"] + ILP + NOLICM),
    ("alloc_lsf", "hx_alloc_lsf.hip", ["-O2"] + ILP + NOLICM),
    ("alloc1", "hx_alloc1.hip", ["-O2"] + ILP + NOLICM),
    ("alloc1_lsf", "hx_alloc1_lsf.hip", ["-O2"] + ILP + NOLICM),
]
# Functions of the stream walk whose FLAT instructions address global or private memory (checked by reading them: the
# double-precision x^(4/3) table and the double-table counter, the psy model's outputs of the front end, packet outputs, a
# by-reference result on the stack), with the number of FLAT instructions each has today: one more in any of them fails the
# check and has to be read before the number is raised.  (A name is matched as a prefix of the demangled function name.)
ALLOWED = {"sweep_run_stored": 3, "void lucky_terms_big<2>": 2, "void lucky_terms_big<3>": 3, "compute_mask_short": 5, "hf_adjust_ch": 1,
           "bitallo_short": 2, "pack_side": 1, "pack_side_lsf": 1, "emit_packet": 2, "bitallo1": 22, "pack_sf_lsf": 2, "pack_sf_lsf_is": 2,
           "pow43_beyond": 0, "dblog": 0}


def allowed(dem, n):
    name = dem.split("(")[0]
    return name in ALLOWED and n <= ALLOWED[name]


def flat_ops(path):
    out = {}
    fn = None
    for line in open(path):
        m = re.match(r"^([_A-Za-z][\w.$]*):", line)
        if m and not m.group(1).startswith(".L"):
            fn = m.group(1)
        t = line.strip()
        if t.startswith("flat_load") or t.startswith("flat_store") or t.startswith("flat_atomic"):
            out[fn] = out.get(fn, 0) + 1
    return out


def main():
    if "--print-flags" in sys.argv:      # the flags this check compiles with, unit by unit (tests compare them with hmp3_amd/build.sh)
        for name, src, flags in UNITS:
            print(name, src, " ".join(flags))
        return
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    tmp = keep or tempfile.mkdtemp(prefix="hxflat.")
    os.makedirs(tmp, exist_ok=True)
    procs = []
    for name, src, flags in UNITS:
        out = os.path.join(tmp, name + ".s")
        procs.append((name, out, subprocess.Popen([HIPCC] + BASE + flags + [src, "-o", out], cwd=SRC, stderr=subprocess.PIPE)))
    bad = []
    for name, out, p in procs:
        err = p.communicate()[1].decode()
        if p.returncode != 0:
            print(err[-2000:], file=sys.stderr)
            raise SystemExit("check_lds_flat: %s did not compile" % name)
        ops = flat_ops(out)
        names = subprocess.run(["c++filt"], input="\n".join(ops), capture_output=True, text=True).stdout.split("\n")
        for mangled, dem in zip(ops, names):
            if name.startswith("alloc") and allowed(dem, ops[mangled]):
                continue
            bad.append("%s: %s has %d FLAT instruction(s)" % (name, dem, ops[mangled]))
    if bad:
        print("\n".join(bad))
        raise SystemExit(1)
    print("check_lds_flat: ok (%d translation units)" % len(UNITS))


if __name__ == "__main__":
    main()
