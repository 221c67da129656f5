"""Randomised whole-file sweep of the command line: random WAV (rate incl. the ones that need the rate converter, mono /
stereo, 8 / 16 / 24 / 32-bit or float samples, RIFF / RIFX / RF64 / Wave64 / extensible header) x random flags,
hmp3_amd/hmp3amd (GPU) against the real reference's CLI (oracle/_ref/hmp3, the prebuilt binary), file for file.  A third
of the cases reach hmp3amd through a pipe (the streaming route).  --batch: several files per case through `hmp3amd -batch`
(same channel count and MPEG version, any mix of rates and sample formats), each against the reference run on it alone.
python tools/fuzz_cli.py [--batch] [n_cases] [seed]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np                      # noqa: E402
import make_golden_cli as M             # noqa: E402
from hmp3_amd import synth              # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "hmp3")
CLI = os.path.join(ROOT, "hmp3_amd", "hmp3amd")
assert os.path.exists(REF), "oracle/_ref/hmp3 missing (make -C oracle ref)"
BATCH = "--batch" in sys.argv
if BATCH:
    sys.argv.remove("--batch")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
bad = done = 0
with tempfile.TemporaryDirectory() as d:
    while BATCH and done < n_cases:
        lsf = bool(rs.rand() < 0.35)
        mono = bool(rs.rand() < 0.3)
        K = int(rs.randint(2, 7))
        flags = []
        if rs.rand() < 0.5: flags.append("-B%d" % int(rs.choice([24, 32, 40] if lsf else [48, 64, 96, 128])))
        else: flags.append("-V%d" % int(rs.randint(10, 141)))
        if rs.rand() < 0.3: flags.append("-HF2")
        if rs.rand() < 0.3: flags.append("-SBT%d" % int(rs.choice([300, 700, 99999])))
        if rs.rand() < 0.3: flags.append("-X%d" % int(rs.choice([0, 1, 2])))
        files, args = [], []
        for k in range(K):
            sr = int(rs.choice([16000, 22050, 24000] if lsf else [32000, 44100, 48000]))
            fmt = [False, True, 8, 24, 32][int(rs.randint(0, 5))]
            if mono and fmt in (24, 32):
                fmt = False
            nsamp = int(rs.randint(3000, 40000))
            pcm = synth.stream_pcm(int(rs.randint(0, 1 << 20)), (nsamp + 1151) // 1152, sr=sr, rho=float(rs.choice([0.0, 0.5, 1.0])), bursts=bool(rs.rand() < 0.5))[:nsamp]
            if mono:
                pcm = pcm[:, 0].copy()
            w, o1, o2 = os.path.join(d, "in%d.wav" % k), os.path.join(d, "ref%d.mp3" % k), os.path.join(d, "gpu%d.mp3" % k)
            M.write_wav(w, pcm, sr, fmt, None)
            subprocess.run([REF, w, o1] + flags, capture_output=True)
            files.append((w, o1, o2, sr, fmt, nsamp))
            args += [w, o2]
        r2 = subprocess.run([CLI, "-batch"] + args + flags, capture_output=True)
        done += 1
        for w, o1, o2, sr, fmt, nsamp in files:
            same = os.path.exists(o1) and os.path.exists(o2) and open(o1, "rb").read() == open(o2, "rb").read()
            if not same:
                bad += 1
                print("MISMATCH batch of %d (lsf %s mono %s) flags %s: file sr %d fmt %s nsamp %d | rc %d %s" % (K, lsf, mono, " ".join(flags), sr, fmt, nsamp, r2.returncode, r2.stderr.decode()[-200:]))
                break
        for w, o1, o2, *_ in files:
            for f in (o1, o2):
                if os.path.exists(f):
                    os.remove(f)
    wav, a, b = os.path.join(d, "in.wav"), os.path.join(d, "ref.mp3"), os.path.join(d, "gpu.mp3")
    it = -1
    while not BATCH and done < n_cases:
        it += 1         # (the generator's iteration: FUZZ_CLI_ONLY names a case by it)
        sr = int(rs.choice([8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000]))
        mono = rs.rand() < 0.3
        fmt = [False, True, 8, 24, 32][int(rs.randint(0, 5))]          # 16-bit, float, 8 / 24 / 32-bit integer
        container = None
        if fmt in (False, 24) and rs.rand() < 0.4:
            container = str(rs.choice(["rifx", "rf64", "w64", "ext"]))
        if mono and fmt in (24, 32):
            fmt = False                 # (the WAV writer's mono branch knows 8 / 16-bit and float)
        if mono:
            container = None
        nsamp = int(rs.randint(3000, 50000))
        nfr = (nsamp + 1151) // 1152
        pcm = synth.stream_pcm(int(rs.randint(0, 1 << 20)), nfr, sr=sr, rho=float(rs.choice([0.0, 0.5, 1.0])), bursts=bool(rs.rand() < 0.5))[:nsamp]
        if mono:
            pcm = pcm[:, 0].copy()
        flags = []
        if rs.rand() < 0.5: flags.append("-B%d" % int(rs.choice([8, 16, 24, 32, 48, 64, 96, 128, 160])))
        else: flags.append("-V%d" % int(rs.randint(0, 151)))
        if rs.rand() < 0.5: flags.append("-M%d" % int(rs.choice([0, 1, 2, 3])))
        if rs.rand() < 0.3: flags.append("-HF%d" % int(rs.choice([0, 2])))
        if rs.rand() < 0.3: flags.append("-F%d" % int(rs.choice([6000, 12000, 16000, 19000, 22000])))
        if rs.rand() < 0.3: flags.append("-SBT%d" % int(rs.choice([0, 300, 700, 2000, 99999])))
        if rs.rand() < 0.2: flags.append("-S1")
        if rs.rand() < 0.3: flags.append("-X%d" % int(rs.choice([0, 1, 2])))
        if rs.rand() < 0.2: flags.append("-A%d" % int(rs.choice([0, 1, 2, 16000, 22050, 24000, 32000, 44100, 48000])))
        if rs.rand() < 0.15: flags.append("-N%d" % int(rs.choice([4, 8, 12, 16])))
        if rs.rand() < 0.15: flags.append("-C%d" % int(rs.choice([0, 1])))
        if rs.rand() < 0.15: flags.append("-T%d" % int(rs.randint(-40, 51)))
        if rs.rand() < 0.15: flags.append("-L%d" % int(rs.choice([64, 96, 128, 160])))
        M.write_wav(wav, pcm, sr, fmt, container)
        for f in (a, b):
            if os.path.exists(f):
                os.remove(f)
        r1 = subprocess.run([REF, wav, a] + flags, capture_output=True)
        # FUZZ_CLI_ONLY=<iteration>,<repeats>: that case alone, the GPU side run <repeats> times (hunting a timing-dependent difference)
        only = os.environ.get("FUZZ_CLI_ONLY")
        if only:
            k_only, reps = (int(x) for x in only.split(","))
            piped = rs.rand() < 0.33
            okr = os.path.exists(a) and os.path.getsize(a) > 0
            if it != k_only:
                done += 1 if okr else 0
                continue
            if not okr:
                print("iteration %d: the reference rejects this combination (flags %s) - not a case" % (k_only, " ".join(flags)))
                sys.exit(0)
            want = open(a, "rb").read()
            nbad = 0
            for r in range(reps):
                if os.path.exists(b):
                    os.remove(b)
                subprocess.run([CLI, wav, b] + flags, capture_output=True)
                got = open(b, "rb").read() if os.path.exists(b) else b""
                if got != want:
                    nbad += 1
                    if nbad <= 3:
                        n = min(len(got), len(want))
                        first = next((i for i in range(n) if got[i] != want[i]), n)
                        print("  repeat %d: %d bytes against %d, first difference at byte %d" % (r, len(got), len(want), first))
            print("case %d (sr %d mono %s fmt %s flags %s): %d of %d runs differ" % (k_only, sr, mono, fmt, " ".join(flags), nbad, reps))
            sys.exit(1 if nbad else 0)
        if rs.rand() < 0.33:
            with open(wav, "rb") as fh:
                r2 = subprocess.run([CLI, "-", b] + flags, stdin=fh, capture_output=True)
            flags = flags + ["(stdin)"]
        else:
            r2 = subprocess.run([CLI, wav, b] + flags, capture_output=True)
        ok1, ok2 = os.path.exists(a) and os.path.getsize(a) > 0, os.path.exists(b) and os.path.getsize(b) > 0
        if not ok1 and not ok2:
            continue                    # both reject the combination: not a case
        done += 1
        same = ok1 and ok2 and open(a, "rb").read() == open(b, "rb").read()
        if not same:
            bad += 1
            print("MISMATCH sr %d mono %s fmt %s container %s nsamp %d flags %s | reference %s bytes rc %d, gpu %s bytes rc %d" % (
                sr, mono, fmt, container, nsamp, " ".join(flags), os.path.getsize(a) if ok1 else None, r1.returncode,
                os.path.getsize(b) if ok2 else None, r2.returncode))
            print("   ", r2.stderr.decode()[-200:].replace("\\n", " | "))
print("cli fuzz: %d cases, %d bad" % (done, bad))
sys.exit(1 if bad else 0)
