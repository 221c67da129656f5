"""Long continuous streams: python tools/soak.py [frames] [cases] [streams]  - the first `cases` of eight configurations (default
all) x `streams` streams (default 6) x `frames` frames (default 20000) through calls of 1 .. 64 frames on one batch object,
compared byte for byte with the oracle.  Exit code 1 on a difference.  (tests/test_gpu_runtime.py runs a slice: 6000 2 2)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmp3_amd import api, synth
from oracle import oracle as O

F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
CASES = [dict(bitrate=64), dict(), dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), dict(samprate=22050, bitrate=32),
         dict(bitrate=64, mode=3), dict(samprate=16000, mode=2, bitrate=16), dict(mode=0, bitrate=160, hf_flag=3), dict(mode=3, bitrate=160, hf_flag=3)]
CASES = CASES[:int(sys.argv[2])] if len(sys.argv) > 2 else CASES
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
bad = 0
rs = np.random.RandomState(7)
for kw in CASES:
    sr = kw.get("samprate", 44100)
    mono = kw.get("mode") == 3
    S, CH = NS, 250                                 # frames per synthesised piece
    b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=64)
    encs = [O.OracleEncoder(O.default_control(**kw)) for _ in range(S)]
    ok = True
    done = 0
    piece = 0
    while done < F and ok:
        n = min(CH, F - done)
        pcm = np.stack([synth.stream_pcm(5000 + 97 * piece + i, n, sr=sr, rho=[0.7, 0.0, 1.0, 0.3][i % 4], bursts=(i & 1) == 1) for i in range(S)])
        if mono:
            pcm = np.ascontiguousarray(pcm[:, :, 0])
        got = [b"" for _ in range(S)]
        f0 = 0
        while f0 < n:
            nf = int(min(n - f0, rs.randint(1, 65)))
            out = b.encode_host(np.ascontiguousarray(pcm[:, f0 * 1152:(f0 + nf) * 1152]))
            for s in range(S): got[s] += out[s]
            f0 += nf
        for s in range(S):
            want = b"".join(encs[s].encode_s16(pcm[s, f * 1152:(f + 1) * 1152]) for f in range(n))
            if got[s] != want:
                print("MISMATCH", kw, "stream", s, "frames", done, "..", done + n); ok = False; bad += 1; break
        done += n
        piece += 1
    print("soak %-70s %6d frames x %d streams  status %d  %s" % (kw, done, S, b.status(), "ok" if ok else "BAD"))
    if b.status() != 0: bad += 1
    b.close()
sys.exit(1 if bad else 0)
