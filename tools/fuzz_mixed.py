"""Randomised parity sweep of mixed-class batches: every stream of a batch its own control (rate, bitrate / VBR quality, HF
mode, cut-off, block switching), as far as one batch may mix them (same MPEG version, channel count and allocator
generation), against the CPU oracle.  python tools/fuzz_mixed.py [n_batches] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmp3_amd import api, synth
from oracle import oracle as O
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = done = 0
while done < n_batches:
    lsf = bool(rs.rand() < 0.3)
    mono = bool(rs.rand() < 0.25)
    rates = [16000, 22050, 24000] if lsf else [32000, 44100, 48000]
    S, F = int(rs.choice([16, 32, 48])), int(rs.choice([6, 10, 16]))
    kws = []
    while len(kws) < S:
        kw = dict(samprate=int(rs.choice(rates)), mode=3 if mono else int(rs.choice([0, 1])))
        if rs.rand() < 0.5: kw["bitrate"] = int(rs.choice([32, 40, 48, 56, 64, 80, 96, 128, 160] if not lsf else [24, 32, 40, 48, 56, 64, 80]))
        else: kw["vbr_mnr"] = int(rs.randint(0, 151))
        if rs.rand() < 0.3: kw["hf_flag"] = int(rs.choice([1, 3]))
        if rs.rand() < 0.3: kw["freq_limit"] = int(rs.choice([8000, 12000, 16000, 19000]))
        if rs.rand() < 0.3: kw["short_block_threshold"] = int(rs.choice([300, 700, 99999]))
        if rs.rand() < 0.1: kw["filter_select"] = 1
        if rs.rand() < 0.1: kw["nsb_limit"] = int(rs.choice([2, 3, 4, 6, 8, 12, 16, 24, 31]))
        if rs.rand() < 0.15: kw["vbr_delta_mnr"] = int(rs.randint(-60, 71))
        if rs.rand() < 0.15: kw["test1"] = int(rs.randint(0, 16))
        if rs.rand() < 0.1: kw["quick"] = int(rs.choice([0, 1]))
        if rs.rand() < 0.1: kw["freq_limit"] = int(rs.randint(500, 24001))
        if rs.rand() < 0.1: kw["short_block_threshold"] = int(rs.randint(0, 3001))
        e = O.OracleEncoder(O.default_control(**kw))
        if not e.ok(): continue
        kws.append(kw)
    pcm = np.stack([(synth.stream_pcm(int(rs.randint(0, 1 << 20)), F, sr=kws[i]["samprate"], rho=float(rs.choice([0.0, 0.3, 0.7, 1.0])), bursts=bool(rs.rand() < 0.6)).astype(np.float64)
                     * float(rs.choice([1.0, 1.0, 0.3, 0.03]))).astype(np.int16) for i in range(S)])
    if mono: pcm = np.ascontiguousarray(pcm[:, :, 0])
    try:
        b = api.Batch([api.default_control(**k) for k in kws], nstreams=S, max_frames=F)
    except Exception as e:      # (a mix the library refuses, e.g. both allocator generations: not a parity failure)
        continue
    got = [b"" for _ in range(S)]
    f0 = 0
    while f0 < F:
        nf = int(rs.randint(1, F - f0 + 1))
        out = b.encode_host(np.ascontiguousarray(pcm[:, f0 * 1152:(f0 + nf) * 1152]))
        for s in range(S): got[s] += out[s]
        f0 += nf
    st = b.status(); b.close()
    for s in range(S):
        enc = O.OracleEncoder(O.default_control(**kws[s]))
        want = b"".join(enc.encode_s16(pcm[s, f * 1152:(f + 1) * 1152]) for f in range(F))
        if got[s] != want or st != 0:
            print("MISMATCH batch", done, "stream", s, kws[s], "status", st, len(got[s]), len(want)); bad += 1; break
    done += 1
print("mixed-class fuzz: %d batches, %d bad" % (done, bad))
sys.exit(1 if bad else 0)
