"""Randomised pin of the oracle against the real reference (oracle/_ref; this container only): random controls x random
signals, byte for byte.  python tools/fuzz_oracle_vs_ref.py [--a1 | --hf] [n_cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmp3_amd import synth
from oracle import oracle as O
assert O.ref() is not None, "make -C oracle ref first"
A1 = "--a1" in sys.argv        # only configurations of the first-generation allocator: dual channel, or joint stereo at low bit rates
if A1:
    sys.argv.remove("--a1")
HF = "--hf" in sys.argv        # only -HF configurations at MPEG-1 rates and high bit rates (band 21 gets quantised; tools/fuzz_parity.py --hf)
if HF:
    sys.argv.remove("--hf")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
RATES = [16000, 22050, 24000, 32000, 44100, 48000]

bad = done = tried = ub = 0
while done < n_cases and tried < 30 * n_cases:
    tried += 1
    sr = int(rs.choice(RATES))
    kw = dict(samprate=sr, mode=int(rs.choice([0, 0, 1, 1, 1, 2, 3])))
    if rs.rand() < 0.5: kw["bitrate"] = int(rs.choice([8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160]))
    else: kw["vbr_mnr"] = int(rs.randint(0, 151))
    if A1:
        kw.pop("vbr_mnr", None)
        kw["mode"] = int(rs.choice([1, 2, 2]))
        kw["bitrate"] = int(rs.choice([8, 16, 24, 32, 40] if (kw["mode"] == 1 or sr < 32000) else [16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160]))
    if rs.rand() < 0.3: kw["hf_flag"] = int(rs.choice([1, 3]))
    if HF:
        kw["samprate"] = sr = int(rs.choice([32000, 44100, 48000]))
        kw["hf_flag"] = int(rs.choice([1, 3, 3]))
        kw.pop("vbr_mnr", None); kw.pop("bitrate", None)
        if rs.rand() < 0.6: kw["bitrate"] = int(rs.choice([96, 112, 128, 160]))
        else: kw["vbr_mnr"] = int(rs.randint(80, 151))
    if rs.rand() < 0.3: kw["freq_limit"] = int(rs.choice([8000, 12000, 16000, 19000, 21000]))
    if rs.rand() < 0.3: kw["short_block_threshold"] = int(rs.choice([300, 700, 2000, 99999]))
    if rs.rand() < 0.15: kw["filter_select"] = 1
    if rs.rand() < 0.15: kw["nsbstereo"] = int(rs.choice([4, 8, 12, 16]))
    if rs.rand() < 0.15: kw["nsb_limit"] = int(rs.choice([2, 3, 4, 6, 8, 12, 16, 20, 24, 28, 31]))
    # the remaining knobs of E_CONTROL (pub/encapp.h:42-72): VBR bitrate cap and MNR offset, tuning switches, header bits,
    # arbitrary cut-offs and block-switching thresholds
    if rs.rand() < 0.2: kw["vbr_br_limit"] = int(rs.choice([16, 32, 48, 64, 96, 128, 160]))
    if rs.rand() < 0.2: kw["vbr_delta_mnr"] = int(rs.randint(-60, 71))
    if rs.rand() < 0.2: kw["test1"] = int(rs.randint(0, 16))
    if rs.rand() < 0.15: kw["quick"] = int(rs.choice([0, 1]))
    if rs.rand() < 0.1: kw["cr_bit"] = int(rs.randint(0, 2)); kw["original"] = int(rs.randint(0, 2))
    if rs.rand() < 0.15: kw["freq_limit"] = int(rs.randint(500, 24001))
    if rs.rand() < 0.15: kw["short_block_threshold"] = int(rs.randint(0, 3001))
    ok_o = O.OracleEncoder(O.default_control(**kw)).ok()
    r = O.RefEncoder(O.default_control(**kw))
    ok_r = r.bytes_in > 0
    if ok_o != ok_r:
        print("INIT MISMATCH", kw, "oracle", ok_o, "reference", ok_r); bad += 1; done += 1
        continue
    if not ok_r:
        continue
    F = int(rs.choice([10, 20, 40]))
    sig = dict(seed=int(rs.randint(0, 1 << 20)), rho=float(rs.choice([0.0, 0.3, 0.7, 1.0])), bursts=bool(rs.rand() < 0.6))
    pcm = synth.stream_pcm(sig["seed"], F, sr=sr, rho=sig["rho"], bursts=sig["bursts"])
    sig["amp"] = float(rs.choice([1.0, 1.0, 0.25, 0.02]))
    pcm = (pcm.astype(np.float64) * sig["amp"]).astype(np.int16)
    n = F * 1152
    kind = int(rs.randint(0, 12))
    sig["kind"] = kind
    if kind == 0: pcm[:] = 0
    elif kind == 1: pcm[:] = np.where((np.arange(n) // int(rs.randint(2, 200))) % 2 == 0, 32767, -32768).astype(np.int16)[:, None]
    elif kind == 2: pcm[:] = rs.randint(-32768, 32768, size=(n, 2)).astype(np.int16)
    elif kind == 3: pcm[:] = 0; pcm[::int(rs.randint(50, 3000))] = 32767
    elif kind == 4: pcm[:] = (32000 * np.sin(2 * np.pi * float(rs.uniform(20, sr / 2.1)) * np.arange(n) / sr)).astype(np.int16)[:, None]
    elif kind == 5: pcm[:] = np.clip(pcm.astype(np.int32) + int(rs.randint(-20000, 20000)), -32768, 32767).astype(np.int16)
    elif kind == 6: pcm[:, 1] = -pcm[:, 0]
    if kw["mode"] == 3:
        pcm = np.ascontiguousarray(pcm[:, 0])
    if rs.rand() < 0.25:        # the fp32 entry point (L3_audio_encode), fractional sample values
        pcmf = (pcm.astype(np.float32) + rs.uniform(-0.5, 0.5, size=pcm.shape).astype(np.float32)).astype(np.float32)
        rf, of = O.RefEncoder(O.default_control(**kw), s16=False), O.OracleEncoder(O.default_control(**kw))
        a = b"".join(rf.encode_f32(pcmf[f * 1152:(f + 1) * 1152]) for f in range(F))
        b = b"".join(of.encode_f32(pcmf[f * 1152:(f + 1) * 1152]) for f in range(F))
        sig["f32"] = True
    else:
        a = O.encode_stream(r, pcm)
        b = O.encode_stream(O.OracleEncoder(O.default_control(**kw)), pcm)
    if a != b and O.ref_zero() is not None:
        # a difference that goes away when the reference's uninitialised locals are zero is the reference reading stack
        # residue (spdsmr.c: stab[npart] for an odd partition count), not a difference of the algorithms.  Checked in a
        # child process: the zero-initialised build (clang) crashes on some intensity-stereo configurations.
        pid = os.fork()
        if pid == 0:
            rz = O.RefEncoder(O.default_control(**kw), s16=not sig.get("f32"), zero_locals=True)
            if sig.get("f32"):
                az = b"".join(rz.encode_f32(pcmf[f * 1152:(f + 1) * 1152]) for f in range(F))
            else:
                az = O.encode_stream(rz, pcm)
            os._exit(0 if az == b else 1)
        if os.waitpid(pid, 0)[1] == 0:
            print("UNINITIALISED-LOCAL CASE (oracle == reference with zeroed locals != reference)", kw, "F", F, sig)
            ub += 1
            a = b
    if a != b:
        import hashlib
        print("MISMATCH", kw, len(a), len(b), "F", F, sig, "md5 reference", hashlib.md5(a).hexdigest()[:8], "oracle", hashlib.md5(b).hexdigest()[:8]); bad += 1
        if os.environ.get("FUZZ_DUMP"):     # keep the case for a closer look: <dir>/caseN.npy + .json
            import json
            np.save(os.path.join(os.environ["FUZZ_DUMP"], "case%d.npy" % bad), pcm)
            json.dump(kw, open(os.path.join(os.environ["FUZZ_DUMP"], "case%d.json" % bad), "w"))
    done += 1
print("oracle vs reference fuzz: %d cases, %d bad%s" % (done, bad, (", %d where the reference read an uninitialised local" % ub) if ub else ""))
sys.exit(1 if bad else 0)
