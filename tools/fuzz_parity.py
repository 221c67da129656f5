"""Randomised parity sweep: random encoder controls x random synthetic streams, GPU batch (plain and submit path) against the
CPU oracle.  python tools/fuzz_parity.py [--a1 | --hf] [--submit] [n_cases] [seed].  Prints every mismatch; exit code 1 if there was one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmp3_amd import api, synth
from oracle import oracle as O

A1 = "--a1" in sys.argv                  # only configurations of the first-generation allocator (dual channel, low-rate joint stereo)
if A1:
    sys.argv.remove("--a1")
HF = "--hf" in sys.argv                  # only -HF configurations at MPEG-1 rates and high bit rates (band 21 gets quantised)
if HF:
    sys.argv.remove("--hf")
SUBMIT = "--submit" in sys.argv          # random-sized calls through hx_batch_submit_s16_device (overlapped) instead of host calls
if SUBMIT:
    sys.argv.remove("--submit")
    import torch
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
RATES = [16000, 22050, 24000, 32000, 44100, 48000]
bad = 0
done = 0
tried = 0
while done < n_cases and tried < 20 * n_cases:
    tried += 1
    sr = int(rs.choice(RATES))
    kw = dict(samprate=sr)
    mode = int(rs.choice([0, 0, 1, 1, 1, 2, 3]))
    kw["mode"] = mode
    if rs.rand() < 0.5:
        kw["bitrate"] = int(rs.choice([8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160]))
    else:
        kw["vbr_mnr"] = int(rs.randint(0, 151))
    if A1:
        kw.pop("vbr_mnr", None)
        kw["mode"] = mode = int(rs.choice([1, 2, 2]))
        kw["bitrate"] = int(rs.choice([8, 16, 24, 32, 40] if (mode == 1 or sr < 32000) else [16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160]))
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
    ec = O.default_control(**kw)
    if not O.OracleEncoder(ec).ok():
        continue
    nch = 1 if mode == 3 else 2
    S, F = 6, int(rs.choice([7, 12, 20, 20, 20, 120]))     # (now and then a long run: reservoir wrap, padding cycle, VBR pool limiter)
    seeds = rs.randint(0, 1 << 20, size=S)
    rhos = rs.choice([0.0, 0.3, 0.7, 1.0], size=S)
    amp = rs.choice([1.0, 1.0, 0.25, 0.02], size=S)
    pcm = np.stack([synth.stream_pcm(int(seeds[i]), F, sr=sr, rho=float(rhos[i]), bursts=bool(rs.rand() < 0.6)) for i in range(S)])
    pcm = (pcm.astype(np.float64) * amp[:, None, None]).astype(np.int16)
    # a few extreme signals: silence, full-scale square wave, full-scale noise, impulses, one sine, DC offset
    n = F * 1152
    kind = int(rs.randint(0, 12))
    if kind == 0: pcm[0] = 0
    elif kind == 1: pcm[0] = np.where((np.arange(n) // int(rs.randint(2, 200))) % 2 == 0, 32767, -32768).astype(np.int16)[:, None]
    elif kind == 2: pcm[0] = rs.randint(-32768, 32768, size=(n, 2)).astype(np.int16)
    elif kind == 3: pcm[0] = 0; pcm[0, ::int(rs.randint(50, 3000))] = 32767
    elif kind == 4: pcm[0] = (32000 * np.sin(2 * np.pi * float(rs.uniform(20, sr / 2.1)) * np.arange(n) / sr)).astype(np.int16)[:, None]
    elif kind == 5: pcm[0] = np.clip(pcm[0].astype(np.int32) + int(rs.randint(-20000, 20000)), -32768, 32767).astype(np.int16)
    elif kind == 6: pcm[0, :, 1] = -pcm[0, :, 0]        # anti-phase channels
    if nch == 1:
        pcm = np.ascontiguousarray(pcm[:, :, 0])
    # a quarter of the cases through the fp32 entry points, with fractional sample values (host-call mode)
    as_f32 = (not SUBMIT) and rs.rand() < 0.25
    if as_f32:
        pcm = (pcm.astype(np.float32) + rs.uniform(-0.5, 0.5, size=pcm.shape).astype(np.float32)).astype(np.float32)
    try:
        b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    except Exception as e:
        print("CREATE FAILED", kw, e); bad += 1; done += 1; continue
    b.close()
    # the same frames in random-sized calls through one batch (state carried from call to call)
    b = api.Batch(api.default_control(**kw), nstreams=S, max_frames=F)
    got = [b"" for _ in range(S)]
    f0 = 0
    if SUBMIT:
        dev = torch.device("cuda:0"); stq = torch.cuda.current_stream().cuda_stream
        keep = []
        while f0 < F:
            nf = int(rs.randint(1, F - f0 + 1))
            d_pcm = torch.from_numpy(np.ascontiguousarray(pcm[:, f0 * 1152:(f0 + nf) * 1152])).to(dev)
            stride = b.out_stride(nf)
            d_out = torch.zeros((S, stride), dtype=torch.uint8, device=dev); d_nb = torch.zeros((S,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize() if not keep else None
            b.submit_device(d_pcm.data_ptr(), nf, d_out.data_ptr(), stride, d_nb.data_ptr(), stq)
            keep.append((d_pcm, d_out, d_nb))
            f0 += nf
        b.wait(stq); torch.cuda.synchronize()
        for _, d_out, d_nb in keep:
            o, nb = d_out.cpu().numpy(), d_nb.cpu().numpy()
            for s in range(S): got[s] += o[s, :nb[s]].tobytes()
    else:
        while f0 < F:
            nf = int(rs.randint(1, F - f0 + 1))
            out = b.encode_host(np.ascontiguousarray(pcm[:, f0 * 1152:(f0 + nf) * 1152]))
            for s in range(S): got[s] += out[s]
            f0 += nf
    st = b.status()
    b.close()
    for s in range(S):
        enc = O.OracleEncoder(O.default_control(**kw))
        want = b"".join((enc.encode_f32 if as_f32 else enc.encode_s16)(pcm[s, f * 1152:(f + 1) * 1152]) for f in range(F))
        if got[s] != want or st != 0:
            nd = next((k for k in range(min(len(got[s]), len(want))) if got[s][k] != want[k]), min(len(got[s]), len(want)))
            print("MISMATCH case", done, kw, "stream", s, "seed", int(seeds[s]), "rho", float(rhos[s]), "amp", float(amp[s]), "status", st, len(got[s]), len(want), "f32" if as_f32 else "s16", "F", F, "kind", kind, "first differing byte", nd, flush=True)
            bad += 1
            break
    done += 1
print("fuzz: %d cases, %d bad" % (done, bad))
sys.exit(1 if bad else 0)
