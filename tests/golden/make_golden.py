"""Generate tests/golden/* from the REAL reference (oracle/_ref/libhmp3ref.so).
Runs only in the build container (needs `make -C oracle ref`).  The fixtures are data:
inputs are regenerated from seeds (hmp3_amd/synth.py, numpy PCG64), outputs are what the
reference produced.  Usage: python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from hmp3_amd import synth              # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name -> (E_CONTROL overrides, sample rate, frames, rho, bursts)
STREAM_CASES = {
    "cbr128_long": (dict(bitrate=64, short_block_threshold=99999), 44100, 24, 0.7, False),
    "cbr128_lr_long": (dict(bitrate=64, mode=0, short_block_threshold=99999), 44100, 16, 0.3, False),
    "cbr128_rho0_long": (dict(bitrate=64, short_block_threshold=99999), 44100, 16, 0.0, False),
    "cbr192_long": (dict(bitrate=96, short_block_threshold=99999), 44100, 16, 0.7, False),
    "cbr128_32k_long": (dict(bitrate=64, samprate=32000, short_block_threshold=99999), 32000, 16, 0.7, False),
    "cbr128_48k_long": (dict(bitrate=64, samprate=48000, short_block_threshold=99999), 48000, 16, 0.7, False),
    "vbr50_long": (dict(short_block_threshold=99999), 44100, 24, 0.7, False),
    "vbr100_hf2_48k_long": (dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000, short_block_threshold=99999), 48000, 16, 0.7, False),
    "cbr128_default_bursts": (dict(bitrate=64), 44100, 40, 0.7, True),
    "vbr50_default_bursts": (dict(), 44100, 40, 0.7, True),
}

# streams whose signal is more than a seed: name -> (E_CONTROL overrides, sample rate, frames, seed, rho, bursts, amplitude,
# right = -left[, DC offset added after scaling, reference build with zeroed locals])
# a1_dual_16k_antiphase: dual channel at 2 x 8 kbps, the first-generation allocator; with the double-precision log10 the
# reference does NOT use there (bitallo1.cpp is C++: log10f) an encoder codes one more line in the second frame
# stab_odd_npart_32k: the reference's spd_smrLongEcho (spdsmr.c) reads stab[npart], a local it never wrote, when the psy
# model's partition count is odd; after short-block frames the slot holds stack residue and the reference's own output
# depends on it (1 case in 90 000 of tools/fuzz_oracle_vs_ref.py).  The oracle and the GPU define the slot as 0; the golden
# bytes come from the reference compiled with -ftrivial-auto-var-init=zero (make -C oracle ref_zero), which is the same
# code with that read defined.
EXTRA_CASES = {
    "a1_dual_16k_antiphase": (dict(samprate=16000, mode=2, bitrate=8), 16000, 40, 181220, 1.0, True, 0.25, True),
    "stab_odd_npart_32k": (dict(samprate=32000, mode=0, vbr_mnr=66, hf_flag=3, freq_limit=5407, short_block_threshold=700,
                                vbr_br_limit=32, vbr_delta_mnr=58), 32000, 40, 411809, 0.0, True, 0.02, False, -19984, True),
}


def extra_case_pcm(name):
    kw, sr, nfr, seed, rho, bursts, amp, anti = EXTRA_CASES[name][:8]
    pcm = synth.stream_pcm(seed, nfr, sr=sr, rho=rho, bursts=bursts)
    pcm = (pcm.astype(np.float64) * amp).astype(np.int16)
    if anti:
        pcm[:, 1] = -pcm[:, 0]
    if len(EXTRA_CASES[name]) > 8:
        pcm = np.clip(pcm.astype(np.int32) + EXTRA_CASES[name][8], -32768, 32767).astype(np.int16)
    return pcm


def main():
    r = O.ref()
    if r is None:
        raise SystemExit("oracle/_ref/libhmp3ref.so missing: run `make -C oracle ref` first")
    os.makedirs(GOLD, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(20260210))

    # ---- known answers for the millibel log / exp and x^(3/4) primitives ----
    xs = np.concatenate([
        np.array([0.0, 1.0, 2.0, 10.0, 1000.0, 1e-12, 1e-30, 7.0e4, 3.4e38, 1.17e-38], dtype=np.float32),
        np.exp(rng.uniform(-60, 60, 4000)).astype(np.float32)])
    r.ref_mbLogC.restype = C.c_int
    ml = np.array([r.ref_mbLogC(C.c_float(float(x))) for x in xs], dtype=np.int32)
    mi = np.arange(-33000, 33001, 7, dtype=np.int32)
    me = np.array([r.ref_mbExp(int(i)) for i in mi], dtype=np.float32)
    px = np.abs(np.concatenate([rng.standard_normal(4000) * 3000, np.array([0.0, 1.0, 32768.0])])).astype(np.float32)
    py = np.zeros_like(px)
    r.ref_fpow34.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    r.ref_fpow34(px.ctypes.data, py.ctypes.data, len(px))
    np.savez_compressed(os.path.join(GOLD, "kat_math.npz"), mblog_x=xs, mblog_y=ml, mbexp_x=mi, mbexp_y=me, pow34_x=px, pow34_y=py)

    # ---- stage vectors: polyphase, frequency inversion, hybrid MDCT, alias, attack detector ----
    r.ref_sbt_L3.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_tables_init.argtypes = [C.c_int, C.c_int]
    r.ref_hybridLong.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    r.ref_antialias.argtypes = [C.c_void_p, C.c_int]
    r.ref_FreqInvert.argtypes = [C.c_void_p, C.c_int]
    r.ref_attack.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    r.ref_tables_init(0, 414)
    vb, so = [], []
    for k in range(4):
        v = (rng.standard_normal(576 + 512) * (3000.0 if k else 1.0)).astype(np.float32)
        if k == 3:
            v[:] = 0
            v[100] = 32767.0
        s = np.zeros(576, dtype=np.float32)
        r.ref_sbt_L3(v.ctypes.data, s.ctypes.data)
        vb.append(v); so.append(s)
    hin1, hin2, hout, hbt = [], [], [], []
    for bt in (0, 1, 3):
        x1 = (rng.standard_normal(576) * 2000).astype(np.float32)
        x2 = (rng.standard_normal(576) * 2000).astype(np.float32)
        y = np.zeros(576, dtype=np.float32)
        x2i = x2.copy()
        r.ref_FreqInvert(x2i.ctypes.data, 23)
        r.ref_hybridLong(x1.ctypes.data, x2i.ctypes.data, y.ctypes.data, bt, 23, 0)
        r.ref_antialias(y.ctypes.data, 23)
        hin1.append(x1); hin2.append(x2); hout.append(y); hbt.append(bt)
    eng = np.full(32, 9000, dtype=np.int32)
    at_in, at_out, at_eng = [], [], []
    prev = 0
    for k in range(6):
        smp = (rng.standard_normal(576) * (100.0 if k != 3 else 8000.0)).astype(np.float32)
        m = r.ref_attack(smp.ctypes.data, eng.ctypes.data, prev)
        at_in.append(smp); at_out.append(m); at_eng.append(eng.copy())
        prev = 1 if m > 700 else 0
    np.savez_compressed(os.path.join(GOLD, "stage_frontend.npz"), sbt_in=np.array(vb), sbt_out=np.array(so),
                        hy_x1=np.array(hin1), hy_x2=np.array(hin2), hy_out=np.array(hout), hy_bt=np.array(hbt),
                        at_in=np.array(at_in), at_out=np.array(at_out, dtype=np.int32), at_eng=np.array(at_eng))

    # ---- whole streams: reference bitstream + resolved parameters + carried state per frame ----
    meta = {}
    for name, (kw, sr, nfr, rho, bursts) in STREAM_CASES.items():
        ec = O.default_control(**kw)
        pcm = synth.stream_pcm(7, nfr, sr=sr, rho=rho, bursts=bursts)
        enc = O.RefEncoder(ec)
        out, sizes, state = [], [], []
        for f in range(nfr + 2):
            frame = pcm[f * 1152:(f + 1) * 1152] if f < nfr else np.zeros((1152, 2), dtype=np.int16)
            b = enc.encode_s16(frame)
            out.append(b); sizes.append(len(b))
            d = enc.dump()
            state.append(dict(MNR=d.MNR, byte_pool=d.byte_pool, ms_mem=d.ms_correlation_memory, call_count=d.call_count,
                              PoolFraction=d.PoolFraction, NTadjust=list(d.NTadjust),
                              block_type=[d.gr[5], d.gr[2 * 26 + 5]],
                              ix_sum=int(np.abs(np.array(d.ix)).sum()), part23=[d.gr[0], d.gr[26], d.gr[52], d.gr[78]]))
        d = enc.dump()
        with open(os.path.join(GOLD, name + ".mp3frames"), "wb") as fh:
            fh.write(b"".join(out))
        meta[name] = dict(control=kw, samprate=sr, frames=nfr, rho=rho, bursts=bursts, stream_seed=7, out_sizes=sizes,
                          resolved=dict(nsb_limit=d.nsb_limit, nsb_limitMS=list(d.nsb_limitMS), band_limit=d.band_limit,
                                        AveTargetBits=d.AveTargetBits, main_framebytes=d.main_framebytes,
                                        initialMNR=d.initialMNR, nsf=list(d.nsf)),
                          state=state)
        print("%-26s %6d bytes, %d frames" % (name, sum(sizes), nfr))
    for name, case in EXTRA_CASES.items():
        kw, nfr = case[0], case[2]
        zero_locals = len(case) > 9 and case[9]
        if zero_locals and O.ref_zero() is None:      # (no clang in this image: the committed file stays as it is)
            print("%-26s skipped: oracle/_ref/libhmp3ref_zero.so missing (make -C oracle ref_zero)" % name)
            continue
        data = O.encode_stream(O.RefEncoder(O.default_control(**kw), zero_locals=zero_locals), extra_case_pcm(name))
        with open(os.path.join(GOLD, name + ".mp3frames"), "wb") as fh:
            fh.write(data)
        print("%-26s %6d bytes, %d frames" % (name, len(data), nfr))
    with open(os.path.join(GOLD, "streams.json"), "w") as fh:
        json.dump(meta, fh)
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
