"""Parity of the HIP path against the oracle on a real MI355X, through the C ABI.
Bar: BIT-EXACT everywhere (the kernels evaluate every float expression in the reference's order,
so even the float stages are compared with ==, which is stricter than the tolerance BASELINE.json
allows: 0 ulp).  Run with:  python -m pytest tests -m gpu"""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O
from hmp3_amd import synth
from conftest import skip_unless_host_libm_is_the_restated_one

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLD, "streams.json")))
RHOS = [0.7, 0.0, 1.0, 0.3]

CONFIGS = {
    "cbr128": dict(bitrate=64, short_block_threshold=99999),
    "cbr128_lr": dict(bitrate=64, mode=0, short_block_threshold=99999),
    "cbr192": dict(bitrate=96, short_block_threshold=99999),
    "cbr320": dict(bitrate=160, short_block_threshold=99999),
    "vbr50": dict(short_block_threshold=99999),
    "vbr100_hf2_48k": dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000, short_block_threshold=99999),
    "cbr128_32k": dict(bitrate=64, samprate=32000, short_block_threshold=99999),
    "cbr128_48k": dict(bitrate=64, samprate=48000, short_block_threshold=99999),
    "cbr128_dcfilter": dict(bitrate=64, filter_select=1, short_block_threshold=99999),    # -S1 (filter2.c:116-144)
    # block switching enabled (CLI default threshold 700); the signal carries noise bursts
    "cbr128_sw": dict(bitrate=64),
    "cbr128_lr_sw": dict(bitrate=64, mode=0),
    "vbr50_sw": dict(),
    "vbr100_hf2_48k_sw": dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000),
    "cbr128_32k_sw": dict(bitrate=64, samprate=32000),
    "vbr50_thr100_mostly_short": dict(short_block_threshold=100),
    "cbr128_thr0_all_short": dict(bitrate=64, short_block_threshold=0),
}


def wants_bursts(kw):
    return kw.get("short_block_threshold", 700) < 99999


def api():
    from hmp3_amd import api as a
    return a


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_bytes(kw, pcm, nfr):
    enc = O.OracleEncoder(O.default_control(**kw))
    return b"".join(enc.encode_s16(pcm[f * 1152:(f + 1) * 1152]) for f in range(nfr))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_batch_bitstream_byte_identical_to_oracle(name):
    kw = CONFIGS[name]
    sr = kw.get("samprate", 44100)
    S, F = 12, 48
    pcm = np.stack([synth.stream_pcm(100 + i, F, sr=sr, rho=RHOS[i % 4], bursts=wants_bursts(kw)) for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    for s in range(S):
        assert got[s] == oracle_bytes(kw, pcm[s], F), "stream %d" % s
    if wants_bursts(kw):        # the case really exercises short blocks
        bt = b.debug_read("bt", np.uint8, S * 2 * F)
        assert (bt == 2).sum() > 0 and (bt == 1).sum() > 0 and (bt == 3).sum() > 0
    fb = [b.frames_bytes(s) for s in range(S)]
    assert all(f[1] == len(got[i]) for i, f in enumerate(fb))
    b.close()


@pytest.mark.parametrize("name", ["cli_cbr128_s16_44k", "cli_vbr75_f32_48k_hf", "cli_cbr192_s16_32k_x1_dc",
                                  "cli_vbr50_s24_44k", "cli_cbr128_u8_44k", "cli_vbr50_s32_48k",
                                  "cli_mono_cbr64_s16_44k", "cli_mono_vbr60_f32_48k",
                                  "cli_downmix_vbr50_s16_44k", "cli_downmix_cbr64_s24_48k",
                                  "cli_lsf_cbr64_s16_22k", "cli_lsf_vbr50_f32_24k", "cli_lsf_mono_cbr24_s16_16k",
                                  "cli_lsf_downmix_vbr80_s24_22k",
                                  "cli_rifx_cbr64_s16_44k", "cli_rifx_cbr64_s24_44k", "cli_rf64_vbr50_s16_48k", "cli_w64_cbr64_s24_44k",
                                  "cli_ext_vbr60_s24_48k", "cli_odd_cbr64_u8_mono_44k",
                                  "cli_src_11k_to_22k_s16", "cli_src_8k_to_16k_u8_mono", "cli_src_32k_to_44k_f32",
                                  "cli_src_48k_to_24k_s24", "cli_src_44k_to_32k_s16", "cli_src_44k_to_22k_downmix",
                                  "cli_src_44k_to_16k_f32_nopad", "cli_src_48k_to_22k_s24_nopad", "cli_src_24k_to_22k_s32_nopad",
                                  "cli_is_lsf_cbr32_s16_22k", "cli_is_lsf_cbr16_f32_16k", "cli_is_n8_cbr128_s16_44k",
                                  "cli_dual_cbr128_s24_44k", "cli_dual_lsf_cbr48_s16_24k"])
def test_cli_whole_file_byte_identical_to_reference_cli(name, tmp_path):
    """hmp3_amd/hmp3amd (GPU path + Xing/Info tag + WAV front end) against files written by the real
    reference CLI (tests/golden/cli_*.mp3, tests/golden/make_golden_cli.py)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_golden_cli as M
    seed, nsamp, sr, as_float, bursts, flags = M.CASES[name]
    wav, mp3 = str(tmp_path / "in.wav"), str(tmp_path / "out.mp3")
    M.write_wav(wav, M.case_pcm(name), sr, as_float, M.CONTAINER.get(name))
    exe = os.path.join(root, "hmp3_amd", "hmp3amd")
    assert os.path.exists(exe), "hmp3_amd/build.sh builds the CLI"
    r = subprocess.run([exe, wav, mp3] + flags, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert open(mp3, "rb").read() == open(os.path.join(GOLD, name + ".mp3"), "rb").read()


@pytest.mark.parametrize("kw", [dict(bitrate=64, short_block_threshold=99999), dict(bitrate=160), dict(vbr_mnr=50), dict(vbr_mnr=120, samprate=48000)],
                         ids=["cbr128_long", "cbr320", "vbr50", "vbr120_48k"])
def test_loud_near_mono_material_takes_the_double_table_path(kw):
    """L = R at full scale: the mid channel quantises beyond the 256-entry float table of ix^(4/3) at low gain steps and the
    noise terms come from the double-precision table (reference l3math.c:521-535 evaluates gain * pow(ix, 4/3) there)"""
    sr = kw.get("samprate", 44100)
    S, F = 8, 40
    pcm = np.stack([synth.stream_pcm(8100 + i, F, sr=sr, rho=1.0, bursts=wants_bursts(kw)) for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    b.debug_enable(True)
    got = b.encode_host(pcm)
    assert b.status() == 0
    for s in range(S):
        assert got[s] == oracle_bytes(kw, pcm[s], F), "stream %d" % s
    if "bitrate" in kw:     # (the CBR rate loop drives the gain steps low enough; VBR stays above)
        assert int(b.debug_read("big_sweeps", np.int32, 1)[0]) > 0, "the case does not reach the table's end"
    b.close()


@pytest.mark.parametrize("kw", [dict(bitrate=64, short_block_threshold=99999), dict(vbr_mnr=50), dict(bitrate=80, hf_flag=3, freq_limit=19000), dict(bitrate=64, samprate=32000),
                                dict(bitrate=32, samprate=22050)],
                         ids=["cbr128_long", "vbr50", "cbr160_hf", "cbr128_32k", "lsf_cbr64"])
def test_certified_band_sums_and_strict_band_sums_give_the_same_bytes(kw, monkeypatch):
    """The stream walk proves the mbLogC bucket of a band's noise from a parallel sum and an error interval, and adds the band's
    terms in the reference's line order only when the interval straddles a bucket boundary (hx_dev.h, "certified band sums").
    Both ways equal the oracle: with the real interval (a small share of the sums fall back: the device counter says how many)
    and with HMP3AMD_EXACT_SUMS=1, which sends every sum down the strict path.  rho cycled, so the double-table path is in."""
    sr = kw.get("samprate", 44100)
    S, F = 8, 40
    pcm = np.stack([synth.stream_pcm(8300 + i, F, sr=sr, rho=RHOS[i % 4], bursts=wants_bursts(kw)) for i in range(S)])
    want = [oracle_bytes(kw, pcm[s], F) for s in range(S)]
    counts = []
    for strict in ("0", "1"):
        monkeypatch.setenv("HMP3AMD_EXACT_SUMS", strict)
        b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
        got = b.encode_host(pcm)
        assert b.status() == 0
        for s in range(S):
            assert got[s] == want[s], ("strict sums" if strict == "1" else "certified sums", "stream %d" % s)
        counts.append(int(b.debug_read("strict_sums", np.int32, 1)[0]))
        b.close()
    # a few per cent of the gain-search sweeps have a band that needs the strict sum; forced, every sweep and every inverse_sf2 does
    assert counts[1] > 20 * max(counts[0], 1) and counts[1] > 4 * S * F, counts
    # ... and on real material the certified path does fall back now and then: a build that certified everything (a zero
    # half-width, a compare the wrong way round) would pass the byte comparison on these streams by luck and show up here
    assert 0 < counts[0] < counts[1] // 4, counts


NEG_SF_CASES = [   # found by tools/fuzz_parity.py: quiet dual-channel VBR material
    (dict(samprate=48000, mode=2, vbr_mnr=131), 109814, 0.0, 12, False),
    (dict(samprate=48000, mode=2, vbr_mnr=36, hf_flag=3, short_block_threshold=2000, filter_select=1), 370635, 0.7, 12, True),
]


@pytest.mark.parametrize("kw,seed,rho,F,bursts", NEG_SF_CASES, ids=["dual_vbr131", "dual_vbr36_hf"])
def test_negative_scalefactors_go_through_the_unmasked_bit_writer(kw, seed, rho, F, bursts):
    """The first-generation allocator leaves an empty band's scalefactor at 0 - pretab when pre-emphasis is on (reference
    bitallo1.cpp:547-586); the reference's bit writer ORs the negative value into its buffer unmasked, which sets bits
    written just before it.  k_pack replays the writer's flush state to put the same stray bits in (hx_pack.hip)."""
    skip_unless_host_libm_is_the_restated_one()
    pcm = synth.stream_pcm(seed, F, sr=kw["samprate"], rho=rho, bursts=bursts)
    pcm = (pcm.astype(np.float64) * 0.02).astype(np.int16)[None]
    b = api().Batch(api().default_control(**kw), nstreams=1, max_frames=F)
    got = b.encode_host(pcm)[0]
    assert b.status() == 0
    assert got == oracle_bytes(kw, pcm[0], F)
    b.close()


@pytest.mark.parametrize("kw", [dict(bitrate=64), dict(vbr_mnr=60)], ids=["cbr128", "vbr60"])
def test_packet_variant_matches_oracle(kw):
    """CMp3Enc::L3_audio_encode_Packet: bitstream plus the self-contained packet of every frame"""
    nfr = 30
    pcm = synth.stream_pcm(23, nfr, bursts=True).astype(np.float32)
    e = api().Mp3Enc()
    assert e.L3_audio_encode_init(api().default_control(**kw)) == 9216
    o = O.OracleEncoder(O.default_control(**kw))
    for f in range(nfr):
        nin, bs, pk = e.L3_audio_encode_Packet(pcm[f * 1152:(f + 1) * 1152])
        want_bs, want_pk = o.encode_packet(pcm[f * 1152:(f + 1) * 1152])
        assert nin == 9216 and bs == want_bs and pk == want_pk, "frame %d" % f
    e.close()


SHIM_CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "hmp3_on_amd")


@pytest.mark.skipif(not os.path.exists(SHIM_CLI), reason="oracle/_ref/hmp3_on_amd not built (make -C oracle shimcli)")
@pytest.mark.parametrize("name", ["cli_cbr128_s16_44k", "cli_vbr75_f32_48k_hf", "cli_cbr192_s16_32k_x1_dc", "cli_vbr50_s24_44k",
                                  "cli_mono_vbr60_f32_48k", "cli_downmix_vbr50_s16_44k", "cli_lsf_cbr64_s16_22k",
                                  "cli_lsf_mono_cbr24_s16_16k", "cli_rf64_vbr50_s16_48k", "cli_src_11k_to_22k_s16",
                                  "cli_src_44k_to_32k_s16", "cli_src_44k_to_22k_downmix"])
def test_reference_cli_source_on_the_drop_in_library(name, tmp_path):
    """The drop-in claim, exercised by the reference's own driver: its command line (test/tomp3.cpp with its WAV
    parser and tag writer, compiled unmodified where it lies) built against include/shim/mp3enc.h and -lhmp3amd
    must write the same files as the reference CLI built on the reference encoder (tests/golden/cli_*.mp3)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_golden_cli as M
    seed, nsamp, sr, as_float, bursts, flags = M.CASES[name]
    wav, mp3 = str(tmp_path / "in.wav"), str(tmp_path / "out.mp3")
    M.write_wav(wav, M.case_pcm(name), sr, as_float, M.CONTAINER.get(name))
    r = subprocess.run([SHIM_CLI, wav, mp3] + flags, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert open(mp3, "rb").read() == open(os.path.join(GOLD, name + ".mp3"), "rb").read()


@pytest.mark.parametrize("which", ["mpeg1", "mpeg2"])
def test_cli_batch_mode_files_byte_identical_to_reference_cli(tmp_path, which):
    """`hmp3amd -batch`: files of different rates / sample formats / lengths encoded as one batch of streams;
    every output file must equal what the reference CLI writes for that input alone"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_golden_cli as M
    inputs, bflags = (M.BATCH_INPUTS, M.BATCH_FLAGS) if which == "mpeg1" else (M.BATCH_LSF_INPUTS, M.BATCH_LSF_FLAGS)
    args = []
    for name in inputs:
        seed, nsamp, sr, as_float, bursts, flags = M.CASES[name]
        wav, mp3 = str(tmp_path / (name + ".wav")), str(tmp_path / (name + ".mp3"))
        M.write_wav(wav, M.case_pcm(name), sr, as_float)
        args += [wav, mp3]
    r = subprocess.run([os.path.join(root, "hmp3_amd", "hmp3amd"), "-batch"] + args + bflags, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    for name in inputs:
        got = open(str(tmp_path / (name + ".mp3")), "rb").read()
        assert got == open(os.path.join(GOLD, "batch_" + name[4:] + ".mp3"), "rb").read(), name


MONO = {
    "mono_cbr64": dict(bitrate=64, mode=3),
    "mono_vbr50": dict(mode=3),
    "mono_cbr96_48k": dict(bitrate=96, mode=3, samprate=48000),
    "mono_vbr100_hf_48k": dict(mode=3, vbr_mnr=100, hf_flag=3, samprate=48000, freq_limit=19000),
    "mono_cbr48_32k_long_dc": dict(bitrate=48, mode=3, samprate=32000, short_block_threshold=99999, filter_select=1),
}


@pytest.mark.parametrize("name", list(MONO))
def test_mono_batch_byte_identical_to_oracle(name):
    """mode 3: one channel, 17-byte side info, encode_singleB budgets; block switching on by default"""
    kw = MONO[name]
    sr = kw.get("samprate", 44100)
    S, F = 6, 40
    pcm = np.stack([synth.stream_pcm(500 + i, F, sr=sr, bursts=True)[:, i & 1] for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm[:, :25 * 1152])
    got2 = b.encode_host(pcm[:, 25 * 1152:])
    assert b.status() == 0
    for s in range(S):
        enc = O.OracleEncoder(O.default_control(**kw))
        want = b"".join(enc.encode_s16(pcm[s, f * 1152:(f + 1) * 1152]) for f in range(F))
        assert got[s] + got2[s] == want, "stream %d" % s
    b.close()


LSF = {
    # MPEG-2 LSF rates: every 1152-sample block yields two single-granule frames
    "lsf_cbr64_22k": dict(bitrate=32, samprate=22050),
    "lsf_cbr64_22k_long": dict(bitrate=32, samprate=22050, short_block_threshold=99999),
    "lsf_cbr64_24k_lr": dict(bitrate=32, samprate=24000, mode=0),
    "lsf_cbr48_16k": dict(bitrate=24, samprate=16000),
    "lsf_cbr160_24k": dict(bitrate=80, samprate=24000),
    "lsf_vbr50_22k": dict(samprate=22050),
    "lsf_vbr0_16k": dict(samprate=16000, vbr_mnr=0),
    "lsf_vbr150_24k": dict(samprate=24000, vbr_mnr=150),
    "lsf_cbr64_22k_dc": dict(bitrate=32, samprate=22050, filter_select=1),
    "lsf_thr0_all_short": dict(bitrate=32, samprate=22050, short_block_threshold=0),
    "lsf_mono_cbr32_22k": dict(bitrate=32, samprate=22050, mode=3),
    "lsf_mono_vbr_16k": dict(samprate=16000, mode=3),
    "lsf_mono_cbr8_16k": dict(bitrate=8, samprate=16000, mode=3),
}


@pytest.mark.parametrize("name", list(LSF))
def test_mpeg2_batch_byte_identical_to_oracle(name):
    kw = LSF[name]
    sr = kw["samprate"]
    mono = kw.get("mode") == 3
    S, F = 8, 40
    pcm = np.stack([synth.stream_pcm(700 + i, F, sr=sr, rho=RHOS[i % 4], bursts=wants_bursts(kw)) for i in range(S)])
    if mono:
        pcm = np.ascontiguousarray(pcm[:, :, 0])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm[:, :23 * 1152])
    got2 = b.encode_host(pcm[:, 23 * 1152:])
    assert b.status() == 0
    for s in range(S):
        assert got[s] + got2[s] == oracle_bytes(kw, pcm[s], F), "stream %d" % s
    fb = [b.frames_bytes(s) for s in range(S)]
    assert all(f[1] == len(got[i]) + len(got2[i]) for i, f in enumerate(fb))
    assert all(2 * F - 34 <= f[0] <= 2 * F for f in fb)        # two frames per block, less what is still pending
    b.close()


A1 = {
    # streams the reference codes with its first-generation allocator (CBitAllo1): joint stereo with an intensity part
    # (at MPEG-1 rates only on request, E_CONTROL nsbstereo / -N; at MPEG-2 rates below 48 kbps total) and dual channel
    "is_n8_cbr128_44k": (dict(bitrate=64, nsbstereo=8), 44100),
    "is_n4_cbr96_48k": (dict(bitrate=48, nsbstereo=4, samprate=48000), 48000),
    "is_n12_cbr112_32k": (dict(bitrate=56, nsbstereo=12, samprate=32000), 32000),
    "is_n16_cbr192_44k_dc": (dict(bitrate=96, nsbstereo=16, filter_select=1), 44100),
    "dual_cbr128": (dict(bitrate=64, mode=2), 44100),
    "dual_cbr96_48k": (dict(bitrate=48, mode=2, samprate=48000), 48000),
    "dual_cbr320": (dict(bitrate=160, mode=2), 44100),
    "lsf_is_cbr32_22k": (dict(bitrate=16, samprate=22050), 22050),
    "lsf_is_cbr16_16k": (dict(bitrate=8, samprate=16000), 16000),
    "lsf_is_cbr40_24k": (dict(bitrate=20, samprate=24000), 24000),
    "lsf_dual_cbr64_22k": (dict(bitrate=32, samprate=22050, mode=2), 22050),
    "lsf_dual_cbr32_16k": (dict(bitrate=16, samprate=16000, mode=2), 16000),
}


@pytest.mark.parametrize("name", list(A1))
def test_intensity_stereo_and_dual_channel_byte_identical_to_oracle(name):
    """the kernels k_alloc1 / k_alloc1_lsf (hx_alloc1.inc): 8 streams of differing channel correlation, ragged calls"""
    skip_unless_host_libm_is_the_restated_one()
    kw, sr = A1[name]
    S, F = 8, 40
    pcm = np.stack([synth.stream_pcm(5200 + i, F, sr=sr, rho=RHOS[i % 4], bursts=(i % 2 == 0)) for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=24)
    got = [b"" for _ in range(S)]
    pos = 0
    for n in (3, 24, 13):
        out = b.encode_host(pcm[:, pos * 1152:(pos + n) * 1152])
        pos += n
        for s in range(S):
            got[s] += out[s]
    assert b.status() == 0
    for s in range(S):
        assert got[s] == oracle_bytes(kw, pcm[s], F), (name, s)
    b.close()


def test_intensity_stereo_stress_signals():
    """full-scale noise, a pure tone on both channels, silence, near-silence, anti-phase tone: the allocator's bit
    seek in both directions, silent channels, intensity positions at the extremes"""
    skip_unless_host_libm_is_the_restated_one()
    F = 24
    n = F * 1152
    rng = np.random.default_rng(99)
    t = np.arange(n)
    noise = rng.integers(-32768, 32768, (n, 2)).astype(np.int16)
    tone = np.round(32767 * np.sin(2 * np.pi * 110.0 * t / 44100.0)).astype(np.int16)
    sigs = [noise, np.stack([tone, tone], axis=1), np.zeros((n, 2), np.int16), (noise // 4096).astype(np.int16),
            np.stack([tone, -tone], axis=1), np.stack([tone, np.zeros(n, np.int16)], axis=1)]
    pcm = np.stack(sigs)
    for kw in (dict(bitrate=64, nsbstereo=6), dict(bitrate=64, mode=2), dict(bitrate=16, samprate=22050), dict(bitrate=32, samprate=16000, mode=2)):
        b = api().Batch(api().default_control(**kw), nstreams=len(sigs), max_frames=F)
        got = b.encode_host(pcm)
        assert b.status() == 0
        for s in range(len(sigs)):
            assert got[s] == oracle_bytes(kw, pcm[s], F), (kw, s)
        b.close()


def test_mpeg2_and_mpeg1_cannot_share_a_batch():
    ecs = [api().default_control(bitrate=32, samprate=22050), api().default_control(bitrate=64)]
    with pytest.raises(RuntimeError, match="cannot share a batch"):
        api().Batch(ecs, nstreams=2, max_frames=4)


@pytest.mark.parametrize("kw", [dict(bitrate=32, samprate=22050), dict(samprate=24000, vbr_mnr=80), dict(bitrate=32, samprate=16000, mode=3)],
                         ids=["cbr64_22k", "vbr80_24k", "mono_cbr32_16k"])
def test_mpeg2_packet_variant_matches_oracle(kw):
    """L3_audio_encode_Packet at an MPEG-2 rate: two packets per call, nbytes_out[0..1]"""
    nfr = 30
    pcm = synth.stream_pcm(23, nfr, sr=kw["samprate"], bursts=True).astype(np.float32)
    nin_want = 9216
    if kw.get("mode") == 3:
        pcm = np.ascontiguousarray(pcm[:, 0])
        nin_want = 4608
    e = api().Mp3Enc()
    assert e.L3_audio_encode_init(api().default_control(**kw)) == nin_want
    o = O.OracleEncoder(O.default_control(**kw))
    for f in range(nfr):
        nin, bs, pk = e.L3_audio_encode_Packet(pcm[f * 1152:(f + 1) * 1152])
        want_bs, want_pk = o.encode_packet(pcm[f * 1152:(f + 1) * 1152])
        assert nin == nin_want and bs == want_bs and pk == want_pk and e.packet_sizes == o.packet_sizes, "frame %d" % f
    fr, by = e.L3_audio_encode_get_frames_bytes()
    assert fr > 40 and abs(e.L3_audio_encode_get_bitrate_float() - 0.008 * by * kw["samprate"] / (576.0 * fr)) < 1e-2
    e.close()


def test_pipelined_submit_equals_plain_calls_and_oracle():
    """hx_batch_submit_s16_device / hx_batch_wait: consecutive calls overlap on two internal streams (the front end
    of call n+1 gated into the tail of call n's allocator kernel); the bytes must not change, also when plain calls
    are mixed in.  Every stream against a batch driven by plain calls, a sample of them against the oracle."""
    import torch
    kw = dict(bitrate=64)
    S, F = 512, 12
    calls = 6
    pcm = np.stack([synth.stream_pcm(900 + i, F * calls, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    d_pcm = [torch.from_numpy(np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152])).to(dev) for c in range(calls)]
    got = {}
    for mode in ("plain", "submit_gate90", "submit_gate0"):
        b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
        if mode == "submit_gate0":
            b.set_gate(0)
        stride = b.out_stride(F)
        d_out = [torch.zeros((S, stride), dtype=torch.uint8, device=dev) for _ in range(calls)]
        d_nb = [torch.zeros((S,), dtype=torch.int32, device=dev) for _ in range(calls)]
        torch.cuda.synchronize()
        for c in range(calls):
            if mode == "plain" or c == 3:      # a plain call in the middle of the submits
                b.encode_device(d_pcm[c].data_ptr(), F, d_out[c].data_ptr(), stride, d_nb[c].data_ptr(), st)
            else:
                b.submit_device(d_pcm[c].data_ptr(), F, d_out[c].data_ptr(), stride, d_nb[c].data_ptr(), st)
        b.wait(st)
        torch.cuda.synchronize()
        assert b.status() == 0
        outs = [o.cpu().numpy() for o in d_out]
        nbs = [n.cpu().numpy() for n in d_nb]
        got[mode] = [b"".join(outs[c][s, :nbs[c][s]].tobytes() for c in range(calls)) for s in range(S)]
        b.close()
    assert got["submit_gate90"] == got["plain"] and got["submit_gate0"] == got["plain"]
    # (512 streams on 256 CUs: the launch order is the previous call's durations and, from the second call on, the CU-mates
    # of the eight longest streams park - every stream against the oracle)
    from oracle_pool import oracle_bytes_many
    want = oracle_bytes_many(kw, pcm, F * calls)
    bad = [s for s in range(S) if got["plain"][s] != want[s]]
    assert not bad, "%d of %d streams differ from the oracle, first: %s" % (len(bad), S, bad[:8])


@pytest.mark.one_k6_build
@pytest.mark.parametrize("park", ["0", "8", "64"])
def test_parking_changes_no_byte_and_the_placement_tap_names_every_cu(park, monkeypatch):
    """HMP3AMD_PARK (hx_alloc3.inc, "parking"): the workgroups that share a CU with one of the launch's longest streams keep
    their slot until that stream retires.  Scheduling only: same bytes as the oracle with it off, at its default and at its
    maximum; the placement tap ("place") reports (XCC, SE, CU) per position of the launch order - 1024 streams of the
    four-per-CU kernel land four to a CU on 256 distinct CUs."""
    import torch
    monkeypatch.setenv("HMP3AMD_PARK", park)
    monkeypatch.setenv("HMP3AMD_K6", "fat")
    kw = dict(bitrate=64, short_block_threshold=99999)
    S, F, calls = 1024, 6, 3
    base = [synth.stream_pcm(7100 + u, F * calls, rho=RHOS[u % 4]) for u in range(64)]
    pcm = np.stack([np.roll(base[i % 64], 1152 * (i // 64), axis=0) for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    got = [b"" for _ in range(S)]
    for c in range(calls):
        out = b.encode_host(pcm[:, c * F * 1152:(c + 1) * F * 1152])
        for s in range(S):
            got[s] += out[s]
    assert b.status() == 0
    place = b.debug_read("place", np.uint32, S)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    cus, counts = np.unique(place & 0xFFFFFF00, return_counts=True)
    assert len(cus) == min(ncu, S) and counts.max() <= 4 and len(np.unique(place >> 16)) == 8, (len(cus), counts.max())
    b.close()
    from oracle_pool import oracle_bytes_many
    want = oracle_bytes_many(kw, pcm, F * calls)
    bad = [s for s in range(S) if got[s] != want[s]]
    assert not bad, "%d of %d streams differ from the oracle, first: %s" % (len(bad), S, bad[:8])


@pytest.mark.one_k6_build
def test_persistent_workgroups_walk_more_streams_than_slots_over_several_calls(monkeypatch):
    """Beyond the resident set k_alloc_slim runs as many workgroups as the chip holds and each claims stream after stream of
    the launch order from a counter (hx_alloc3.inc, HX_PERSIST): 1600 streams on 1536 slots, three calls (the second and third
    in the order of the previous call's durations), every stream against the oracle; the counter is back at zero after
    every launch (the next one starts from position 0: all 1600 streams come out)."""
    monkeypatch.setenv("HMP3AMD_K6", "slim")
    kw = dict()      # VBR -V50, block switching
    S, F, calls = 1600, 5, 3
    base = [synth.stream_pcm(7300 + u, F * calls, rho=RHOS[u % 4], bursts=True) for u in range(50)]
    pcm = np.stack([np.roll(base[i % 50], 1152 * (i // 50), axis=0) for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    assert b.k6_variant() == 1 and b.resident_streams() < S
    got = [b"" for _ in range(S)]
    for c in range(calls):
        out = b.encode_host(pcm[:, c * F * 1152:(c + 1) * F * 1152])
        for s in range(S):
            got[s] += out[s]
    assert b.status() == 0
    b.close()
    from oracle_pool import oracle_bytes_many
    want = oracle_bytes_many(kw, pcm, F * calls)
    bad = [s for s in range(S) if got[s] != want[s]]
    assert not bad, "%d of %d streams differ from the oracle, first: %s" % (len(bad), S, bad[:8])


def test_pipelined_host_calls_equal_plain_host_calls():
    """hx_batch_submit_s16_host / hx_batch_wait_host with page-locked buffers: copies in both directions overlap the
    encoding of the neighbouring calls; every byte must equal what plain host calls return"""
    import torch
    kw = dict(vbr_mnr=60)
    S, F, calls = 96, 10, 5
    pcm = np.stack([synth.stream_pcm(1200 + i, F * calls, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    b0 = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    want = [b0.encode_host(pcm[:, c * F * 1152:(c + 1) * F * 1152]) for c in range(calls)]
    b0.close()
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    stride = b.out_stride(F)
    h_pcm = [torch.from_numpy(np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152])).pin_memory() for c in range(calls)]
    h_out = [torch.zeros((S, stride), dtype=torch.uint8).pin_memory() for _ in range(calls)]
    h_nb = [torch.zeros((S,), dtype=torch.int32).pin_memory() for _ in range(calls)]
    for c in range(calls):
        b.submit_host(h_pcm[c].data_ptr(), F, h_out[c].data_ptr(), stride, h_nb[c].data_ptr())
    b.wait_host()
    assert b.status() == 0
    for c in range(calls):
        o, n = h_out[c].numpy(), h_nb[c].numpy()
        for s in range(S):
            assert o[s, :n[s]].tobytes() == want[c][s], "call %d stream %d" % (c, s)
    # pageable memory works too (the copies just do not overlap)
    b2 = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    p_out = [np.zeros((S, stride), np.uint8) for _ in range(calls)]
    p_nb = [np.zeros((S,), np.int32) for _ in range(calls)]
    p_pcm = [np.ascontiguousarray(pcm[:, c * F * 1152:(c + 1) * F * 1152]) for c in range(calls)]
    for c in range(calls):
        b2.submit_host(p_pcm[c].ctypes.data, F, p_out[c].ctypes.data, stride, p_nb[c].ctypes.data)
    b2.wait_host()
    for c in range(calls):
        for s in range(0, S, 7):
            assert p_out[c][s, :p_nb[c][s]].tobytes() == want[c][s]
    b.close(); b2.close()


def random_control(rng):
    """a random but accepted E_CONTROL: rate, mode, CBR/VBR, -HF, DC filter, short-block threshold, tuning knobs"""
    sr = int(rng.choice([44100, 48000, 32000, 22050, 24000, 16000]))
    lsf = sr < 32000
    kw = dict(samprate=sr, mode=int(rng.choice([0, 1, 1, 3])))
    if rng.random() < 0.5:
        kw["bitrate"] = int(rng.choice([24, 32, 40, 48, 56, 64, 80] if lsf else [48, 56, 64, 80, 96, 112, 128, 160]))
    else:
        kw["vbr_mnr"] = int(rng.integers(0, 151))
        if rng.random() < 0.3:
            kw["vbr_br_limit"] = int(rng.choice([40, 64, 96, 160]))
        if rng.random() < 0.3:
            kw["vbr_delta_mnr"] = int(rng.integers(-40, 51))
    if not lsf and rng.random() < 0.3:
        kw["hf_flag"] = int(rng.choice([1, 3]))
        kw["freq_limit"] = int(rng.choice([17000, 19000, 22000, 24000]))
    if rng.random() < 0.2:
        kw["filter_select"] = 1
    if rng.random() < 0.3:
        kw["short_block_threshold"] = int(rng.choice([0, 100, 400, 99999]))
    if rng.random() < 0.2:
        kw["freq_limit"] = int(rng.choice([6000, 9000, 14000]))
    if rng.random() < 0.2:
        kw["test1"] = int(rng.integers(0, 12))
    if rng.random() < 0.15:
        kw["quick"] = 1
    if rng.random() < 0.2:
        kw["nsb_limit"] = int(rng.choice([2, 4, 6, 10, 14, 18, 22, 26, 30]))      # subband limit set by the caller
    return kw


def test_random_configurations_against_the_oracle():
    """60 random accepted controls (rates of both MPEG versions, mono / stereo / joint stereo, CBR / VBR, -HF modes,
    DC filter, thresholds, tuning knobs), three streams each, through ragged calls"""
    rng = np.random.default_rng(20240917)
    done = 0
    for trial in range(200):
        kw = random_control(rng)
        if not O.OracleEncoder(O.default_control(**kw)).ok():
            continue        # out of scope for both (intensity stereo) or rejected by the reference
        sr, mono = kw["samprate"], kw["mode"] == 3
        S, F = 3, 22
        pcm = np.stack([synth.stream_pcm(3000 + 7 * trial + i, F, sr=sr, rho=RHOS[(trial + i) % 4], bursts=True) for i in range(S)])
        if mono:
            pcm = np.ascontiguousarray(pcm[:, :, 0])
        b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
        cut = int(rng.integers(1, F))
        got = b.encode_host(pcm[:, :cut * 1152])
        got2 = b.encode_host(pcm[:, cut * 1152:])
        assert b.status() == 0, kw
        for s in range(S):
            assert got[s] + got2[s] == oracle_bytes(kw, pcm[s], F), (kw, s)
        b.close()
        done += 1
        if done == 60:
            break
    assert done == 60


@pytest.mark.parametrize("kw", [dict(samprate=48000, bitrate=64, nsb_limit=4), dict(samprate=48000, mode=0, bitrate=112, hf_flag=3, freq_limit=8000, nsb_limit=4),
                                dict(samprate=44100, bitrate=160, nsb_limit=2), dict(samprate=32000, vbr_mnr=120, nsb_limit=3)],
                         ids=["48k_c4", "48k_c4_hf", "44k_c2", "32k_c3"])
def test_very_low_subband_limits(kw):
    """E_CONTROL.nsb_limit of a handful of subbands (72 lines or fewer are coded): the scalefactor refinement's work list shares the
    line buffer, and lines past the coded range must still come out zero (found by the round-3 sweep, 3 cases in 2500)"""
    S, F = 6, 12
    pcm = np.stack([synth.stream_pcm(656405 + i, F, sr=kw["samprate"], rho=[1.0, 0.7, 0.0, 0.3][i % 4], bursts=(i % 2 == 0)) for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    for s in range(S):
        assert got[s] == oracle_bytes(kw, pcm[s], F), s
    b.close()


@pytest.mark.parametrize("kw,seed,rho,F,cut,bursts", [
    (dict(samprate=44100, mode=3, bitrate=160, hf_flag=3, nsbstereo=16, nsb_limit=28), 721109, 1.0, 12, 12, False),
    (dict(samprate=44100, mode=0, bitrate=160, hf_flag=3, filter_select=1), 395960, 0.7, 120, 20, True)], ids=["mono_hf3_160k", "stereo_hf3_160k"])
def test_band_21_lines_of_an_earlier_hf_pass_survive_the_rate_loop(kw, seed, rho, F, cut, bursts):
    """-HF 3, L/R or mono granules: the rate loop's step towards more bits re-quantises without clearing band 21 (reference
    bitallo3.cpp:2636-2668), so when the step drops the -HF quantisation the lines of the earlier pass stay in the buffer - out of the
    counted range, except that the last quadruple reaches up to three lines into them.  The low-footprint kernel's quantiser wrote
    zeros there (found by the round-4 sweep: 2 cases in 3000, first differing frame 16 and frame 1 of these two streams)."""
    pcm = synth.stream_pcm(seed, F, sr=44100, rho=rho, bursts=bursts)[None, :cut * 1152]
    if kw.get("mode") == 3:
        pcm = np.ascontiguousarray(pcm[:, :, 0])
    b = api().Batch(api().default_control(**kw), nstreams=1, max_frames=cut)
    got = b.encode_host(pcm)
    assert b.status() == 0
    assert got[0] == oracle_bytes(kw, pcm[0], cut)
    b.close()


def test_reset_stream_starts_a_fresh_stream_in_a_running_batch():
    """hx_batch_reset_stream: a slot of a long-lived batch takes over a new input; its output equals a fresh encode
    of that input, and the neighbouring streams carry on undisturbed"""
    kws = [dict(bitrate=64), dict(vbr_mnr=70), dict(bitrate=64, samprate=48000)]
    S, F = 6, 18
    ctl = [api().default_control(**kws[i % 3]) for i in range(S)]
    srs = [kws[i % 3].get("samprate", 44100) for i in range(S)]
    first = np.stack([synth.stream_pcm(1500 + i, F, sr=srs[i], bursts=True) for i in range(S)])
    second = np.stack([synth.stream_pcm(1600 + i, F, sr=srs[i], bursts=True) for i in range(S)])
    b = api().Batch(ctl, nstreams=S, max_frames=F)
    out1 = b.encode_host(first)
    for i in (1, 2, 5):
        b.reset_stream(i)
    out2 = b.encode_host(second)
    assert b.status() == 0
    for i in range(S):
        if i in (1, 2, 5):      # a new stream: as if encoded from scratch
            assert out1[i] == oracle_bytes(kws[i % 3], first[i], F)
            assert out2[i] == oracle_bytes(kws[i % 3], second[i], F), i
        else:                   # an old stream: one continuous encode of both halves
            assert out1[i] + out2[i] == oracle_bytes(kws[i % 3], np.concatenate([first[i], second[i]]), 2 * F), i
    b.close()


def test_stream_checkpoint_resumes_in_another_batch():
    """hx_batch_get / set_stream_state: streams saved from one batch continue in other slots of a new batch (other
    size, other max_frames) with the bitstream of an uninterrupted encode"""
    kw = dict(vbr_mnr=55)
    F1, F2 = 13, 17
    pcm = np.stack([synth.stream_pcm(1700 + i, F1 + F2, rho=RHOS[i % 4], bursts=True) for i in range(4)])
    b1 = api().Batch(api().default_control(**kw), nstreams=4, max_frames=F1)
    out1 = b1.encode_host(pcm[:, :F1 * 1152])
    saved = [b1.get_stream_state(i) for i in range(4)]
    b1.close()
    b2 = api().Batch(api().default_control(**kw), nstreams=6, max_frames=F2)
    place = [5, 0, 3, 2]        # stream i of the old batch goes to slot place[i]
    for i, slot in enumerate(place):
        b2.set_stream_state(slot, saved[i])
    pcm2 = np.zeros((6, F2 * 1152, 2), np.int16)
    for i, slot in enumerate(place):
        pcm2[slot] = pcm[i, F1 * 1152:]
    out2 = b2.encode_host(pcm2)
    assert b2.status() == 0
    for i, slot in enumerate(place):
        assert out1[i] + out2[slot] == oracle_bytes(kw, pcm[i], F1 + F2), i
    b2.close()


def test_float_input_and_dc_filter_mixed_batch():
    """fp32 PCM at int16 scale with non-integral samples (the L3_audio_encode form), half of the
    streams with the DC blocker on"""
    S, F = 6, 24
    rng = np.random.default_rng(9)
    pcm = np.stack([synth.stream_pcm(300 + i, F, rho=RHOS[i % 4]) for i in range(S)]).astype(np.float32)
    pcm += rng.uniform(-0.49, 0.49, pcm.shape).astype(np.float32)
    pcm[1] += 700.0         # a DC offset for the blocker to remove
    kws = [dict(bitrate=64, short_block_threshold=99999, filter_select=(i & 1)) for i in range(S)]
    b = api().Batch([api().default_control(**k) for k in kws], nstreams=S, max_frames=F)
    got = b.encode_host(pcm[:, :12 * 1152])
    got2 = b.encode_host(pcm[:, 12 * 1152:])
    assert b.status() == 0
    for s in range(S):
        enc = O.OracleEncoder(O.default_control(**kws[s]))
        want = b"".join(enc.encode_f32(pcm[s, f * 1152:(f + 1) * 1152]) for f in range(F))
        assert got[s] + got2[s] == want, "stream %d" % s
    b.close()


@pytest.mark.parametrize("name", ["cbr128_long", "vbr50_long", "vbr100_hf2_48k_long", "cbr128_32k_long",
                                  "cbr128_default_bursts", "vbr50_default_bursts"])
def test_golden_reference_streams(name):
    """committed reference bitstreams (tests/golden, captured from the real reference)"""
    m = META[name]
    nfr = m["frames"]
    pcm = synth.stream_pcm(m["stream_seed"], nfr, sr=m["samprate"], rho=m["rho"], bursts=m["bursts"])
    pcm = np.concatenate([pcm, np.zeros((2 * 1152, 2), dtype=np.int16)])[None]
    b = api().Batch(api().default_control(**m["control"]), nstreams=1, max_frames=nfr + 2)
    got = b.encode_host(pcm)[0]
    assert got == open(os.path.join(GOLD, name + ".mp3frames"), "rb").read()
    b.close()


@pytest.mark.parametrize("name", ["a1_dual_16k_antiphase", "stab_odd_npart_32k"])
def test_extra_golden_reference_streams(name):
    """committed reference bitstreams of make_golden.EXTRA_CASES (signals that are more than a seed): the dual-channel
    stream of the first-generation allocator that tells libm's log10f / logf (hx_libm32.h) from the double functions; the
    32 kHz stream on which the reference's psy model reads a local it never wrote (golden from the zeroed-locals build)"""
    sys.path.insert(0, GOLD)
    import make_golden as M
    kw, nfr = M.EXTRA_CASES[name][0], M.EXTRA_CASES[name][2]
    pcm = np.concatenate([M.extra_case_pcm(name), np.zeros((2 * 1152, 2), dtype=np.int16)])
    b = api().Batch(api().default_control(**kw), nstreams=3, max_frames=nfr + 2)
    got = b.encode_host(np.stack([pcm, pcm, pcm]))
    assert b.status() == 0
    want = open(os.path.join(GOLD, name + ".mp3frames"), "rb").read()
    assert got[0] == want and got[1] == want and got[2] == want
    b.close()


STAGE_CASES = {
    # name: (control, sample rate, bursts, what the oracle taps for this path)
    "cbr128_long": (dict(bitrate=64, short_block_threshold=99999), 44100, False, "full"),
    "vbr50_sw": (dict(), 44100, True, "full"),                              # block types 0 / 1 / 2 / 3, short spectra [3][192]
    "vbr100_hf2_48k_sw": (dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), 48000, True, "full"),
    "cbr128_lr_sw": (dict(bitrate=64, mode=0), 44100, True, "full"),        # plain stereo: the L/R start values
    "mono_vbr50": (dict(mode=3), 44100, True, "lines"),
    "lsf_cbr64_22k": (dict(bitrate=32, samprate=22050), 22050, True, "lines"),
    "lsf_mono_vbr_16k": (dict(samprate=16000, mode=3), 16000, True, "lines"),
}


@pytest.mark.parametrize("name", list(STAGE_CASES))
def test_every_stage_bit_exact(name):
    """Stage by stage against the oracle's taps, bit patterns compared with == : K1 polyphase, K4 MDCT (long and
    short layouts) and psy model, block types, K5 stereo decision, K6 side info / scalefactors / reservoir, and
    what K6 hands to the packer (quantised lines and signs).  Mono and MPEG-2 paths: the taps the oracle has there
    (spectrum, block types, quantised lines, signs)."""
    kw, sr, bursts, taps = STAGE_CASES[name]
    mono = kw.get("mode") == 3
    S, F = 4, 24
    NG = 2 * F
    pcm = np.stack([synth.stream_pcm(200 + i, F, sr=sr, rho=RHOS[i % 4], bursts=bursts) for i in range(S)])
    if mono:
        pcm = pcm[:, :, 0]
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    b.debug_enable(True)
    got = b.encode_host(pcm)
    assert b.status() == 0
    sb = b.debug_read("sb", np.float32, S * 2 * (NG + 3) * 576).reshape(S, 2, NG + 3, 576)
    xr = b.debug_read("xr", np.float32, S * NG * 1152).reshape(S, NG, 2, 576)
    etab = b.debug_read("etab", np.float32, S * NG * 128).reshape(S, NG, 2, 64)
    thr = b.debug_read("thr", np.float32, S * NG * 128).reshape(S, NG, 2, 64)
    btg = b.debug_read("bt", np.uint8, S * NG).reshape(S, NG)
    ixq = b.debug_read("ixq", np.int16, S * NG * 1152).reshape(S, NG, 2, 576).astype(np.int32) & 0xFFFF
    # the signs travel as one bit per line (bit j & 31 of word j >> 5; 20 words per granule and channel, 18 used)
    sgw = b.debug_read("sgn", np.uint32, S * NG * 2 * 20).reshape(S, NG, 2, 20)
    sgn = np.unpackbits(sgw[..., :18].copy().view(np.uint8), axis=-1, bitorder="little").reshape(S, NG, 2, 576)
    npart = np.zeros(1, np.int32)
    assert api().lib().hx_debug_host_table(C.byref(api().default_control(**kw)), b"psy_npart", npart.ctypes.data, 4) == 4
    np2 = int(npart[0] + 1) & ~1

    class GDbg(C.Structure):
        _fields_ = [("ms", C.c_int), ("ms_metric", C.c_int * 2), ("byte_pool", C.c_int), ("MNR_after", C.c_int),
                    ("mask_mb", C.c_int * 88), ("gr", C.c_int * 96), ("sf", C.c_int * 88), ("scfsi", C.c_int * 2),
                    ("main_bytes", C.c_int)]
    raw = b.debug_read("dbg", np.uint8, S * F * C.sizeof(GDbg))
    seen_bt = set()
    for s in range(S):
        enc = O.OracleEncoder(O.default_control(**kw))
        d = O.oracle_enable_debug(enc)
        out = []
        for f in range(F):
            out.append(enc.encode_s16(pcm[s, f * 1152:(f + 1) * 1152]))
            xp = np.array(d.xr_pre).reshape(2, 2, 576)
            oix = np.array(d.ix).reshape(2, 2, 576)
            osg = np.array(d.signx).reshape(2, 2, 576)
            ogr = np.array(d.gr).reshape(2, 2, 27)
            gd = GDbg.from_buffer_copy(raw[(s * F + f) * C.sizeof(GDbg):(s * F + f + 1) * C.sizeof(GDbg)].tobytes())
            ggr = np.array(gd.gr).reshape(2, 2, 24)
            for igr in range(2):
                g = 2 * f + igr
                bt = int(d.block_type[igr])
                seen_bt.add(bt)
                assert btg[s, g] == bt, ("block type", s, f, igr)
                for ch in range(1 if mono else 2):
                    assert np.array_equal(bits(xr[s, g, ch]), bits(xp[igr, ch])), ("mdct", s, f, igr, ch, bt)
                    # what K6 hands to k_pack: the quantised lines of the coded range and the signs of the non-zero ones
                    n = 2 * int(ggr[igr, ch, 1]) + 4 * max(int(ggr[igr, ch, 18]), 0) if ggr[igr, ch, 20] else 0
                    assert np.array_equal(ixq[s, g, ch, :n], oix[igr, ch, :n]), ("ix", s, f, igr, ch, bt)
                    nz = oix[igr, ch, :n] != 0
                    assert np.array_equal(sgn[s, g, ch, :n][nz], osg[igr, ch, :n][nz]), ("signs", s, f, igr, ch, bt)
                    assert np.array_equal(ggr[igr, ch], ogr[igr, ch, :24]), ("side info", s, f, igr, ch, bt)
            if taps != "full":
                continue
            sn = np.array(d.sample_new).reshape(2, 2, 576)
            oe = np.array(d.etab).reshape(2, 2, 64)
            ot = np.array(d.thr).reshape(2, 2, 64)
            for igr in range(2):
                g = 2 * f + igr
                for ch in range(2):
                    assert np.array_equal(bits(sb[s, ch, 3 + g]), bits(sn[igr, ch])), ("polyphase", s, f, igr, ch)
                    if d.block_type[igr] != 2:      # (the oracle taps the long model's partition tables)
                        assert np.array_equal(bits(etab[s, g, ch, :np2]), bits(oe[igr, ch, :np2])), ("etab", s, f, igr, ch)
                        assert np.array_equal(bits(thr[s, g, ch, :np2]), bits(ot[igr, ch, :np2])), ("thr", s, f, igr, ch)
            assert gd.ms == d.ms and list(gd.ms_metric) == list(d.ms_metric)
            assert gd.byte_pool == d.byte_pool and gd.MNR_after == d.MNR_after and gd.main_bytes == d.main_bytes
            if d.block_type[0] != 2 and d.block_type[1] != 2:
                assert np.array_equal(np.array(gd.sf), np.array(d.sf))
                assert list(gd.scfsi) == list(d.scfsi)
        assert got[s] == b"".join(out)
    if bursts and "long" not in name:
        assert {0, 1, 2, 3} <= seen_bt, seen_bt      # every block type went through the stage comparison
    b.close()


@pytest.mark.parametrize("cfg", ["config3", "config4_share", "config5_share"])
def test_full_batch_configs_3_and_5(cfg):
    """BASELINE configs 3 (4096 streams, VBR -V50, block switching), one GPU's share of config 4 (4096 streams, 48 kHz,
    VBR -V100 -HF2 -F19000) and of config 5 (4096 streams,
    32 / 44.1 / 48 kHz by stream, CBR-128, correlation cycled) at full batch width with 256 distinct signals:
    frame structure of every stream, and byte equality with the oracle on a 16-stream subset."""
    a = api()
    S, F = 4096, 32
    U = 252 if cfg == "config5_share" else 256  # distinct signals; a multiple of the class count (3) and of the correlation cycle (4)
    if cfg == "config3":
        classes = [(dict(), 44100)]
    elif cfg == "config4_share":
        classes = [(dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), 48000)]
    else:
        classes = [(dict(bitrate=64, samprate=32000), 32000), (dict(bitrate=64), 44100), (dict(bitrate=64, samprate=48000), 48000)]
    base = [synth.stream_pcm(9000 + u, F, sr=classes[u % len(classes)][1], rho=RHOS[u % 4], bursts=True) for u in range(U)]
    # stream i carries signal i mod U, rotated in time for the later copies
    pcm = np.stack([np.roll(base[i % U], 1152 * 3 * (i // U), axis=0) for i in range(S)])
    ctl = [a.default_control(**classes[(i % U) % len(classes)][0]) for i in range(S)]
    b = a.Batch(ctl if len(classes) > 1 else ctl[0], nstreams=S, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    rng = np.random.Generator(np.random.PCG64(11))
    for s in rng.choice(S, 16, replace=False):
        assert got[s] == oracle_bytes(classes[(s % U) % len(classes)][0], pcm[s], F), "stream %d" % s
    for s in range(0, S, 61):       # every frame starts with a sync word and the sizes add up
        bs, pos, n = got[s], 0, 0
        kw, sr = classes[(s % U) % len(classes)]
        while pos < len(bs):
            assert bs[pos] == 0xFF and (bs[pos + 1] & 0xFE) == 0xFA
            br = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320][bs[pos + 2] >> 4]
            pos += 144000 * br // sr + ((bs[pos + 2] >> 1) & 1)
            n += 1
        assert pos == len(bs) and F - 6 <= n <= F
    assert len(set(got[:U])) == U       # distinct signals give distinct streams
    b.close()


def test_config_4_share_at_full_length():
    """One GPU's share of BASELINE config 4 at its full size - 4096 streams x 128 frames, 48 kHz, VBR -V100 -HF2 -F19000 - in
    one call (the 4096-stream configs otherwise run at 32 frames here and at full length only in bench.py's verify): 64
    distinct signals rotated in time, ALL 4096 streams byte for byte against the oracle on the low-footprint build (a 64-stream
    subset on the forced four-stream build), frame structure of every 97th stream"""
    a = api()
    S, F, U = 4096, 128, 64
    kw, sr = dict(samprate=48000, vbr_mnr=100, hf_flag=3, freq_limit=19000), 48000
    base = [synth.stream_pcm(9900 + u, F, sr=sr, rho=RHOS[u % 4], bursts=True) for u in range(U)]
    pcm = np.stack([np.roll(base[i % U], 1152 * 2 * (i // U), axis=0) for i in range(S)])
    b = a.Batch(a.default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    assert b.k6_variant() == (1 if os.environ.get("HMP3AMD_K6", "slim") == "slim" else 0)      # beyond the resident set: the low-footprint build unless forced
    from oracle_pool import oracle_bytes_many
    ids = list(range(S)) if os.environ.get("HMP3AMD_K6", "slim") == "slim" else list(range(1, S, 64))   # (the forced four-stream build: a 64-stream subset)
    want = oracle_bytes_many(kw, pcm, F, ids)
    bad = [s for s in ids if got[s] != want[s]]
    assert not bad, "%d of %d streams differ from the oracle, first: %s" % (len(bad), len(ids), bad[:8])
    for s in range(0, S, 97):
        bs, pos, n = got[s], 0, 0
        while pos < len(bs):
            assert bs[pos] == 0xFF and (bs[pos + 1] & 0xFE) == 0xFA
            br = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320][bs[pos + 2] >> 4]
            pos += 144000 * br // sr + ((bs[pos + 2] >> 1) & 1)
            n += 1
        assert pos == len(bs) and F - 6 <= n <= F
    b.close()


@pytest.mark.one_k6_build
@pytest.mark.parametrize("cfg", ["config3", "config5_share"])
def test_configs_3_and_5_at_full_length_on_the_low_footprint_build(cfg, monkeypatch):
    """BASELINE config 3 (4096 streams x 256 frames, VBR -V50, block switching) and one GPU's share of config 5 (32 / 44.1 / 48 kHz
    by stream, CBR-128, correlation cycled) at their full size in one call on k_alloc_slim, the kernel bench.py runs them on:
    48 distinct signals rotated in time (86 rotation classes); ALL 4096 streams byte for byte against the oracle (a process per CPU:
    half a minute), every stream's status word"""
    monkeypatch.setenv("HMP3AMD_K6", "slim")
    a = api()
    S, F, U = 4096, 256, 48
    classes = [(dict(), 44100)] if cfg == "config3" else [(dict(bitrate=64, samprate=32000), 32000), (dict(bitrate=64), 44100), (dict(bitrate=64, samprate=48000), 48000)]
    base = [synth.stream_pcm(9950 + u, F, sr=classes[u % len(classes)][1], rho=RHOS[u % 4], bursts=True) for u in range(U)]
    pcm = np.stack([np.roll(base[i % U], 1152 * 5 * (i // U), axis=0) for i in range(S)])
    ctl = [a.default_control(**classes[(i % U) % len(classes)][0]) for i in range(S)]
    b = a.Batch(ctl if len(classes) > 1 else ctl[0], nstreams=S, max_frames=F)
    assert b.k6_variant() == 1
    got = b.encode_host(pcm)
    assert b.status() == 0
    from oracle_pool import oracle_bytes_many
    ids = list(range(S))
    want = oracle_bytes_many([classes[(s % U) % len(classes)][0] for s in range(S)], pcm, F, ids)
    bad = [s for s in ids if got[s] != want[s]]
    assert not bad, "%d of %d streams differ from the oracle, first: %s" % (len(bad), len(ids), bad[:8])
    b.close()


def test_ragged_calls_equal_one_shot():
    """state carry across calls: 1 + 7 + 24 frames in three calls == 32 frames in one call"""
    kw = CONFIGS["vbr50_sw"]       # block switching on: the carry includes detector and block-type state
    S, F = 5, 32
    pcm = np.stack([synth.stream_pcm(300 + i, F, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    one = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    ref = one.encode_host(pcm)
    one.close()
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=24)
    parts = [b"" for _ in range(S)]
    pos = 0
    for n in (1, 7, 24):
        out = b.encode_host(pcm[:, pos * 1152:(pos + n) * 1152])
        pos += n
        for s in range(S):
            parts[s] += out[s]
    assert parts == ref
    b.close()


def test_mixed_configuration_classes_in_one_batch():
    """per-stream E_CONTROL: 32 / 44.1 / 48 kHz and CBR / VBR side by side (config 5 style)"""
    a = api()
    kws = [dict(bitrate=64, samprate=32000), dict(bitrate=64), dict(bitrate=64, samprate=48000), dict(),
           dict(bitrate=96, mode=0, short_block_threshold=99999)]
    S, F = 10, 24
    pcm = np.stack([synth.stream_pcm(400 + i, F, sr=kws[i % 5].get("samprate", 44100), rho=RHOS[i % 4], bursts=True) for i in range(S)])
    b = a.Batch([a.default_control(**kws[i % 5]) for i in range(S)], max_frames=F)
    got = b.encode_host(pcm)
    for s in range(S):
        assert got[s] == oracle_bytes(kws[s % 5], pcm[s], F), "stream %d" % s
    b.close()


def test_edge_inputs_silence_fullscale_single_stream():
    kw = CONFIGS["cbr128"]
    F = 16
    z = np.zeros((F * 1152, 2), dtype=np.int16)
    sq = np.where((np.arange(F * 1152) // 24) % 2 == 0, 32767, -32768).astype(np.int16)
    fs = np.stack([sq, (-sq.astype(np.int32) - 1).astype(np.int16)], axis=1)
    imp = z.copy(); imp[5000, 0] = 32767; imp[9000, 1] = -32768
    pcm = np.stack([z, fs, imp])
    b = api().Batch(api().default_control(**kw), nstreams=3, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    for s in range(3):
        assert got[s] == oracle_bytes(kw, pcm[s], F), "edge stream %d" % s
    b.close()


@pytest.mark.parametrize("name,kw", [
    ("cbr96_lowrate", dict(bitrate=48)),                       # starved: decrease_bits / limit_bits / part2_3 cap
    ("cbr320_highrate", dict(bitrate=160)),                    # rich: increase_bits, large quantised values, linbits
    ("vbr150_hf", dict(vbr_mnr=150, hf_flag=3, freq_limit=22000)),
    ("vbr0", dict(vbr_mnr=0)),
    ("cbr128_lr", dict(bitrate=64, mode=0)),
])
def test_stress_signals_rare_paths(name, kw):
    """signals built to reach the allocator's rare branches: full-scale white noise (bit starvation, the
    4021-bit part2_3 cap), a full-scale low tone (quantised values beyond the 256-entry x^(4/3) table ->
    pow() path, escape codes), Nyquist alternation, DC offset, clicks in silence, hard-panned and
    anti-phase channels, and a stream that changes character every few frames"""
    F = 36
    n = F * 1152
    rng = np.random.default_rng(77)
    t = np.arange(n)
    noise = rng.integers(-32768, 32768, (n, 2)).astype(np.int16)
    tone = np.round(32767 * np.sin(2 * np.pi * 110.0 * t / 44100.0)).astype(np.int16)
    lowtone = np.stack([tone, tone], axis=1)
    nyq = np.stack([np.where(t % 2 == 0, 32767, -32768), np.where(t % 2 == 0, -20000, 20000)], axis=1).astype(np.int16)
    dc = np.stack([np.full(n, 12000), np.full(n, -32768)], axis=1).astype(np.int16)
    clicks = np.zeros((n, 2), dtype=np.int16); clicks[::4001, 0] = 32767; clicks[1000::5003, 1] = -32768
    panned = np.stack([tone, np.zeros(n, dtype=np.int16)], axis=1)
    anti = np.stack([noise[:, 0] // 2, -(noise[:, 0] // 2)], axis=1).astype(np.int16)
    seg = np.concatenate([x[i * 4608:(i + 1) * 4608] for i, x in enumerate([noise, lowtone, clicks, nyq, anti, dc, panned, noise, lowtone] * 2)])[:n]
    pcm = np.stack([noise, lowtone, nyq, dc, clicks, panned, anti, seg])
    S = pcm.shape[0]
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    for s in range(S):
        assert got[s] == oracle_bytes(kw, pcm[s], F), "%s: stress stream %d" % (name, s)
    b.close()


@pytest.mark.parametrize("kw", [dict(bitrate=64), dict(vbr_mnr=80), dict(bitrate=56, samprate=32000)], ids=["cbr128", "vbr80", "cbr112_32k"])
def test_long_run_many_ragged_calls(kw):
    """1500 frames through calls of 1..37 frames, s16 and fp32 calls interleaved: the pending-frame carry
    between calls, the CBR padding sequence and the reservoir never drift from the oracle"""
    sr = kw.get("samprate", 44100)
    F = 1500
    pcm = np.stack([synth.stream_pcm(40 + i, F, sr=sr, rho=RHOS[i % 4], bursts=(i & 1) == 1) for i in range(3)])
    b = api().Batch(api().default_control(**kw), nstreams=3, max_frames=37)
    rng = np.random.default_rng(3)
    got = [b"", b"", b""]
    f = 0
    while f < F:
        n = int(min(F - f, rng.integers(1, 38)))
        chunk = pcm[:, f * 1152:(f + n) * 1152]
        out = b.encode_host(chunk.astype(np.float32) if (f // 7) & 1 else chunk)
        for s in range(3):
            got[s] += out[s]
        f += n
    assert b.status() == 0
    for s in range(3):
        assert got[s] == oracle_bytes(kw, pcm[s], F), "stream %d" % s
    fb = [b.frames_bytes(s) for s in range(3)]
    assert all(x[1] == len(got[i]) and F - 12 <= x[0] <= F for i, x in enumerate(fb))
    b.close()


@pytest.mark.parametrize("kw", [dict(), dict(bitrate=64, mode=0), dict(samprate=22050, bitrate=32), dict(samprate=16000, mode=2, bitrate=16)],
                         ids=["vbr50_sw", "cbr128_lr_sw", "lsf_cbr64_22k", "a1_dual_16k"])
def test_one_call_longer_than_the_flag_staging_block(kw):
    """300 frames in ONE call: the stream walk stages the granules' block types and stereo decisions in LDS 128 frames at
    a time (hx_alloc3.inc), so a call this long crosses two refills - with block switching and changing M/S decisions on
    both sides of them, for the MPEG-1, MPEG-2 and first-generation kernels"""
    sr = kw.get("samprate", 44100)
    S, F = 4, 300
    pcm = np.stack([synth.stream_pcm(900 + i, F, sr=sr, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    for s in range(S):
        assert got[s] == oracle_bytes(kw, pcm[s], F), "stream %d" % s
    b.close()


@pytest.mark.parametrize("kw", [dict(bitrate=64), dict(bitrate=64, mode=3)], ids=["stereo", "mono"])
def test_device_pcm_pointer_without_16_byte_alignment(kw):
    """k_polyphase stages the PCM with 16-byte loads when the caller's pointer allows it and sample by sample when it
    does not: a device buffer that starts 4 (stereo) / 2 (mono) bytes past an aligned address takes the second way"""
    import torch
    mono = kw.get("mode") == 3
    S, F = 3, 9
    pcm = np.stack([synth.stream_pcm(70 + i, F, rho=RHOS[i % 4], bursts=True) for i in range(S)])
    if mono:
        pcm = np.ascontiguousarray(pcm[:, :, 0])
    dev = torch.device("cuda:0")
    flat = torch.zeros(pcm.size + 8, dtype=torch.int16, device=dev)
    off = 1 if mono else 2                      # int16 elements
    flat[off:off + pcm.size] = torch.from_numpy(pcm.reshape(-1)).to(dev)
    assert (flat.data_ptr() + 2 * off) % 16 != 0
    b = api().Batch(api().default_control(**kw), nstreams=S, max_frames=F)
    stride = b.out_stride(F)
    d_out = torch.zeros((S, stride), dtype=torch.uint8, device=dev); d_nb = torch.zeros((S,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    b.encode_device(flat.data_ptr() + 2 * off, F, d_out.data_ptr(), stride, d_nb.data_ptr(), None)
    torch.cuda.synchronize()
    assert b.status() == 0
    o, nb = d_out.cpu().numpy(), d_nb.cpu().numpy()
    for s in range(S):
        assert o[s, :nb[s]].tobytes() == oracle_bytes(kw, pcm[s], F), "stream %d" % s
    b.close()


def test_api_misuse_fails_loudly_and_leaves_the_batch_usable():
    a = api()
    with pytest.raises(RuntimeError):           # mono and stereo streams cannot share a batch
        a.Batch([a.default_control(bitrate=64), a.default_control(bitrate=64, mode=3)], nstreams=2, max_frames=4)
    with pytest.raises(RuntimeError):           # streams of the two allocator generations cannot share a batch
        a.Batch([a.default_control(bitrate=64), a.default_control(bitrate=64, mode=2)], nstreams=2, max_frames=4)
    with pytest.raises(RuntimeError):           # CBR below 48 kbps per channel at an MPEG-1 rate: rejected like the reference does
        a.Batch(a.default_control(bitrate=32), nstreams=1, max_frames=4)
    b = a.Batch(a.default_control(bitrate=64), nstreams=2, max_frames=4)
    pcm = np.stack([synth.stream_pcm(5, 8), synth.stream_pcm(6, 8)])
    with pytest.raises(RuntimeError):           # more frames than the batch was created for
        b.encode_host(pcm)
    import torch
    dev = torch.device("cuda:0")
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm[:, :4 * 1152])).to(dev)
    d_out = torch.zeros((2, 4096), dtype=torch.uint8, device=dev)
    d_nb = torch.zeros((2,), dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="out_stride"):      # an output buffer smaller than hx_batch_out_stride asks for
        b.encode_device(d_pcm.data_ptr(), 4, d_out.data_ptr(), 256, d_nb.data_ptr(), None)
    with pytest.raises(RuntimeError, match="null"):
        b.encode_device(d_pcm.data_ptr(), 4, None, b.out_stride(4), d_nb.data_ptr(), None)
    first = b.encode_host(pcm[:, :4 * 1152])    # the failed calls consumed nothing
    second = b.encode_host(pcm[:, 4 * 1152:])
    for s in range(2):
        assert first[s] + second[s] == oracle_bytes(dict(bitrate=64), pcm[s], 8)
    b.close()


def test_cmp3enc_surface_single_stream():
    """the CMp3Enc-compatible entry points: init return values, per-frame encode, getters"""
    a = api()
    kw = CONFIGS["cbr128"]
    F = 12
    pcm = synth.stream_pcm(500, F)
    e = a.Mp3Enc()
    assert e.MP3_audio_encode_init(a.default_control(**kw), 16, 0) == 4612     # 1153 sample frames: what a call needs buffered (mp3enc.cpp:2655)
    out = []
    for f in range(F):
        nin, bs = e.MP3_audio_encode(pcm[f * 1152:(f + 1) * 1152])
        assert nin == 4608
        out.append(bs)
    assert out[0] == b""                       # 2-granule look-ahead: the first call emits nothing
    assert b"".join(out) == oracle_bytes(kw, pcm, F)
    assert e.L3_audio_encode_get_frames() == sum(1 for _ in range(len(b"".join(out)) // 417))  # 417/418-byte frames
    ec = e.L3_audio_encode_info_ec()
    assert ec.bitrate == 64 and ec.samprate == 44100 and ec.nsb_limit == 23
    h = e.L3_audio_encode_info_head()
    assert h.id == 1 and h.option == 1 and h.br_index == 9 and h.mode == 1
    assert "Layer III" in e.L3_audio_encode_info_string()
    assert abs(e.L3_audio_encode_get_bitrate_float() - 128.0) < 3.0
    # float entry point, re-init on the same object (mp3enc.cpp:267-272)
    assert e.L3_audio_encode_init(a.default_control(**kw)) == 9216
    out2 = [e.L3_audio_encode(pcm[f * 1152:(f + 1) * 1152].astype(np.float32))[1] for f in range(F)]
    assert b"".join(out2) == b"".join(out)
    # rejected configurations return 0 like the reference
    assert e.L3_audio_encode_init(a.default_control(bitrate=40)) == 0
    e.close()


@pytest.mark.one_k6_build
def test_per_frame_graph_calls_never_hand_back_stale_results():
    """hx_enc_* replays one HIP graph per call and picks the call's results up from page-locked host memory, where the packing
    workgroup puts them with a sequence word behind system-scope fences.  (Round 6's first form polled the last of the graph's
    device-to-host copy nodes and read what the copy ahead of it had brought down: one call in 20 000 saw a stale byte count
    or stale bytes - copies complete in order on the device, their writes do not land in order in host memory.)  200 encoders x
    40 calls of an MPEG-2 mono VBR stream (two frames per call, sizes that change from call to call, so a stale byte count
    shows) against the oracle, every call's bytes."""
    a = api()
    kw = dict(samprate=24000, mode=3, vbr_mnr=15, short_block_threshold=2000)
    F = 40
    pcm = synth.stream_pcm(377305, F, sr=24000, rho=0.0, bursts=True)[:, 0].copy()
    enc = O.OracleEncoder(O.default_control(**kw))
    want = [enc.encode_f32(pcm[f * 1152:(f + 1) * 1152].astype(np.float32)) for f in range(F)]
    assert len(set(len(w) for w in want)) > 4           # the byte count moves from call to call
    for rep in range(200):
        e = a.Mp3Enc()
        assert e.L3_audio_encode_init(a.default_control(**kw)) == 4608
        got = [e.L3_audio_encode(pcm[f * 1152:(f + 1) * 1152].astype(np.float32))[1] for f in range(F)]
        e.close()
        assert got == want, "encoder %d: call %d differs" % (rep, next(i for i in range(F) if got[i] != want[i]))


def test_full_size_config2_properties():
    """BASELINE configs[1] at full size (1024 streams x 256 frames): frame structure, padding
    sequence, determinism, and byte equality with the oracle on a random subset of streams."""
    a = api()
    kw = CONFIGS["cbr128"]
    S, F = 1024, 256
    # 1024 distinct signals, synthesised on the GPU by the bench's own generator (the host one takes 0.2 s per stream)
    import torch
    import bench
    pcm = bench.synth_batch_gpu(torch, np, S, F, [44100] * S, [0.7] * S, False, torch.device("cuda:0")).cpu().numpy()
    b = a.Batch(a.default_control(**kw), nstreams=S, max_frames=F)
    got = b.encode_host(pcm)
    assert b.status() == 0
    # CBR-128 @ 44.1 kHz: every frame 417 or 418 bytes, 255 frames out after 256 calls, header valid
    for s in range(0, S, 37):
        bs = got[s]
        pos, n, pads = 0, 0, []
        while pos < len(bs):
            assert bs[pos] == 0xFF and bs[pos + 1] == 0xFB and (bs[pos + 2] >> 4) == 9
            pad = (bs[pos + 2] >> 1) & 1
            pads.append(pad)
            pos += 417 + pad
            n += 1
        # the first call emits nothing and the bit reservoir may hold the newest frame(s) back
        assert pos == len(bs) and F - 3 <= n <= F - 1
        # padding: 417.96 bytes/frame -> 49 of every 50 frames padded, from the slot counter
        assert abs(sum(pads) / len(pads) - (144000 * 128 % 44100) / 44100.0) < 0.02
    assert len(set(got)) == S      # distinct signals give distinct streams
    # every one of the 1024 streams byte for byte against the oracle (a process per CPU: seconds)
    from oracle_pool import oracle_bytes_many
    want = oracle_bytes_many(kw, pcm, F)
    bad = [s for s in range(S) if got[s] != want[s]]
    assert not bad, "%d of %d streams differ from the oracle, first: %s" % (len(bad), S, bad[:8])
    # determinism: a second batch object gives the same bytes
    b2 = a.Batch(a.default_control(**kw), nstreams=S, max_frames=F)
    again = b2.encode_host(pcm)
    assert again == got
    b.close(); b2.close()
