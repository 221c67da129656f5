"""Xing / Info / LAME tag frame: the product's host code (hmp3_amd/csrc/hx_xhead.cpp,
through the C ABI) against the reference's own xhead.c compiled into oracle/_ref, on randomised
encode histories.  CPU only; skipped where the reference build is absent.  The file-level check
(whole .mp3 byte-identical to the reference CLI) is tests/test_gpu_parity.py::test_cli_*."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built")


def product():
    from hmp3_amd import api
    L = api.lib()
    L.hx_xing_create.restype = C.c_void_p
    L.hx_xing_destroy.argtypes = [C.c_void_p]
    L.hx_xing_header.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p] * 4 + [C.c_int]
    L.hx_xing_toc.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.hx_xing_update_info.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_ulonglong, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_ushort]
    L.hx_xing_update_crc.argtypes = [C.c_ushort, C.c_void_p, C.c_int]
    L.hx_xing_update_crc.restype = C.c_ushort
    return L


def reference():
    R = O.ref()
    R.XingHeader.argtypes = [C.c_int] * 8 + [C.c_void_p] * 4 + [C.c_int]
    R.XingHeaderTOC.argtypes = [C.c_int, C.c_int]
    R.XingHeaderUpdateInfo.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_ulonglong, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_ushort]
    R.XingHeaderUpdateCRC.argtypes = [C.c_ushort, C.c_void_p, C.c_int]
    R.XingHeaderUpdateCRC.restype = C.c_ushort
    return R


CASES = [
    # samprate, mode, flags, vbr_scale, kbps, frames, frame bytes range
    (44100, 1, 0x4F, -1, 128, 2302, (417, 418)),       # CLI default, CBR-128: "Info" + TOC + LAME fields
    (44100, 1, 0x4F, 50, 128, 2302, (104, 1044)),      # VBR: "Xing"
    (48000, 1, 0x4F, 100, 128, 9000, (96, 960)),
    (32000, 0, 0x4F, -1, 160, 700, (720, 720)),
    (44100, 1, 0x4F, -1, 48, 300, (156, 157)),         # CBR below 64 kbps: the TOC is dropped
    (44100, 1, 0x0B, 50, 128, 120, (104, 1044)),       # -X1: no TOC, no info tag
    (22050, 3, 0x4F, -1, 64, 100, (208, 209)),         # MPEG-2 mono layout of the tag frame
]


@pytest.mark.parametrize("case", CASES, ids=[str(c[:5]) for c in CASES])
def test_tag_frame_matches_reference(case):
    sr, mode, flags, scale, kbps, nframes, (lo, hi) = case
    P, R = product(), reference()
    rng = np.random.default_rng(sr + nframes)
    x = P.hx_xing_create()
    a = (C.c_ubyte * 2048)()
    b = (C.c_ubyte * 2048)()
    na = R.XingHeader(sr, mode, 1, 1, flags, 0, 0, scale, None, a, None, None, kbps)
    nb = P.hx_xing_header(x, sr, mode, 1, 1, flags, 0, 0, scale, None, b, None, None, kbps)
    assert na == nb and na > 0
    assert bytes(a[:na]) == bytes(b[:nb])
    # an encode history: seek points on the cadence the functions ask for, MusicCRC over the payload
    frames, nbytes, counter = 0, 0, 0
    crc_a = crc_b = 0
    for _ in range(nframes):
        fb = int(rng.integers(lo, hi + 1))
        payload = rng.integers(0, 256, fb, dtype=np.uint8)
        crc_a = R.XingHeaderUpdateCRC(crc_a, payload.ctypes.data, fb)
        crc_b = P.hx_xing_update_crc(crc_b, payload.ctypes.data, fb)
        frames += 1
        nbytes += fb
        counter -= 1
        if counter <= 0:
            ca = R.XingHeaderTOC(frames + 1, nbytes + na)
            cb = P.hx_xing_toc(x, frames + 1, nbytes + na)
            assert ca == cb
            counter = ca
    assert crc_a == crc_b
    samples = nframes * 1152 - 1680 - int(rng.integers(0, 1152))
    ra = R.XingHeaderUpdateInfo(frames, nbytes + na, scale, None, a, None, None, samples, nbytes + na, 16000, sr, sr, crc_a)
    rb = P.hx_xing_update_info(x, frames, nbytes + na, scale, None, b, None, None, samples, nbytes + na, 16000, sr, sr, crc_b)
    assert ra == rb == 1
    assert bytes(a[:na]) == bytes(b[:nb])
    P.hx_xing_destroy(x)


def test_tag_does_not_fit_or_unknown_rate():
    P, R = product(), reference()
    x = P.hx_xing_create()
    a = (C.c_ubyte * 2048)()
    for sr, kbps in [(44100, 32), (12345, 128), (44100, 999)]:
        assert R.XingHeader(sr, 1, 1, 1, 0x4F, 0, 0, -1, None, a, None, None, kbps) == \
            P.hx_xing_header(x, sr, 1, 1, 1, 0x4F, 0, 0, -1, None, a, None, None, kbps)
    P.hx_xing_destroy(x)
