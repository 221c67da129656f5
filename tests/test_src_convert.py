"""Sample-format / sample-rate converter in front of the encoder: the product's host code
(hmp3_amd/csrc/hx_src.cpp, through the C ABI of libhmp3amd.so) against the reference's own Csrc compiled into
oracle/_ref, bit for bit, over consecutive calls (the filter phase and the two-stage buffer carry over).
CPU only; skipped where the reference build is absent."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built")

# (source rate, target rate): every case of the converter
PAIRS = [(44100, 44100),                    # case 0: copy
         (11025, 22050), (8000, 16000),     # case 1: exactly 1:2
         (12000, 16000), (11025, 16000), (32000, 44100), (22050, 24000),   # case 2: linear interpolation
         (48000, 24000), (44100, 22050), (48000, 32000), (24000, 16000),   # case 3: small polyphase bank
         (44100, 32000), (48000, 44100), (44100, 24000), (32000, 22050)]   # case 4: two stages


def libs():
    from hmp3_amd import api
    P, R = api.lib(), O.ref()
    P.hx_src_create.restype = C.c_void_p
    P.hx_src_destroy.argtypes = [C.c_void_p]
    P.hx_src_init.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
    P.hx_src_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_src_new.restype = C.c_void_p
    R.ref_src_free.argtypes = [C.c_void_p]
    R.ref_src_init.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
    R.ref_src_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return P, R


def source_bytes(rng, n, channels, bits, is_float):
    if is_float:
        return rng.uniform(-1.0, 1.0, n * channels).astype("<f4").tobytes()
    if bits == 8:
        return rng.integers(0, 256, n * channels, dtype=np.uint8).tobytes()
    if bits == 16:
        return rng.integers(-32768, 32768, n * channels).astype("<i2").tobytes()
    if bits == 24:
        v = rng.integers(-(1 << 23), 1 << 23, n * channels).astype("<i4").tobytes()
        return b"".join(v[i:i + 3] for i in range(0, len(v), 4))
    return rng.integers(-(1 << 31), 1 << 31, n * channels).astype("<i4").tobytes()


@pytest.mark.parametrize("source,target", PAIRS, ids=["%d_%d" % p for p in PAIRS])
@pytest.mark.parametrize("channels,target_channels", [(1, 1), (2, 2), (2, 1)], ids=["mono", "stereo", "downmix"])
def test_converter_bit_identical_to_reference(source, target, channels, target_channels):
    P, R = libs()
    rng = np.random.default_rng(source + 7 * target + channels)
    for bits, is_float in ((16, 0), (24, 0), (32, 1), (8, 0), (32, 0)):
        hp, hr = P.hx_src_create(), R.ref_src_new()
        ca, cb = C.c_int(0), C.c_int(0)
        na = R.ref_src_init(hr, source, channels, bits, is_float, target, target_channels, C.byref(ca))
        nb = P.hx_src_init(hp, source, channels, bits, is_float, target, target_channels, C.byref(cb))
        assert na == nb and na > 0 and ca.value == cb.value
        ncalls = 12 if bits == 16 else 3
        bpf = channels * bits // 8
        need = 1152 * (source // target + 2) + 256          # frames a call may stage
        data = source_bytes(rng, need * (ncalls + 1), channels, bits, is_float)
        pos = 0
        for _ in range(ncalls):
            chunk = (C.c_ubyte * (need * bpf)).from_buffer_copy(data[pos:pos + need * bpf])
            ya = np.zeros(2304, np.float32); yb = np.zeros(2304, np.float32)
            oa, ob = C.c_int(0), C.c_int(0)
            ia = R.ref_src_convert(hr, chunk, ya.ctypes.data, C.byref(oa))
            ib = P.hx_src_convert(hp, chunk, yb.ctypes.data, C.byref(ob))
            assert ia == ib and oa.value == ob.value == 4 * 1152 * target_channels
            assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)), (bits, is_float)
            assert ia % bpf == 0 and 0 < ia <= na
            pos += ia
        P.hx_src_destroy(hp); R.ref_src_free(hr)


def test_converter_rejects_what_the_reference_rejects():
    P, R = libs()
    for args in [(44100, 2, 12, 0, 44100, 2), (44100, 3, 16, 0, 44100, 2), (4000, 2, 16, 0, 16000, 2), (44100, 2, 16, 1, 44100, 2),
                 (48000, 2, 16, 0, 4000, 2), (47999, 2, 16, 0, 32000, 2)]:
        hp, hr = P.hx_src_create(), R.ref_src_new()
        ca, cb = C.c_int(0), C.c_int(0)
        assert R.ref_src_init(hr, *args, C.byref(ca)) == P.hx_src_init(hp, *args, C.byref(cb)), args
        P.hx_src_destroy(hp); R.ref_src_free(hr)
