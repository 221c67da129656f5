"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present, the real
reference (oracle/_ref/libhmp3ref.so).  TEST INFRASTRUCTURE ONLY - see oracle/hxo.h."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Control(C.Structure):
    """E_CONTROL (pub/encapp.h:42-72)"""
    _fields_ = [(n, C.c_int) for n in (
        "mode", "bitrate", "samprate", "nsbstereo", "filter_select", "freq_limit", "nsb_limit",
        "layer", "cr_bit", "original", "hf_flag", "vbr_flag", "vbr_mnr", "vbr_br_limit",
        "vbr_delta_mnr", "chan_add_f0", "chan_add_f1", "sparse_scale")] + \
        [("mnr_adjust", C.c_int * 21)] + \
        [(n, C.c_int) for n in ("cpu_select", "quick", "test1", "test2", "test3", "short_block_threshold")]


def default_control(**kw):
    """CLI defaults (test/tomp3.cpp:357-384); bitrate=N (per channel) selects CBR like -B N"""
    ec = Control()
    ec.mode = 1; ec.bitrate = -1; ec.samprate = 44100; ec.nsbstereo = -1; ec.filter_select = -1
    ec.freq_limit = 24000; ec.nsb_limit = -1; ec.layer = 3; ec.cr_bit = 1; ec.original = 1
    ec.hf_flag = 0; ec.vbr_flag = 1; ec.vbr_mnr = 50; ec.vbr_br_limit = 160; ec.vbr_delta_mnr = 0
    ec.chan_add_f0 = ec.chan_add_f1 = 24000; ec.sparse_scale = -1; ec.cpu_select = 0
    ec.quick = -1; ec.test1 = -1; ec.test2 = ec.test3 = 0; ec.short_block_threshold = 700
    for k, v in kw.items():
        setattr(ec, k, v)
    if "bitrate" in kw and kw["bitrate"] > 0 and "vbr_flag" not in kw:
        ec.vbr_flag = 0
    return ec


def _load(path):
    return C.CDLL(path) if os.path.exists(path) else None


_lib = None
_ref = None
_ref_zero = None


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(p):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle`")
        _lib = C.CDLL(p)
        _lib.hxo_new.restype = C.c_void_p
        _lib.hxo_free.argtypes = [C.c_void_p]
        _lib.hxo_init.argtypes = [C.c_void_p, C.POINTER(Control)]
        _lib.hxo_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.hxo_encode_frame_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.hxo_encode_frame_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.hxo_mblog.argtypes = [C.c_float]
        _lib.hxo_mbexp.argtypes = [C.c_int]
        _lib.hxo_mbexp.restype = C.c_float
        _lib.hxo_frames_out.argtypes = [C.c_void_p]
        _lib.hxo_frames_out.restype = C.c_uint
    return _lib


def ref():
    """the real reference, or None when oracle/_ref has not been built"""
    global _ref
    if _ref is None:
        _ref = _load(os.path.join(HERE, "_ref", "libhmp3ref.so"))
        if _ref is not None:
            _ref.ref_new.restype = C.c_void_p
            _ref.ref_free.argtypes = [C.c_void_p]
            _ref.ref_init.argtypes = [C.c_void_p, C.POINTER(Control)]
            _ref.ref_init_s16.argtypes = [C.c_void_p, C.POINTER(Control)]
            _ref.ref_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            _ref.ref_encode_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            _ref.ref_encode_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            _ref.ref_encode_stream_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
            _ref.ref_encode_stream_s16.restype = C.c_long
            _ref.ref_dump.argtypes = [C.c_void_p, C.c_void_p]
            _ref.ref_mbLogC.argtypes = [C.c_float]
            _ref.ref_mbExp.argtypes = [C.c_int]
            _ref.ref_mbExp.restype = C.c_float
    return _ref


def ref_zero():
    """the reference compiled with every uninitialised local defined as zero (make -C oracle ref_zero), or None"""
    global _ref_zero
    if _ref_zero is None:
        _ref_zero = _load(os.path.join(HERE, "_ref", "libhmp3ref_zero.so"))
        if _ref_zero is not None:
            _ref_zero.ref_new.restype = C.c_void_p
            _ref_zero.ref_free.argtypes = [C.c_void_p]
            _ref_zero.ref_init.argtypes = [C.c_void_p, C.POINTER(Control)]
            _ref_zero.ref_init_s16.argtypes = [C.c_void_p, C.POINTER(Control)]
            _ref_zero.ref_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            _ref_zero.ref_encode_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return _ref_zero


class RefDump(C.Structure):
    _fields_ = [
        ("sample", C.c_float * (2 * 4 * 576)), ("xr", C.c_float * (2 * 2 * 576)),
        ("sig_mask", C.c_float * (2 * 36 * 2)), ("ix", C.c_int * (2 * 576)),
        ("signx", C.c_ubyte * (2 * 576)), ("sf_l", C.c_int * (2 * 2 * 23)),
        ("gr", C.c_int * (2 * 2 * 26)), ("scfsi", C.c_int * 2), ("attack_buf", C.c_int * 64),
        ("ecsave", C.c_float * 128), ("igrx", C.c_int), ("byte_pool", C.c_int), ("MNR", C.c_int),
        ("PoolFraction", C.c_int), ("call_count", C.c_int), ("ms_correlation_memory", C.c_int),
        ("NTadjust", C.c_int * 44), ("nsb_limit", C.c_int), ("nsb_limitMS", C.c_int * 2),
        ("band_limit", C.c_int), ("AveTargetBits", C.c_int), ("main_framebytes", C.c_int),
        ("initialMNR", C.c_int), ("nsf", C.c_int * 2)]


class OracleEncoder:
    """one stream through the restatement"""

    def __init__(self, ec):
        self.l = lib()
        self.h = self.l.hxo_new()
        self.bytes_in = self.l.hxo_init(self.h, C.byref(ec))
        self.out = (C.c_ubyte * 16384)()

    def ok(self):
        return self.bytes_in != 0

    def encode_s16(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.int16)
        n = self.l.hxo_encode_frame_s16(self.h, frame.ctypes.data, self.out)
        return bytes(self.out[:n])

    def encode_f32(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        n = self.l.hxo_encode_frame(self.h, frame.ctypes.data, self.out)
        return bytes(self.out[:n])

    def encode_packet(self, frame):
        """-> (bitstream bytes, reformatted packet bytes) of L3_audio_encode_Packet"""
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        pk = (C.c_ubyte * 4096)()
        nb = (C.c_int * 2)()
        n = self.l.hxo_encode_frame_packet(self.h, frame.ctypes.data, self.out, pk, nb)
        self.packet_sizes = (nb[0], nb[1])      # MPEG-2: two single-granule packets back to back
        return bytes(self.out[:n]), bytes(pk[:nb[0] + nb[1]])

    def __del__(self):
        try:
            self.l.hxo_free(self.h)
        except Exception:
            pass


class RefEncoder:
    """one stream through the real reference (None-safe: check oracle.ref() first)"""

    def __init__(self, ec, s16=True, zero_locals=False):
        self.r = ref_zero() if zero_locals else ref()
        self.h = self.r.ref_new()
        self.s16 = s16
        self.bytes_in = (self.r.ref_init_s16 if s16 else self.r.ref_init)(self.h, C.byref(ec))
        self.out = (C.c_ubyte * 16384)()

    def encode_s16(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.int16)
        n = self.r.ref_encode_s16(self.h, frame.ctypes.data, self.out)
        return bytes(self.out[:n])

    def encode_f32(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        n = self.r.ref_encode(self.h, frame.ctypes.data, self.out)
        return bytes(self.out[:n])

    def encode_packet(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        pk = (C.c_ubyte * 4096)()
        nb = (C.c_int * 2)()
        n = self.r.ref_encode_packet(self.h, frame.ctypes.data, self.out, pk, nb)
        self.packet_sizes = (nb[0], nb[1])
        return bytes(self.out[:n]), bytes(pk[:nb[0] + nb[1]])

    def dump(self):
        d = RefDump()
        self.r.ref_dump(self.h, C.byref(d))
        return d

    def __del__(self):
        try:
            self.r.ref_free(self.h)
        except Exception:
            pass


def encode_stream(enc, pcm_i16, flush_frames=2):
    """encode [n*1152, 2] int16 through enc, then feed zero frames; returns bytes"""
    nfr = pcm_i16.shape[0] // 1152
    out = []
    for f in range(nfr):
        out.append(enc.encode_s16(pcm_i16[f * 1152:(f + 1) * 1152]))
    z = np.zeros((1152, 2), dtype=np.int16)
    for _ in range(flush_frames):
        out.append(enc.encode_s16(z))
    return b"".join(out)


GR_FIELDS = ["part2_3_length", "big_values", "global_gain", "scalefac_compress", "window_switching_flag",
             "block_type", "mixed_block_flag", "table_select0", "table_select1", "table_select2",
             "subblock_gain0", "subblock_gain1", "subblock_gain2", "region0_count", "region1_count",
             "preflag", "scalefac_scale", "count1table_select", "aux_nquads", "aux_bits", "aux_not_null",
             "aux_nreg0", "aux_nreg1", "aux_nreg2"]


class FrameDebug(C.Structure):
    """hxo_frame_debug (oracle/hxo.h)"""
    _fields_ = [
        ("sample_new", C.c_float * (2 * 2 * 576)), ("xr_pre", C.c_float * (2 * 2 * 576)),
        ("etab", C.c_float * (2 * 2 * 64)), ("thr", C.c_float * (2 * 2 * 64)), ("mask", C.c_float * (2 * 2 * 22)),
        ("block_type", C.c_int * 2), ("attack", C.c_int * 4),
        ("ms", C.c_int), ("ms_metric", C.c_int * 2), ("byte_pool", C.c_int), ("MNR_after", C.c_int),
        ("gr", C.c_int * (2 * 2 * 27)), ("sf", C.c_int * (2 * 2 * 22)), ("ix", C.c_int * (2 * 2 * 576)),
        ("signx", C.c_ubyte * (2 * 2 * 576)), ("scfsi", C.c_int * 2), ("main_bytes", C.c_int)]


def oracle_enable_debug(enc):
    """attach a FrameDebug record to an OracleEncoder; returns it (refilled on every frame)"""
    l = lib()
    l.hxo_sizeof_frame_debug.restype = C.c_int
    assert l.hxo_sizeof_frame_debug() == C.sizeof(FrameDebug), (l.hxo_sizeof_frame_debug(), C.sizeof(FrameDebug))
    d = FrameDebug()
    l.hxo_set_debug.argtypes = [C.c_void_p, C.c_void_p]
    l.hxo_set_debug(enc.h, C.byref(d))
    enc._dbg = d
    return d
