"""Python binding of the C ABI in include/hmp3_amd.h (hmp3_amd/libhmp3amd.so).

The library is the product: hand-written HIP kernels for gfx950 behind a C ABI.  This module
only loads it (ctypes) and mirrors the reference's CMp3Enc interface for tests / bench.  There
is no CPU fallback: if the library or a GPU is missing, calls raise.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# HMP3AMD_LIB selects another build of the same library (tools/: the -DHX_PROFILE build)
LIB_PATH = os.environ.get("HMP3AMD_LIB") or os.path.join(HERE, "libhmp3amd.so")


class EControl(C.Structure):
    """E_CONTROL (reference pub/encapp.h:42-72)"""
    _fields_ = [(n, C.c_int) for n in (
        "mode", "bitrate", "samprate", "nsbstereo", "filter_select", "freq_limit", "nsb_limit",
        "layer", "cr_bit", "original", "hf_flag", "vbr_flag", "vbr_mnr", "vbr_br_limit",
        "vbr_delta_mnr", "chan_add_f0", "chan_add_f1", "sparse_scale")] + \
        [("mnr_adjust", C.c_int * 21)] + \
        [(n, C.c_int) for n in ("cpu_select", "quick", "test1", "test2", "test3", "short_block_threshold")]


class MpegHead(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sync", "id", "option", "prot", "br_index", "sr_index", "pad",
                                       "private_bit", "mode", "mode_ext", "cr", "original", "emphasis")]


class InOut(C.Structure):
    _fields_ = [("in_bytes", C.c_int), ("out_bytes", C.c_int)]


class IntPair(C.Structure):
    _fields_ = [("a", C.c_int), ("b", C.c_int)]


EXPORTS = [
    "hx_last_error", "hx_device_count", "hx_default_control",
    "hx_enc_create", "hx_enc_destroy", "hx_enc_L3_audio_encode_init", "hx_enc_L3_audio_encode",
    "hx_enc_MP3_audio_encode_init", "hx_enc_MP3_audio_encode", "hx_enc_L3_audio_encode_Packet", "hx_enc_MP3_audio_encode_Packet", "hx_batch_packet_buffers", "hx_batch_frame_stats_buffer", "hx_batch_encode_f32_host_stats", "hx_control_info", "hx_enc_get_bitrate",
    "hx_enc_get_bitrate_float", "hx_enc_get_bitrate2_float", "hx_enc_get_frames",
    "hx_enc_get_frames_bytes", "hx_enc_info_ec", "hx_enc_info_head", "hx_enc_info_string",
    "hx_batch_create", "hx_batch_destroy", "hx_batch_nstreams", "hx_batch_out_stride",
    "hx_batch_reset_stream", "hx_batch_stream_state_bytes", "hx_batch_get_stream_state", "hx_batch_set_stream_state", "hx_src_create", "hx_src_destroy", "hx_src_init", "hx_src_convert",
    "hx_batch_submit_s16_device", "hx_batch_submit_f32_device", "hx_batch_wait", "hx_batch_set_gate",
    "hx_batch_submit_s16_host", "hx_batch_submit_f32_host", "hx_batch_wait_host", "hx_pinned_alloc", "hx_pinned_free",
    "hx_batch_encode_s16_device", "hx_batch_encode_s16_host", "hx_batch_encode_f32_device", "hx_batch_encode_f32_host",
    "hx_xing_create", "hx_xing_destroy", "hx_xing_header", "hx_xing_toc", "hx_xing_update_info", "hx_xing_update_crc", "hx_xing_bitrate_index", "hx_batch_status",
    "hx_batch_gate_timeouts", "hx_enc_out_stats", "hx_multi_create", "hx_multi_destroy", "hx_multi_ndevices", "hx_multi_nstreams", "hx_multi_shard", "hx_multi_batch",
    "hx_multi_out_stride", "hx_multi_encode_s16_host", "hx_multi_encode_f32_host", "hx_multi_encode_f32_host_stats", "hx_multi_status",
    "hx_build_id", "hx_batch_frames_bytes", "hx_batch_alloc_kernel_ms", "hx_batch_debug_read", "hx_batch_debug_enable", "hx_debug_host_table",
    "hx_batch_k6_variant", "hx_batch_resident_streams", "hx_debug_slim_tables_ok", "hx_libc_version", "hx_libm_spot_check", "hx_device_numa_node", "hx_bind_thread_to_device", "hx_bind_thread_to_node", "hx_refresh_process_cpus",
]

_lib = None


def lib():
    """load libhmp3amd.so (raises if it has not been built: run hmp3_amd/build.sh)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("hmp3_amd/libhmp3amd.so is missing - build it with hmp3_amd/build.sh "
                               "(there is no CPU fallback)")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64, and whichever copy is
        # mapped first serves both.  Import torch (when present) before our library so that tensors
        # and our kernels share a runtime; without torch the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.hx_last_error.restype = C.c_char_p
        L.hx_default_control.argtypes = [C.POINTER(EControl)]
        L.hx_enc_create.restype = C.c_void_p
        L.hx_enc_create.argtypes = [C.c_int]
        L.hx_enc_destroy.argtypes = [C.c_void_p]
        L.hx_enc_L3_audio_encode_init.argtypes = [C.c_void_p, C.POINTER(EControl)]
        L.hx_enc_L3_audio_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hx_enc_L3_audio_encode.restype = InOut
        L.hx_enc_L3_audio_encode_Packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hx_enc_L3_audio_encode_Packet.restype = InOut
        L.hx_enc_MP3_audio_encode_Packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hx_enc_MP3_audio_encode_Packet.restype = InOut
        L.hx_batch_packet_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.hx_batch_packet_buffers.restype = None
        L.hx_enc_MP3_audio_encode_init.argtypes = [C.c_void_p, C.POINTER(EControl), C.c_int, C.c_int, C.c_int, C.c_int]
        L.hx_enc_MP3_audio_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hx_enc_MP3_audio_encode.restype = InOut
        L.hx_enc_get_bitrate.argtypes = [C.c_void_p]
        L.hx_enc_get_bitrate_float.argtypes = [C.c_void_p]
        L.hx_enc_get_bitrate_float.restype = C.c_float
        L.hx_enc_get_bitrate2_float.argtypes = [C.c_void_p]
        L.hx_enc_get_bitrate2_float.restype = C.c_float
        L.hx_enc_get_frames.argtypes = [C.c_void_p]
        L.hx_enc_get_frames.restype = C.c_uint
        L.hx_enc_get_frames_bytes.argtypes = [C.c_void_p]
        L.hx_enc_get_frames_bytes.restype = IntPair
        L.hx_enc_info_ec.argtypes = [C.c_void_p, C.POINTER(EControl)]
        L.hx_enc_info_head.argtypes = [C.c_void_p, C.POINTER(MpegHead)]
        L.hx_enc_info_string.argtypes = [C.c_void_p, C.c_char_p]
        L.hx_batch_create.restype = C.c_void_p
        L.hx_batch_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.hx_batch_destroy.argtypes = [C.c_void_p]
        L.hx_batch_nstreams.argtypes = [C.c_void_p]
        L.hx_batch_out_stride.argtypes = [C.c_void_p, C.c_int]
        L.hx_batch_out_stride.restype = C.c_longlong
        L.hx_batch_encode_s16_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        L.hx_batch_encode_s16_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
        L.hx_batch_submit_s16_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        L.hx_batch_submit_f32_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        L.hx_batch_wait.argtypes = [C.c_void_p, C.c_void_p]
        L.hx_batch_submit_s16_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
        L.hx_batch_submit_f32_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
        L.hx_batch_wait_host.argtypes = [C.c_void_p]
        L.hx_pinned_alloc.argtypes = [C.c_longlong]
        L.hx_pinned_alloc.restype = C.c_void_p
        L.hx_pinned_free.argtypes = [C.c_void_p]
        L.hx_pinned_free.restype = None
        L.hx_batch_reset_stream.argtypes = [C.c_void_p, C.c_int]
        L.hx_batch_stream_state_bytes.argtypes = [C.c_void_p]
        L.hx_batch_stream_state_bytes.restype = C.c_longlong
        L.hx_batch_get_stream_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hx_batch_set_stream_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hx_batch_set_gate.argtypes = [C.c_void_p, C.c_int]
        L.hx_batch_set_gate.restype = None
        L.hx_batch_encode_f32_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        L.hx_batch_encode_f32_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
        L.hx_batch_status.argtypes = [C.c_void_p]
        if hasattr(L, "hx_multi_create"):           # absent from older builds selected through HMP3AMD_LIB (A/B timing runs)
            L.hx_batch_gate_timeouts.argtypes = [C.c_void_p]
            L.hx_multi_create.restype = C.c_void_p
            L.hx_multi_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
            L.hx_multi_destroy.argtypes = [C.c_void_p]
            L.hx_multi_ndevices.argtypes = [C.c_void_p]
            L.hx_multi_nstreams.argtypes = [C.c_void_p]
            L.hx_multi_shard.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
            L.hx_multi_batch.restype = C.c_void_p
            L.hx_multi_batch.argtypes = [C.c_void_p, C.c_int]
            L.hx_multi_out_stride.restype = C.c_longlong
            L.hx_multi_out_stride.argtypes = [C.c_void_p, C.c_int]
            L.hx_multi_encode_s16_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
            L.hx_multi_encode_f32_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
            L.hx_multi_encode_f32_host_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
            L.hx_multi_status.argtypes = [C.c_void_p]
        L.hx_batch_frames_bytes.argtypes = [C.c_void_p, C.c_int]
        L.hx_batch_frames_bytes.restype = IntPair
        L.hx_batch_alloc_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.hx_batch_alloc_kernel_ms.restype = C.c_float
        L.hx_batch_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong]
        L.hx_batch_debug_read.restype = C.c_longlong
        L.hx_batch_debug_enable.argtypes = [C.c_void_p, C.c_int]
        L.hx_debug_host_table.argtypes = [C.POINTER(EControl), C.c_char_p, C.c_void_p, C.c_longlong]
        L.hx_debug_host_table.restype = C.c_longlong
        _lib = L
    return _lib


def default_control(**kw):
    """CLI defaults (reference test/tomp3.cpp:357-384); bitrate=N per channel selects CBR (-B N)"""
    ec = EControl()
    lib().hx_default_control(C.byref(ec))
    for k, v in kw.items():
        setattr(ec, k, v)
    if kw.get("bitrate", -1) > 0 and "vbr_flag" not in kw:
        ec.vbr_flag = 0
    return ec


def libm_report(points=1000):
    """the host's C library and whether its logf / log10f agree with the restatement the first-generation allocator's
    kernels use (hx_libm32.h = glibc 2.35): {"glibc": "2.35", "points": 1000, "mismatches": 0}"""
    L = lib()
    L.hx_libc_version.restype = C.c_char_p
    return {"glibc": L.hx_libc_version().decode(), "points": points, "mismatches": int(L.hx_libm_spot_check(points))}


def device_numa_node(device):
    return int(lib().hx_device_numa_node(int(device)))


def bind_thread_to_device(device):
    """restrict the calling thread to the CPUs of the device's NUMA node; returns their number (0: nothing changed)"""
    return int(lib().hx_bind_thread_to_device(int(device)))


def refresh_process_cpus():
    """re-capture the CPUs the process may use (after a launcher narrowed it); returns their number"""
    return int(lib().hx_refresh_process_cpus())


def bind_thread_to_node(node):
    """restrict the calling thread to the CPUs of NUMA node `node` that the process may use; returns their number (0: nothing changed)"""
    return int(lib().hx_bind_thread_to_node(int(node)))


def build_id():
    """hash of the sources / flags the loaded library was built from (None for builds that predate it)"""
    L = lib()
    if not hasattr(L, "hx_build_id"):
        return None
    L.hx_build_id.restype = C.c_char_p
    return L.hx_build_id().decode()


def last_error():
    return lib().hx_last_error().decode()


class Batch:
    """N independent streams on one GPU (hx_batch_*)."""

    def __init__(self, controls, nstreams=None, max_frames=256, device=0):
        L = lib()
        if isinstance(controls, EControl):
            self.n = int(nstreams)
            self._ec = controls
            self.h = L.hx_batch_create(device, self.n, C.byref(controls), 1, max_frames)
        else:
            self.n = len(controls)
            arr = (EControl * self.n)(*controls)
            self._ec = arr
            self.h = L.hx_batch_create(device, self.n, arr, 0, max_frames)
        if not self.h:
            raise RuntimeError("hx_batch_create failed: " + last_error())
        self.max_frames = max_frames

    def out_stride(self, nframes):
        return int(lib().hx_batch_out_stride(self.h, nframes))

    def encode_host(self, pcm):
        """pcm: int16 (or float32 at int16 scale) [n, nframes*1152, 2] -> list of bytes per stream"""
        f32 = np.asarray(pcm).dtype == np.float32
        pcm = np.ascontiguousarray(pcm, dtype=np.float32 if f32 else np.int16)
        if pcm.ndim == 2:
            pcm = pcm[:, :, None]           # mono batch: [n, samples]
        assert pcm.shape[0] == self.n and pcm.shape[2] in (1, 2) and pcm.shape[1] % 1152 == 0
        nfr = pcm.shape[1] // 1152
        stride = self.out_stride(nfr)
        out = np.zeros((self.n, stride), dtype=np.uint8)
        nb = np.zeros(self.n, dtype=np.int32)
        fn = lib().hx_batch_encode_f32_host if f32 else lib().hx_batch_encode_s16_host
        r = fn(self.h, pcm.ctypes.data, nfr, out.ctypes.data, stride, nb.ctypes.data)
        if r != 0:
            raise RuntimeError("hx_batch_encode host call failed: " + last_error())
        return [out[i, :nb[i]].tobytes() for i in range(self.n)]

    def encode_device(self, d_pcm_ptr, nframes, d_out_ptr, out_stride, d_out_bytes_ptr, stream=None):
        r = lib().hx_batch_encode_s16_device(self.h, d_pcm_ptr, nframes, d_out_ptr, out_stride, d_out_bytes_ptr, stream)
        if r != 0:
            raise RuntimeError("hx_batch_encode_s16_device failed: " + last_error())

    def submit_device(self, d_pcm_ptr, nframes, d_out_ptr, out_stride, d_out_bytes_ptr, stream=None):
        """pipelined encode_device: outputs are ordered on `stream` by wait()"""
        r = lib().hx_batch_submit_s16_device(self.h, d_pcm_ptr, nframes, d_out_ptr, out_stride, d_out_bytes_ptr, stream)
        if r != 0:
            raise RuntimeError("hx_batch_submit_s16_device failed: " + last_error())

    def submit_host(self, pcm_ptr, nframes, out_ptr, out_stride, out_bytes_ptr):
        """pipelined host-buffer call (int16 PCM); outputs are valid after wait_host()"""
        if lib().hx_batch_submit_s16_host(self.h, pcm_ptr, nframes, out_ptr, out_stride, out_bytes_ptr) != 0:
            raise RuntimeError("hx_batch_submit_s16_host failed: " + last_error())

    def wait_host(self):
        if lib().hx_batch_wait_host(self.h) != 0:
            raise RuntimeError("hx_batch_wait_host failed: " + last_error())

    def reset_stream(self, i):
        """slot i starts a new stream (same configuration)"""
        if lib().hx_batch_reset_stream(self.h, i) != 0:
            raise RuntimeError("hx_batch_reset_stream failed: " + last_error())

    def get_stream_state(self, i):
        """checkpoint of stream i (bytes)"""
        buf = (C.c_ubyte * int(lib().hx_batch_stream_state_bytes(self.h)))()
        if lib().hx_batch_get_stream_state(self.h, i, buf) != 0:
            raise RuntimeError("hx_batch_get_stream_state failed: " + last_error())
        return bytes(buf)

    def set_stream_state(self, i, state):
        need = int(lib().hx_batch_stream_state_bytes(self.h))
        if len(state) != need:
            raise ValueError("stream state blob has %d bytes, this library's has %d" % (len(state), need))
        buf = (C.c_ubyte * len(state)).from_buffer_copy(state)
        if lib().hx_batch_set_stream_state(self.h, i, buf) != 0:
            raise RuntimeError("hx_batch_set_stream_state failed: " + last_error())

    def set_gate(self, percent):
        lib().hx_batch_set_gate(self.h, percent)

    def wait(self, stream=None):
        if lib().hx_batch_wait(self.h, stream) != 0:
            raise RuntimeError("hx_batch_wait failed: " + last_error())

    def status(self):
        return int(lib().hx_batch_status(self.h))

    def k6_variant(self):
        """0 = k_alloc (four streams per CU), 1 = k_alloc_slim (six)"""
        L = lib()
        L.hx_batch_k6_variant.argtypes = [C.c_void_p]
        return int(L.hx_batch_k6_variant(self.h))

    def resident_streams(self):
        L = lib()
        L.hx_batch_resident_streams.argtypes = [C.c_void_p]
        return int(L.hx_batch_resident_streams(self.h))

    def gate_timeouts(self):
        return int(lib().hx_batch_gate_timeouts(self.h))

    def frames_bytes(self, i):
        p = lib().hx_batch_frames_bytes(self.h, i)
        return p.a, p.b

    def alloc_kernel_ms(self):
        n = C.c_int(0)
        ms = lib().hx_batch_alloc_kernel_ms(self.h, C.byref(n))
        return float(ms), n.value

    def debug_enable(self, on=True):
        lib().hx_batch_debug_enable(self.h, 1 if on else 0)

    def debug_read(self, name, dtype, count):
        a = np.zeros(count, dtype=dtype)
        n = lib().hx_batch_debug_read(self.h, name.encode(), a.ctypes.data, a.nbytes)
        if n < 0:
            raise RuntimeError("unknown debug buffer " + name)
        return a[: n // a.itemsize]

    def close(self):
        if self.h:
            lib().hx_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multi:
    """nstreams streams over several GPUs of one node (hx_multi_*): contiguous blocks, one host thread per device"""

    def __init__(self, controls, nstreams=None, max_frames=256, ndev=0, devices=None):
        L = lib()
        dv = (C.c_int * len(devices))(*devices) if devices else None
        if devices:
            ndev = len(devices)
        if isinstance(controls, EControl):
            self.n = int(nstreams)
            self._ec = controls
            self.h = L.hx_multi_create(ndev, dv, self.n, C.byref(controls), 1, max_frames)
        else:
            self.n = len(controls)
            self._ec = (EControl * self.n)(*controls)
            self.h = L.hx_multi_create(ndev, dv, self.n, self._ec, 0, max_frames)
        if not self.h:
            raise RuntimeError("hx_multi_create failed: " + last_error())

    def ndevices(self):
        return int(lib().hx_multi_ndevices(self.h))

    def shard(self, k):
        d, f, c = C.c_int(), C.c_int(), C.c_int()
        if lib().hx_multi_shard(self.h, k, C.byref(d), C.byref(f), C.byref(c)) != 0:
            raise IndexError(k)
        return d.value, f.value, c.value

    def encode_host(self, pcm):
        f32 = np.asarray(pcm).dtype == np.float32
        pcm = np.ascontiguousarray(pcm, dtype=np.float32 if f32 else np.int16)
        nfr = pcm.shape[1] // 1152
        stride = int(lib().hx_multi_out_stride(self.h, nfr))
        out = np.zeros((self.n, stride), dtype=np.uint8)
        nb = np.zeros(self.n, dtype=np.int32)
        fn = lib().hx_multi_encode_f32_host if f32 else lib().hx_multi_encode_s16_host
        if fn(self.h, pcm.ctypes.data, nfr, out.ctypes.data, stride, nb.ctypes.data) != 0:
            raise RuntimeError("hx_multi_encode host call failed: " + last_error())
        return [out[i, :nb[i]].tobytes() for i in range(self.n)]

    def status(self):
        return int(lib().hx_multi_status(self.h))

    def close(self):
        if self.h:
            lib().hx_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mp3Enc:
    """Same surface as the reference's CMp3Enc (pub/mp3enc.h:74-139), one stream."""

    def __init__(self, device=0):
        self.h = lib().hx_enc_create(device)
        self._out = (C.c_ubyte * (1 << 17))()

    def L3_audio_encode_init(self, ec):
        return lib().hx_enc_L3_audio_encode_init(self.h, C.byref(ec))

    def MP3_audio_encode_init(self, ec, source_bits=16, source_is_float=0, mpeg_select=0, mono_convert=0):
        return lib().hx_enc_MP3_audio_encode_init(self.h, C.byref(ec), source_bits, source_is_float, mpeg_select, mono_convert)

    def L3_audio_encode(self, pcm_f32):
        pcm = np.ascontiguousarray(pcm_f32, dtype=np.float32)
        x = lib().hx_enc_L3_audio_encode(self.h, pcm.ctypes.data, self._out)
        return x.in_bytes, bytes(self._out[: x.out_bytes])

    def L3_audio_encode_Packet(self, pcm_f32):
        """-> (in_bytes, bitstream bytes, packet bytes); self.packet_sizes = nbytes_out[2] (MPEG-2: two packets)"""
        pcm = np.ascontiguousarray(pcm_f32, dtype=np.float32)
        pk = (C.c_ubyte * 4096)()
        nb = (C.c_int * 2)()
        x = lib().hx_enc_L3_audio_encode_Packet(self.h, pcm.ctypes.data, self._out, pk, nb)
        self.packet_sizes = (nb[0], nb[1])
        return x.in_bytes, bytes(self._out[: x.out_bytes]), bytes(pk[: nb[0] + nb[1]])

    def MP3_audio_encode(self, pcm_i16):
        pcm = np.ascontiguousarray(pcm_i16, dtype=np.int16)
        x = lib().hx_enc_MP3_audio_encode(self.h, pcm.ctypes.data, self._out)
        return x.in_bytes, bytes(self._out[: x.out_bytes])

    def L3_audio_encode_get_frames(self):
        return int(lib().hx_enc_get_frames(self.h))

    def L3_audio_encode_get_bitrate_float(self):
        return float(lib().hx_enc_get_bitrate_float(self.h))

    def L3_audio_encode_get_frames_bytes(self):
        p = lib().hx_enc_get_frames_bytes(self.h)
        return p.a, p.b

    def L3_audio_encode_info_ec(self):
        ec = EControl()
        lib().hx_enc_info_ec(self.h, C.byref(ec))
        return ec

    def L3_audio_encode_info_head(self):
        h = MpegHead()
        lib().hx_enc_info_head(self.h, C.byref(h))
        return h

    def L3_audio_encode_info_string(self):
        s = C.create_string_buffer(160)
        lib().hx_enc_info_string(self.h, s)
        return s.value.decode()

    def close(self):
        if self.h:
            lib().hx_enc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
