import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from hmp3_amd import api, synth
from oracle import oracle as O
kw = eval(sys.argv[1]) if len(sys.argv) > 1 else dict(bitrate=64, nsbstereo=8)
sr = kw.get("samprate", 44100)
F = 12
pcm = synth.stream_pcm(5200, F, sr=sr, rho=0.7, bursts=True)
enc = O.OracleEncoder(O.default_control(**kw))
dbg = O.oracle_enable_debug(enc)
b = api.Batch(api.default_control(**kw), nstreams=1, max_frames=1)
b.debug_enable(True)
class GDbg(C.Structure):
    _fields_ = [("ms", C.c_int), ("ms_metric", C.c_int * 2), ("byte_pool", C.c_int), ("MNR_after", C.c_int),
                ("mask_mb", C.c_int * 88), ("gr", C.c_int * 96), ("sf", C.c_int * 88), ("scfsi", C.c_int * 2), ("main_bytes", C.c_int)]
for f in range(F):
    fr = pcm[f*1152:(f+1)*1152]
    w = enc.encode_s16(fr)
    g = b.encode_host(np.ascontiguousarray(fr[None]))[0]
    oix = np.array(dbg.ix).reshape(2, 2, 576); ogr = np.array(dbg.gr).reshape(2, 2, 27)
    ixq = b.debug_read("ixq", np.int16, 2 * 1152).reshape(2, 2, 576).astype(np.int32) & 0xFFFF
    raw = b.debug_read("dbg", np.uint8, C.sizeof(GDbg))
    gd = GDbg.from_buffer_copy(raw.tobytes())
    ggr = np.array(gd.gr).reshape(2, 2, 24)
    gsf = np.array(gd.sf).reshape(2, 2, 22)
    print(f, "same" if w == g else "DIFF", "status", b.status(), "ms", gd.ms, dbg.ms, "metric", list(gd.ms_metric), list(dbg.ms_metric), "pool", gd.byte_pool, dbg.byte_pool)
    if w != g:
        for gr in range(2):
            for ch in range(2):
                n = 2 * int(ogr[gr, ch, 1]) + 4 * max(int(ogr[gr, ch, 18]), 0)
                print("  gr", gr, "ch", ch, "side eq", np.array_equal(ggr[gr, ch], ogr[gr, ch, :24]), "ix eq", np.array_equal(ixq[gr, ch, :n], oix[gr, ch, :n]), "n", n)
                if not np.array_equal(ggr[gr, ch], ogr[gr, ch, :24]):
                    print("    gpu", dict(zip(O.GR_FIELDS, ggr[gr, ch])))
                    print("    ora", dict(zip(O.GR_FIELDS, ogr[gr, ch, :24])))
                print("    sf gpu", list(gsf[gr, ch, :21]))
                print("    sf ora", list(np.array(dbg.sf).reshape(2,2,22)[gr, ch, :21]) if hasattr(dbg, "sf") else None)
        break
