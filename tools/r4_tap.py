"""Lines of one frame, product vs oracle: python tools/r4_tap.py  (HMP3AMD_K6=slim|fat).  Prints where the quantised lines differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmp3_amd import api, synth
from oracle import oracle as O

kw = dict(samprate=44100, mode=0, bitrate=160, hf_flag=3, filter_select=1)
F = 24
pcm = synth.stream_pcm(395960, 120, sr=44100, rho=0.7, bursts=True)[None, :F * 1152]
NG = 2 * F
b = api.Batch(api.default_control(**kw), nstreams=1, max_frames=F)
b.debug_enable(True)
got = b.encode_host(pcm)
print("variant", b.k6_variant() if hasattr(b, "k6_variant") else "?")
ixq = b.debug_read("ixq", np.int16, NG * 1152).reshape(NG, 2, 576).astype(np.int32) & 0xFFFF
enc = O.OracleEncoder(O.default_control(**kw))
d = O.oracle_enable_debug(enc)
for f in range(F):
    enc.encode_s16(pcm[0, f * 1152:(f + 1) * 1152])
    oix = np.array(d.ix).reshape(2, 2, 576)
    ogr = np.array(d.gr).reshape(2, 2, 27)
    for igr in range(2):
        for ch in range(2):
            a, o = ixq[2 * f + igr, ch], oix[igr, ch]
            w = np.nonzero(a != o)[0]
            if f in (15, 16, 17) or len(w):
                print("f", f, "gr", igr, "ch", ch, "ndiff", len(w), "first", w[:8], "prod", a[w[:8]], "oracle", o[w[:8]],
                      "big_values", ogr[igr, ch, 1], "nquads", ogr[igr, ch, 18], "last nz prod", np.nonzero(a)[0][-1:], "oracle", np.nonzero(o)[0][-1:],
                      "o[570:576]", o[570:576], "o1[0:4]", oix[igr, 1, :4])
