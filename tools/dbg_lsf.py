import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from hmp3_amd import api, synth
from oracle import oracle as O
kw = dict(bitrate=32, samprate=22050, mode=3)
F = 26
pcm = synth.stream_pcm(77, F, sr=22050)
pcm[F*1152//2:] = -32768
pcm = pcm[:, :1]
enc = O.OracleEncoder(O.default_control(**kw))
dbg = O.oracle_enable_debug(enc)
b = api.Batch(api.default_control(**kw), nstreams=1, max_frames=1)
for f in range(F):
    fr = pcm[f*1152:(f+1)*1152]
    w = enc.encode_s16(fr[:, 0])
    g = b.encode_host(np.ascontiguousarray(fr[None, :, 0]))[0]
    oix = np.array(dbg.ix).reshape(2, 2, 576)
    osg = np.array(dbg.signx).reshape(2, 2, 576)
    ogr = np.array(dbg.gr).reshape(2, 2, 27)
    ixq = b.debug_read("ixq", np.int16, 2 * 1152).reshape(2, 2, 576)
    sgn = b.debug_read("sgn", np.uint8, 2 * 1152).reshape(2, 2, 576)
    seg = b.debug_read("seg", np.uint8, 104 * 4).reshape(2, 2, 104)
    st = b.status()
    print(f, "same" if w == g else "DIFF", "status", st, "bt", list(dbg.block_type))
    if w != g or st:
        for gr in range(2):
            a, c = oix[gr, 0], ixq[gr, 0].astype(np.int32) & 0xFFFF
            nz = np.nonzero(a)[0]
            hi = nz.max() + 1 if len(nz) else 0
            print(" gr", gr, "oracle ix max", a.max(), "gpu ixq max", c.max(), "equal on coded range", np.array_equal(a[:hi], c[:hi]), "signs eq", np.array_equal(osg[gr,0][:hi]*(a[:hi]!=0), sgn[gr,0][:hi]*(a[:hi]!=0)))
            print("   oracle gr", dict(zip(O.GR_FIELDS, ogr[gr,0][:24])))
            sg = seg[gr, 0]
            print("   seg start_bit", sg[0:4].view(np.int32)[0], "huff_bits", sg[4:8].view(np.int32)[0], "nreg", sg[8:14].view(np.uint16), "nquads", sg[14:16].view(np.uint16)[0], "tab", sg[16:19], "c1", sg[19], "nn", sg[20])
        break
