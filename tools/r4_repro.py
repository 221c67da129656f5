import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from hmp3_amd import api, synth
from oracle import oracle as O
kw = {'samprate': 44100, 'mode': 3, 'bitrate': 160, 'hf_flag': 3, 'nsbstereo': 16, 'nsb_limit': 28}
F = 12
for bursts in (False, True):
    pcm = synth.stream_pcm(721109, F, sr=44100, rho=1.0, bursts=bursts)
    pcm = np.ascontiguousarray(pcm[None, :, 0])
    for var in ("fat", "slim"):
        os.environ["HMP3AMD_K6"] = var
        b = api.Batch(api.default_control(**kw), nstreams=1, max_frames=F)
        got = b.encode_host(pcm)[0]
        st = b.status(); b.close()
        enc = O.OracleEncoder(O.default_control(**kw))
        want = b"".join(enc.encode_s16(pcm[0, f * 1152:(f + 1) * 1152]) for f in range(F))
        nd = next((k for k in range(min(len(got), len(want))) if got[k] != want[k]), -1)
        print("bursts", bursts, var, "equal" if got == want else "DIFF at %d" % nd, len(got), len(want), "status", st)
kw = {'samprate': 44100, 'mode': 0, 'bitrate': 160, 'hf_flag': 3, 'filter_select': 1}
F = 120
for bursts in (False, True):
    pcm = synth.stream_pcm(395960, F, sr=44100, rho=0.7, bursts=bursts)[None]
    for var in ("fat", "slim"):
        os.environ["HMP3AMD_K6"] = var
        b = api.Batch(api.default_control(**kw), nstreams=1, max_frames=F)
        got = b.encode_host(pcm)[0]
        st = b.status(); b.close()
        enc = O.OracleEncoder(O.default_control(**kw))
        want = b"".join(enc.encode_s16(pcm[0, f * 1152:(f + 1) * 1152]) for f in range(F))
        nd = next((k for k in range(min(len(got), len(want))) if got[k] != want[k]), -1)
        print("case601 bursts", bursts, var, "equal" if got == want else "DIFF at %d" % nd, len(got), len(want), "status", st)
