import os, sys, subprocess
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests","golden"))
import make_golden_cli as M
bad=0
for name in ["cli_lsf_vbr50_f32_24k","cli_cbr128_s16_44k","cli_rifx_cbr64_s24_44k"]:
    seed, nsamp, sr, as_float, bursts, flags = M.CASES[name]
    wav="/tmp/in_%s.wav"%name; mp3="/tmp/out_%s.mp3"%name
    M.write_wav(wav, M.case_pcm(name), sr, as_float, M.CONTAINER.get(name))
    gold=open(os.path.join(ROOT,"tests","golden",name+".mp3"),"rb").read()
    for i in range(25):
        with open(wav,"rb") as f:
            r=subprocess.run([os.path.join(ROOT,"hmp3_amd","hmp3amd"),"-",mp3]+flags+["-EC"],stdin=f,capture_output=True)
        ok = r.returncode==0 and open(mp3,"rb").read()==gold
        if not ok:
            bad+=1; print(name,i,"rc",r.returncode, r.stderr.decode()[-600:])
print("bad",bad)
