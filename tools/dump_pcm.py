"""Dump the bench's synthetic PCM of a few streams (the GPU generator's noise cannot be re-made on the CPU) for offline
analysis with the oracle.   python tools/dump_pcm.py <out.npz> <cfg> <ids,comma> [F]  (AB_RHO as in tools/k6_tail.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
out, CFG, ids = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")]
F = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
w = bench.workload(CFG)
S = w["S"]
rho = [float(x) for x in os.environ["AB_RHO"].split(",")] if os.environ.get("AB_RHO") else w["rho"]
ncls = len(w["classes"])
pcm = bench.synth_batch_gpu(torch, np, S, F, [w["classes"][i % ncls][1] for i in range(S)], [rho[i % len(rho)] for i in range(S)], w["bursts"], dev)
np.savez_compressed(out, ids=np.array(ids), pcm=pcm[torch.as_tensor(ids, device=dev)].cpu().numpy())
print("wrote", out, ids)
