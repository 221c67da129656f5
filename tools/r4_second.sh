#!/bin/bash
mkdir -p gpurun_out
exec > gpurun_out/r4_second.log 2>&1
echo "== CLI nopad test, both builds"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "nopad" 2>&1 | tail -15
echo "== occupancy sweep config 2"; timeout 900 python tools/r4_occ.py 64 2 2>&1 | grep -v amdgpu.ids
echo "== occupancy sweep config 3"; timeout 900 python tools/r4_occ.py 64 3 2>&1 | grep -v amdgpu.ids
