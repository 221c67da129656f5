#!/bin/bash
mkdir -p gpurun_out
exec > gpurun_out/r4_third.log 2>&1
echo "== occupancy sweep config 2"; timeout 900 python tools/r4_occ.py 64 2 2>&1 | grep -v amdgpu.ids
echo "== occupancy sweep config 3"; timeout 900 python tools/r4_occ.py 64 3 2>&1 | grep -v amdgpu.ids
echo "== pytest gpu (both builds)"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
for c in 3 2; do
  for v in fat slim; do
    echo "== bench config $c $v"; HMP3AMD_K6=$v timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-worst-case --host-fed 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['verify']['identical'])"
  done
done
