#!/bin/bash
# round 4, first look at k_alloc_slim on the GPU: stage-by-stage check, a parity slice, K6 time on configs 2 and 3
mkdir -p gpurun_out
exec > gpurun_out/r4_first.log 2>&1
export HMP3AMD_K6=slim
for cfg in cbr128 vbr50_sw vbr100hf_sw cbr128lr_sw cbr128_thr100; do
  echo "== gpu_check $cfg (slim)"; timeout 300 python tests/gpu_check.py 8 24 $cfg 2>&1 | tail -6
done
unset HMP3AMD_K6
echo "== pytest slim"; timeout 900 python -m pytest tests -m gpu -q -x -k "slim" 2>&1 | tail -15
for c in 3 2; do
  for v in fat slim; do
    echo "== bench config $c $v"; HMP3AMD_K6=$v timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-worst-case --host-fed 0 2>&1 | tail -2
  done
done
