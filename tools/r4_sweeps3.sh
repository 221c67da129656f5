#!/bin/bash
# Third sweep of round 4 (after the band-21 fix of k_alloc_slim): -HF configurations on both stream-walk kernels, and another general seed
# on the low-footprint one.  bash tools/r4_sweeps3.sh > gpurun_out/r04_sweeps3.log
python -c "from hmp3_amd import api; print('build', api.build_id())"
for job in "slim --hf 4000 4301" "slim 4000 4302" "fat --hf 2000 4303" "slim --submit 600 4304"; do
  set -- $job; v=$1; shift
  echo "== HMP3AMD_K6=$v fuzz_parity $*"
  HMP3AMD_K6=$v timeout 1500 python tools/fuzz_parity.py "$@" 2>&1 | grep -v amdgpu.ids | tail -6
done
