#!/bin/bash
# Fourth sweep of round 4 (after k_prep's grouped band maxima): general and -HF seeds on both stream-walk kernels, mixed-class batches.
# bash tools/r4_sweeps4.sh > gpurun_out/r04_sweeps4.log
python -c "from hmp3_amd import api; print('build', api.build_id())"
for job in "fat 2500 4401" "slim 2500 4402" "fat --hf 1500 4403" "slim --hf 1500 4404"; do
  set -- $job; v=$1; shift
  echo "== HMP3AMD_K6=$v fuzz_parity $*"
  HMP3AMD_K6=$v timeout 1500 python tools/fuzz_parity.py "$@" 2>&1 | grep -v amdgpu.ids | tail -6
done
echo "== fuzz_mixed 100 (seed 4405)"
timeout 900 python tools/fuzz_mixed.py 100 4405 2>&1 | grep -v amdgpu.ids | tail -3
