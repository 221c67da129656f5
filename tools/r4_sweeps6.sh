#!/bin/bash
# Sixth sweep of round 4 (7-granule polyphase tiles): general seeds on both stream-walk kernels, random call lengths.
# bash tools/r4_sweeps6.sh > gpurun_out/r04_sweeps6.log
python -c "from hmp3_amd import api; print('build', api.build_id())"
for job in "fat 1500 4601" "slim 1500 4602" "fat --submit 400 4603"; do
  set -- $job; v=$1; shift
  echo "== HMP3AMD_K6=$v fuzz_parity $*"
  HMP3AMD_K6=$v timeout 1200 python tools/fuzz_parity.py "$@" 2>&1 | grep -v amdgpu.ids | tail -6
done
