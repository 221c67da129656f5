#!/bin/bash
# Fifth sweep of round 4 (final build): another general seed per stream-walk kernel, overlapped submits, the first-generation
# allocator, whole files beside the reference binary.   bash tools/r4_sweeps5.sh > gpurun_out/r04_sweeps5.log
python -c "from hmp3_amd import api; print('build', api.build_id())"
for job in "fat 1500 4501" "slim 1500 4502" "slim --submit 500 4503" "fat --a1 500 4504"; do
  set -- $job; v=$1; shift
  echo "== HMP3AMD_K6=$v fuzz_parity $*"
  HMP3AMD_K6=$v timeout 1200 python tools/fuzz_parity.py "$@" 2>&1 | grep -v amdgpu.ids | tail -6
done
echo "== fuzz_cli 120 (seed 4505)"
timeout 900 python tools/fuzz_cli.py 120 4505 2>&1 | grep -v amdgpu.ids | tail -3
