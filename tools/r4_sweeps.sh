#!/bin/bash
# round 4's long sweeps on the final build, both builds of the stream walk: fuzz_parity (host calls, overlapped submits),
# fuzz_mixed, soak.  Writes gpurun_out/r4_sweeps.log
mkdir -p gpurun_out
exec > gpurun_out/r4_sweeps.log 2>&1
python -c "from hmp3_amd import api; print('build', api.build_id())"
for v in slim fat; do
  export HMP3AMD_K6=$v
  echo "=== $v: fuzz_parity 2500 host calls (seed 4101)"; timeout 1500 python tools/fuzz_parity.py 2500 4101 2>&1 | tail -3
  echo "=== $v: fuzz_parity 1200 overlapped submits (seed 4102)"; timeout 1200 python tools/fuzz_parity.py --submit 1200 4102 2>&1 | tail -3
  echo "=== $v: fuzz_mixed 120 batches (seed 4103)"; timeout 900 python tools/fuzz_mixed.py 120 4103 2>&1 | tail -3
done
export HMP3AMD_K6=slim
echo "=== slim: soak 20000 frames x 6 configurations x 6 streams"; timeout 1500 python tools/soak.py 20000 2>&1 | tail -8
