#!/bin/bash
# round 4, second long sweep on the final build: other seeds for the low-footprint stream walk, the first-generation allocator's
# slice, whole files beside the reference's binary.  Writes gpurun_out/r4_sweeps2.log
mkdir -p gpurun_out
exec > gpurun_out/r4_sweeps2.log 2>&1
python -c "from hmp3_amd import api; print('build', api.build_id(), api.libm_report(100000))"
export HMP3AMD_K6=slim
echo "=== slim: fuzz_parity 3000 host calls (seed 4201)"; timeout 1800 python tools/fuzz_parity.py 3000 4201 2>&1 | tail -2
echo "=== slim: fuzz_parity 1500 overlapped submits (seed 4202)"; timeout 1500 python tools/fuzz_parity.py --submit 1500 4202 2>&1 | tail -2
echo "=== slim: fuzz_mixed 150 batches (seed 4203)"; timeout 1200 python tools/fuzz_mixed.py 150 4203 2>&1 | tail -2
unset HMP3AMD_K6
echo "=== first-generation allocator: fuzz_parity --a1 800 (seed 4204)"; timeout 1200 python tools/fuzz_parity.py --a1 800 4204 2>&1 | tail -2
echo "=== whole files beside the reference binary: fuzz_cli 200 (seed 4205)"; timeout 1500 python tools/fuzz_cli.py 200 4205 2>&1 | tail -3
