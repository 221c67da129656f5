#!/bin/bash
mkdir -p gpurun_out
exec > gpurun_out/r4_prof2.log 2>&1
export HMP3AMD_LIB=hmp3_amd/libhmp3amd_prof.so
for v in fat slim; do
  echo "=== config 2, $v"; HMP3AMD_K6=$v timeout 600 python tools/gpu_prof_bench.py 2>&1 | grep -v amdgpu.ids
done
