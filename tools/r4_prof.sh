#!/bin/bash
mkdir -p gpurun_out
exec > gpurun_out/r4_prof.log 2>&1
export HMP3AMD_LIB=hmp3_amd/libhmp3amd_prof.so
for v in fat slim; do
  echo "=== config 2, $v"; HMP3AMD_K6=$v timeout 600 python tools/gpu_prof_bench.py 2>&1 | grep -v amdgpu.ids
done
echo "=== config 2 rho=1, fat"; HMP3AMD_K6=fat PF_RHO=1.0 timeout 600 python tools/gpu_prof_bench.py 2>&1 | grep -v amdgpu.ids
echo "=== config 3 (1024 of its streams), slim"; HMP3AMD_K6=slim PF_CFG=3 timeout 600 python tools/gpu_prof_bench.py 2>&1 | grep -v amdgpu.ids
