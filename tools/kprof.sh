#!/bin/bash
# kernel averages of plain (unpipelined) bench calls for the given library variants: tools/kprof.sh A B ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  HMP3AMD_LIB=hmp3_amd/libhmp3amd_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp_$v -o t -- python3 bench.py --no-cpu-baseline --no-worst-case --host-fed 0 --other-configs 0 --no-pipeline --verify 0 --steps 4 $KPROF_ARGS > /dev/null 2>&1
  f=$(find gpurun_out/kp_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; python tools/kstats.py $f; rm -rf gpurun_out/kp_$v
done
