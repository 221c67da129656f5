#!/bin/bash
# Everything profiles/rNN_* holds, collected on the GPU box in one gpurun call:
#   bash tools/collect_profiles.sh r03        (writes gpurun_out/profiles_r03/; copy what is to be judged into profiles/)
# 1. rocprofv3 --kernel-trace --stats of the bench command for configs 2 and 3 (k_* rows of the kernel_stats.csv);
# 2. counter passes (tools/collect_pmc.sh) for configs 2 and 3; 3. bench lines of BASELINE configs 2..5, which then quote 2.
set -e
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_$R
mkdir -p $O
for c in 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$c -o t -- python3 bench.py --config $c --no-cpu-baseline --no-worst-case --host-fed 0 --other-configs 0 --verify 0 > $O/kt$c.log 2>&1 || true
  f=$(find $O/kt$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { head -1 "$f"; grep '^"k_' "$f"; } > $O/${R}_kernel_stats_bench_config$c.csv
  rm -rf $O/kt$c
done
bash tools/collect_pmc.sh $O/pmc2 && cp $O/pmc2/summary.json $O/${R}_pmc_counters_config2_1024x256.json
bash tools/collect_pmc.sh $O/pmc3 --config 3 && cp $O/pmc3/summary.json $O/${R}_pmc_counters_config3_4096x256.json
# the bench lines last, with this build's counter files in place (in this scratch copy of the repo), so that they quote them
cp $O/${R}_pmc_counters_*.json profiles/
for c in 2 3 4 5; do
  python3 bench.py --config $c ${BENCH_EXTRA} 2> $O/bench_config$c.err | tail -1 > $O/${R}_bench_line_config$c.json || true
done
python3 bench.py --config 2 --no-pipeline --no-worst-case --host-fed 0 --other-configs 0 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_line_config2_no_pipeline.json || true
ls -la $O
