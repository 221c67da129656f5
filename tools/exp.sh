cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r03
for c in 2 3 4 5; do python3 bench.py --config $c 2>/dev/null | tail -1 > gpurun_out/profiles_r03/r03_bench_line_config$c.json; done
