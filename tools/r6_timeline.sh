#!/bin/bash
# kernel timeline of the pipelined config-2 bench (last steps): where a step's time goes beside its slowest stream
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o t -- python3 bench.py --steps ${TL_STEPS:-6} --warmup 2 --no-cpu-baseline --no-worst-case --host-fed 0 --other-configs 0 --verify 0 ${BENCH_EXTRA} > $out/bench.log 2>&1
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
al = [r for r in rows if r["Kernel_Name"].startswith("k_alloc")]
t0 = int(al[-4]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if s < -14 or s > float(__import__("os").environ.get("TL_MAX", "40")): continue
    print("%-16s q%-3s start %9.3f  end %9.3f  dur %8.3f" % (r["Kernel_Name"].split("(")[0][:16], r.get("Queue_Id", "?"), s, e, e - s))
PY
grep '^{' $out/bench.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms per step', d['ms_per_step'], 'K6', d['roofline']['kernel_ms'], d['stream_ms'])" || true
rm -rf $out/kt
