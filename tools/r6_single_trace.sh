#!/bin/bash
# per-kernel durations of the one-stream chain (hx_enc_*, graph replay): plain calls under rocprofv3 --kernel-trace: what the thirteen kernels of a call take by themselves
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
HMP3AMD_ENC_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o t -- python3 tools/bench_single.py 1500 > $out/log.txt 2>&1
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    if n.startswith('k_'):
        print('%-16s calls %5s  mean %8.2f us' % (n, r['Calls'], float(r['AverageNs'])*1e-3)); tot+=float(r['AverageNs'])*1e-3
print('sum of kernel means %.1f us' % tot)
" "$f"
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-26:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-16s start %8.1f us  end %8.1f  dur %6.1f" % (r["Kernel_Name"].split("(")[0][:16], s, e, e - s))
PY
tail -1 $out/log.txt
rm -rf $out/kt
