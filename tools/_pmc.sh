cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for U in 16 1; do
 for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/tools/pmc_run.py 1024 64 $U 2>&1 | grep "k_alloc ms"
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1); [ -z "$f" ] && find /tmp/pm | head
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "k_alloc" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print("   ", k, acc[k] / n[k])
PY
 done
done
