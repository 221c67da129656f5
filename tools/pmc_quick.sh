#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmcq
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmcq/sq -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-worst-case --host-fed 0 --other-configs 0 --verify 0 > gpurun_out/pmcq/log.txt 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for p in glob.glob("gpurun_out/pmcq/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].split("(")[0]
        if n.startswith("k_"): acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, c in acc.items():
    if c.get("SQ_LDS_IDX_ACTIVE"): print(n, "conflict %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]), "lds_active", c["SQ_LDS_IDX_ACTIVE"])
PY
rm -rf gpurun_out/pmcq/sq
