#!/bin/bash
# Bound of a polyphase -> MDCT fusion (EXPERIMENTS.md round 6): per-kernel times of plain calls at config 3's size with the
# product library and with the timing-only variant whose k_polyphase stores no subband samples and whose k_spec_direct reads
# them from cache (libhmp3amd_mockfuse.so: results are garbage, only k_polyphase / k_spec_direct rows are read).
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in base mockfuse base mockfuse; do
  HMP3AMD_LIB=hmp3_amd/libhmp3amd_$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$v -o t -- python3 tools/k6_alone.py 256 3 "slim:4096" 2 > $out/kt_$v.log 2>&1
  f=$(find $out/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; [ -n "$f" ] && python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    if n.startswith('k_'): print('%-16s calls %4s  mean %10.3f ms' % (n, r['Calls'], float(r['AverageNs'])*1e-6))
" "$f"
  rm -rf $out/kt_$v
done
