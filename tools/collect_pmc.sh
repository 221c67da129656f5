#!/bin/bash
# Counter passes for profiles/rNN_pmc_counters_*.json on the GPU box (run through gpurun):
#   bash tools/collect_pmc.sh <outdir under gpurun_out> [bench.py args, e.g. --config 3]
# One rocprofv3 run per counter group (SQ takes 8 per pass, FETCH_SIZE and WRITE_SIZE cannot share one), each with
# --kernel-trace only, as the MI355X guide prescribes; the program itself follows "--".
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
B="--steps 1 --warmup 0 --no-cpu-baseline --no-worst-case --host-fed 0 --other-configs 0 --verify 0 $*"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o p -- python3 bench.py $B > "$out/$name.log" 2>&1 || true; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/pmc_summarize.py "$out" $* > "$out/summary.json"
for d in sq1 sq2 sq3 fetch write; do rm -rf "$out/$d"; done
