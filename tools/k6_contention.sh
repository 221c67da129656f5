#!/bin/bash
# What do co-resident streams contend for?  K6 alone at 1 .. 4 (k_alloc) and 1 .. 6 (k_alloc_slim) streams per CU, once
# un-profiled (launch and per-stream times) and once per counter group (rocprofv3 --pmc, --kernel-trace only, program after --).
#   bash tools/k6_contention.sh <outdir under gpurun_out> [F] [cfg]
out=$1; F=${2:-256}; CFG=${3:-2}
PLAN=${K6_PLAN:-"fat:256,512,768,1024;slim:256,512,1024,1536"}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 tools/k6_alone.py $F $CFG "$PLAN" 3 > "$out/unprofiled.txt" 2>&1
pass() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o p -- python3 tools/k6_alone.py $F $CFG "$PLAN" 2 > "$out/$name.log" 2>&1 || echo "pass $name failed" >> "$out/failed.txt"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SMEM
pass sq4 SQ_INST_LEVEL_LDS SQ_INSTS_LDS
pass sq5 SQ_IFETCH SQ_IFETCH_LEVEL
pass sq6 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES
python3 tools/k6_contention.py "$out" > "$out/summary.json" 2> "$out/summary.err"
for d in sq1 sq2 sq3 sq4 sq5 sq6; do rm -rf "$out/$d"; done
