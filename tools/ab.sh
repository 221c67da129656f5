#!/bin/bash
# A/B library builds on the same box (hmp3_amd/libhmp3amd_<name>.so), alternating, AB_ROUNDS rounds (default 2):
#   AB_ARGS="--config 3 --steps 8" bash tools/ab.sh base d3 ...
for r in $(seq 1 ${AB_ROUNDS:-2}); do for v in "$@"; do echo -n "$v: "; HMP3AMD_LIB=hmp3_amd/libhmp3amd_$v.so python bench.py --no-cpu-baseline --no-worst-case --host-fed 0 --other-configs 0 --verify ${AB_VERIFY:-0} $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('verified_streams'), d['kernel_status'])"; done; done
