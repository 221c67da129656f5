#!/bin/bash
# A/B two library builds on the same box: alternate, 3 rounds
for r in 1 2 3; do for v in "$@"; do echo -n "$v: "; HMP3AMD_LIB=hmp3_amd/libhmp3amd_$v.so python bench.py --no-cpu-baseline --no-worst-case --verify 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done; done
