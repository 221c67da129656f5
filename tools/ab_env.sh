#!/bin/bash
# A/B of an environment switch on one box, alternating: AB_ARGS="--config 3 --steps 6" bash tools/ab_env.sh HMP3AMD_FRONT_CHUNK 0 64 128 ...
var=$1; shift
for r in $(seq 1 ${AB_ROUNDS:-2}); do for v in "$@"; do echo -n "$var=$v: "; env $var=$v python3 bench.py --no-cpu-baseline --no-worst-case --host-fed 0 --other-configs 0 --verify ${AB_VERIFY:-0} $AB_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'ms/step', d['ms_per_step'], 'K6', d['roofline']['kernel_ms'], 'verified', d.get('verified_streams'), 'status', d['kernel_status'])"; done; done
