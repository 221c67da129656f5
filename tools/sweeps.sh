#!/bin/bash
# The long randomised parity sweeps of a build, both builds of the stream walk (through gpurun):
#   bash tools/sweeps.sh <label> <seed base> [general cases] [-HF cases] [overlapped-submit cases] [mixed-class batches] [soak frames] [strict-sum cases]
# fuzz_parity (random controls x random signals, host calls in random-sized pieces) on k_alloc and k_alloc_slim, the -HF
# slice, the overlapped-submit slice, the same with every certified band sum replaced by the strict one
# (HMP3AMD_EXACT_SUMS=1: the two must produce the same bytes as the oracle), fuzz_mixed, a soak of the slim build.
# Writes gpurun_out/<label>_sweeps.log (copy it to profiles/ with the results commit: one per build id).
label=${1:?label}; seed=${2:?seed base}
ngen=${3:-1500}; nhf=${4:-1000}; nsub=${5:-400}; nmix=${6:-60}; soak=${7:-6000}; nstrict=${8:-400}
mkdir -p gpurun_out
exec > gpurun_out/${label}_sweeps.log 2>&1
python -c "from hmp3_amd import api; print('build', api.build_id())"
# every slice's exit status is recorded (rc=N; 124 = killed by timeout) and its summary line must say ", 0 bad" (the soak: a line ending in "ok"):
# a slice that crashed or was cut short marks the whole log FAILED and the script exits non-zero
fail=0
run() {
  echo "== $*"
  local t; t=$(mktemp)
  "$@" > "$t" 2>&1; local rc=$?
  grep -v amdgpu.ids "$t" | tail -4
  if [ $rc -ne 0 ] || ! grep -qE ", 0 bad| ok$" "$t"; then echo "rc=$rc FAILED: $*"; fail=1; else echo "rc=0"; fi
  rm -f "$t"
}
for v in slim fat; do
  export HMP3AMD_K6=$v
  [ $ngen -gt 0 ] && run timeout 3000 python tools/fuzz_parity.py $ngen $((seed + 1))
  [ $nhf -gt 0 ] && run timeout 3000 python tools/fuzz_parity.py --hf $nhf $((seed + 2))
  [ $nsub -gt 0 ] && run timeout 2000 python tools/fuzz_parity.py --submit $nsub $((seed + 3))
  [ $nstrict -gt 0 ] && HMP3AMD_EXACT_SUMS=1 run timeout 2000 python tools/fuzz_parity.py $nstrict $((seed + 4))
  seed=$((seed + 10))
done
unset HMP3AMD_K6
[ $nmix -gt 0 ] && run timeout 1500 python tools/fuzz_mixed.py $nmix $((seed + 5))
[ $soak -gt 0 ] && HMP3AMD_K6=slim run timeout 1500 python tools/soak.py $soak
if [ $fail -ne 0 ]; then echo "== FAILED (at least one slice did not finish clean)"; exit 1; fi
echo "== done"
