#!/bin/bash
# Build the MI355X encoder library in-tree: hmp3_amd/libhmp3amd.so (gfx950 only).
# -ffp-contract=off: the kernels must not fuse multiply-adds (bit-exactness against the oracle).
# HX_EXTRA: extra compiler flags (e.g. -DHX_PROFILE); HX_ALLOC_EXTRA: the same for the allocator kernels only; HX_LIBNAME: build a variant library next to the product
# (objects go to a directory of their own, so variants can be built side by side).
set -e
cd "$(dirname "$0")/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LIB=${HX_LIBNAME:-libhmp3amd.so}
OBJ=$(mktemp -d /tmp/hxbuild.XXXXXX)
trap 'rm -rf "$OBJ"' EXIT
ILP="-mllvm -amdgpu-sched-strategy=iterative-ilp"
ALLOC_SCHED="${HX_ALLOC_SCHED-$ILP}"
# The allocator kernels without MachineLICM: hoisted out of the frame loop, some sixty constants and lane addresses (v_mov
# of an immediate, base + 4 * lane ...) each held a VGPR for the whole kernel while loop-carried values went to scratch and
# came back behind s_waitcnt vmcnt(0).  256 -> 217 VGPRs, no scratch access left in the frame loop, K6 -1.3 .. -2 %.
# (k_polyphase gains 3 % from the same switch; k_spec loses 2 %, k_prep and k_pack do not care.)
NOLICM="${HX_NOLICM--mllvm -disable-machine-licm}"
# build id = hash of the kernel / host sources and of the flags that change the generated code
BUILD_ID=$( (LC_ALL=C; cat *.hip *.inc *.h *.cpp ../build.sh; $HIPCC --version; echo "${HX_OPT:--O3} $HX_EXTRA ${HX_ALLOC_OPT:--O2} $HX_ALLOC_EXTRA $ALLOC_SCHED $NOLICM $HX_FRONT_EXTRA $HX_PACK_EXTRA") | sha256sum | cut -c1-16)
FLAGS="-DHX_BUILD_ID=\"$BUILD_ID\" --offload-arch=gfx950 ${HX_OPT:--O3} $HX_EXTRA -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wall -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value -Wno-unused-result"
pids=()
# hx_front.hip without SLP vectorisation: a packed f32 instruction (v_pk_mul_f32 / v_pk_add_f32) issues in the time of 1.65
# plain ones on this chip (tools/ubench/pk.hip: 69.8 against 57.7 T lane-operations/s), and the pairs the vectoriser forms
# cost k_spec 300 register moves (measured: k_spec 2.27 -> 2.16 ms).  k_polyphase is written in pairs by hand instead.
# Scheduling strategy per translation unit, by measurement (ILP = "-mllvm -amdgpu-sched-strategy=iterative-ilp"):
# the allocator kernels and k_spec / k_prep gain 1 .. 9 %, k_polyphase loses 18 %, k_pack does not care.  The allocator
# kernels are built at -O2: -O3 is 1 % slower there (measured twice, alternating builds), -Os 2.5 %.
$HIPCC $FLAGS -fno-slp-vectorize $NOLICM $HX_FRONT_EXTRA -DHX_FRONT_PART=1 -c hx_front.hip -o $OBJ/hx_front1.o & pids+=($!)
$HIPCC $FLAGS -fno-slp-vectorize $ILP $HX_FRONT_EXTRA -DHX_FRONT_PART=2 -c hx_front.hip -o $OBJ/hx_front2.o & pids+=($!)
for f in hx_alloc hx_alloc_slim hx_alloc_lsf hx_alloc1 hx_alloc1_lsf; do
  $HIPCC $FLAGS $ALLOC_SCHED $NOLICM ${HX_ALLOC_OPT:--O2} $HX_ALLOC_EXTRA -c $f.hip -o $OBJ/$f.o & pids+=($!)
done
for f in hx_pack hx_cabi; do
  $HIPCC $FLAGS $HX_PACK_EXTRA -c $f.hip -o $OBJ/$f.o & pids+=($!)
done
for f in hx_host hx_xhead hx_src; do
  g++ -O2 -fPIC -ffp-contract=off -std=c++17 -c $f.cpp -o $OBJ/$f.o & pids+=($!)
done
# (every compiler is waited for before a failure ends the script: the EXIT trap removes the object directory)
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc -eq 0 ] || { echo "build failed" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -o ../$LIB $OBJ/*.o
[ -n "$HX_LIBNAME" ] || g++ -O2 -std=c++17 -Wall ../cli/hmp3amd.cpp -o ../hmp3amd -L.. -lhmp3amd -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
echo built hmp3_amd/$LIB build_id=$BUILD_ID
