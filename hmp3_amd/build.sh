#!/bin/bash
# Build the MI355X encoder library in-tree: hmp3_amd/libhmp3amd.so (gfx950 only).
# -ffp-contract=off: the kernels must not fuse multiply-adds (bit-exactness against the oracle).
set -e
cd "$(dirname "$0")/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 ${HX_OPT:--O3} $HX_EXTRA -fPIC -ffp-contract=off -fno-fast-math -std=c++17 -Wall -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value -Wno-unused-result"
$HIPCC $FLAGS -c hx_front.hip -o hx_front.o
$HIPCC $FLAGS -c hx_alloc.hip -o hx_alloc.o
$HIPCC $FLAGS -c hx_alloc_lsf.hip -o hx_alloc_lsf.o
$HIPCC $FLAGS -c hx_alloc1.hip -o hx_alloc1.o
$HIPCC $FLAGS -c hx_alloc1_lsf.hip -o hx_alloc1_lsf.o
$HIPCC $FLAGS -c hx_pack.hip -o hx_pack.o
$HIPCC $FLAGS -c hx_cabi.hip -o hx_cabi.o
g++ -O2 -fPIC -ffp-contract=off -std=c++17 -c hx_host.cpp -o hx_host.o
g++ -O2 -fPIC -ffp-contract=off -std=c++17 -c hx_xhead.cpp -o hx_xhead.o
g++ -O2 -fPIC -ffp-contract=off -std=c++17 -c hx_src.cpp -o hx_src.o
$HIPCC --offload-arch=gfx950 -shared -o ../${HX_LIBNAME:-libhmp3amd.so} hx_front.o hx_alloc.o hx_alloc_lsf.o hx_alloc1.o hx_alloc1_lsf.o hx_pack.o hx_cabi.o hx_host.o hx_xhead.o hx_src.o
rm -f *.o
[ -n "$HX_LIBNAME" ] || g++ -O2 -std=c++17 -Wall ../cli/hmp3amd.cpp -o ../hmp3amd -L.. -lhmp3amd -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
echo built hmp3_amd/libhmp3amd.so hmp3_amd/hmp3amd
