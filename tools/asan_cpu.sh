#!/bin/bash
# Host side of the library (parameter resolution and tables, tag frame, sample-rate converter) under AddressSanitizer +
# UBSan, on the CPU: the tests that need no GPU, against a variant library whose host-only sources are compiled with
# the sanitizers (a g++ wrapper on PATH adds the flags, so hmp3_amd/build.sh - and with it the product's build id -
# stays as it is).   bash tools/asan_cpu.sh
set -e
cd "$(dirname "$0")/.."
SHIM=$(mktemp -d /tmp/hxasan.XXXXXX)
trap 'rm -rf "$SHIM"' EXIT
printf '#!/bin/bash\nexec /usr/bin/g++ "$@" -fsanitize=address,undefined -fno-omit-frame-pointer -g\n' > $SHIM/g++
chmod +x $SHIM/g++
PATH=$SHIM:$PATH HX_LIBNAME=libhmp3amd_asan.so bash hmp3_amd/build.sh
# (the objects are gcc's, the link is hipcc's: gcc's sanitizer runtimes come in by preload)
ASAN_LIB="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
LD_PRELOAD="$ASAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 HMP3AMD_LIB=hmp3_amd/libhmp3amd_asan.so \
  python -m pytest tests/test_host_and_abi.py tests/test_src_convert.py tests/test_xing_tag.py -x -q -m "not gpu" "$@"
rm -f hmp3_amd/libhmp3amd_asan.so
