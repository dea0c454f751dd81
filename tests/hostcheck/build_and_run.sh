#!/bin/bash
# Builds the host side of ppde_api.hip (no device code) + the mock runtime + the driver with AddressSanitizer and runs
# it: bash tests/hostcheck/build_and_run.sh [sweep]. Output: tests/hostcheck/_build/ (git-ignored).
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); B=$HERE/_build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$B"
FLAGS="-O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer -fsanitize=address -w"
$HIPCC $FLAGS --cuda-host-only -c "$ROOT/ppde_amd/csrc/ppde_api.hip" -o "$B/api.o"
# the host object refers to the embedded device image by a hashed name: give it an empty one
SYM=$(nm "$B/api.o" | awk '/ U __hip_fatbin/ {print $2; exit}')
echo "const char ${SYM:-__hip_fatbin_unused}[64] = {0};" > "$B/fatbin_stub.c"
$HIPCC $FLAGS --cuda-host-only -x hip -c "$HERE/hipmock.cpp" -o "$B/hipmock.o"
g++ -O1 -g -std=c++17 -fno-omit-frame-pointer -c "$HERE/driver.cpp" -o "$B/driver.o"
gcc -c "$B/fatbin_stub.c" -o "$B/fatbin_stub.o"
# the split-precision helpers of cnn.h (scales, two-term fp16 split, cross-term error bound): host-only, no runtime calls
$HIPCC -O1 -std=c++17 -w --cuda-host-only -x hip "$HERE/splitcheck.cpp" -o "$B/splitcheck" -L/opt/rocm/lib -lamdhip64
"$B/splitcheck"
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address "$B/api.o" "$B/hipmock.o" "$B/driver.o" "$B/fatbin_stub.o" -o "$B/hostcheck" -lpthread -ldl
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 "$B/hostcheck" "$@"
