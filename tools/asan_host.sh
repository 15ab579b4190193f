#!/bin/bash
# Host-side AddressSanitizer pass (no GPU needed; GPU ASAN is not available on this pool):
#   * the oracle's C restatement under -fsanitize=address,undefined, driven by the CPU test-suite;
#   * libpsk.so with the HOST half of every translation unit instrumented (-Xarch_host -fsanitize=address; the device
#     code is compiled as usual), driven by the ABI / host tests and by tools/fuzz_framing.py.
# Builds into a scratch directory; the in-tree libraries are not touched except for the oracle, which is restored.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$(mktemp -d)}
mkdir -p "$OUT"
cd "$ROOT"
GCC_ASAN=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
gcc -O1 -g -fPIC -Wall -std=c11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o "$OUT/libpsk_oracle.so" oracle/psk_oracle.c -lm
cp oracle/libpsk_oracle.so "$OUT/oracle_orig.so"
trap 'cp "$OUT/oracle_orig.so" "$ROOT/oracle/libpsk_oracle.so"' EXIT
cp "$OUT/libpsk_oracle.so" oracle/libpsk_oracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$GCC_ASAN python -m pytest tests/test_oracle_golden.py tests/test_host_modeling.py tests/test_weights.py -x -q -m "not gpu"
cp "$OUT/oracle_orig.so" oracle/libpsk_oracle.so

SRCS=$(sed -n 's/^SRCS := //p' phenotypeseeker_amd/csrc/Makefile | sed 's/\.hip//g')      # every translation unit of the library
for f in $SRCS; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xarch_host -fsanitize=address \
        -c phenotypeseeker_amd/csrc/$f.hip -o "$OUT/$f.o" &
    while [ "$(jobs -r | wc -l)" -ge 8 ]; do sleep 0.5; done      # (eight compilers at a time: the build box has eight cores)
done
wait
OBJS=""
for f in $SRCS; do OBJS="$OBJS $OUT/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -o "$OUT/libpsk.so" $OBJS -lpthread -lz
CLANG_ASAN=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
PSK_LIB="$OUT/libpsk.so" ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$CLANG_ASAN python -m pytest tests/test_abi_and_host.py -x -q
PSK_LIB="$OUT/libpsk.so" ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$CLANG_ASAN python tools/fuzz_framing.py 1 30000
echo "host ASAN pass ok ($OUT)"
