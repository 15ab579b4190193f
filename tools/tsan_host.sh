#!/bin/bash
# ThreadSanitizer pass over the threaded HOST side of libpsk (no GPU needed):
#   every translation unit's host half is compiled with -fsanitize=thread (the device code as usual), linked -- with clang++, not
#   hipcc, so that no real HIP runtime comes in -- against tools/tsan/hip_stub.cpp, a HIP with no device (kernels do nothing,
#   "device" memory is host memory), and driven by tools/tsan/driver.cpp: the file-ingest pipeline (reader + upload | inflate |
#   count stage threads, kmer_count.hip), the framing pool, prediction's counting, two contexts on two threads, the error path.
# Builds into a scratch directory; the in-tree library is not touched.  usage: tools/tsan_host.sh [scratch-dir]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$(mktemp -d)}
mkdir -p "$OUT/files"
cd "$ROOT"
CLANG=/opt/rocm/lib/llvm/bin/clang++
SRCS=$(sed -n 's/^SRCS := //p' phenotypeseeker_amd/csrc/Makefile | sed 's/\.hip//g')
for f in $SRCS; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xarch_host -fsanitize=thread \
        -c phenotypeseeker_amd/csrc/$f.hip -o "$OUT/$f.o" &
    while [ "$(jobs -r | wc -l)" -ge 8 ]; do sleep 0.5; done
done
wait
OBJS=""
for f in $SRCS; do OBJS="$OBJS $OUT/$f.o"; done
$CLANG -O1 -g -std=c++17 -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c tools/tsan/hip_stub.cpp -o "$OUT/hip_stub.o"
$CLANG -O1 -g -std=c++17 -fsanitize=thread -c tools/tsan/driver.cpp -o "$OUT/driver.o"
$CLANG -fsanitize=thread -o "$OUT/tsan_driver" "$OUT/driver.o" $OBJS "$OUT/hip_stub.o" -lz -lpthread -ldl
TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 exitcode=66" "$OUT/tsan_driver" "$OUT/files" 2> "$OUT/tsan.log" || { tail -60 "$OUT/tsan.log"; echo "host TSAN pass FAILED ($OUT/tsan.log)"; exit 1; }
if grep -q "WARNING: ThreadSanitizer" "$OUT/tsan.log"; then tail -60 "$OUT/tsan.log"; echo "host TSAN pass FAILED ($OUT/tsan.log)"; exit 1; fi
rm -rf "$OUT/files"
echo "host TSAN pass ok ($OUT)"
