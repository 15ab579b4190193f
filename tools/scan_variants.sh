#!/bin/bash
# Builds libpsk variants with different scan-kernel knobs (for A/B runs on one GPU box).
# usage: tools/scan_variants.sh   -> gpurun_out is scratch, so variants go to tools/_variants/ (git-ignored .so)
set -e
cd "$(dirname "$0")/../phenotypeseeker_amd/csrc"
mkdir -p ../../tools/_variants
OBJS="api.o scan.o radix_sort.o kmer_count.o presence.o solver.o minhash.o"
make -s -j8 >/dev/null
for v in "4 8 1" "8 8 1" "2 8 1" "4 16 1" "4 4 1" "4 8 0" "8 16 1" "6 8 1"; do
  set -- $v
  out=../../tools/_variants/libpsk_u$1_g$2_nt$3.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPSK_SC_UNROLL=$1 -DPSK_SC_GRID_MULT=$2 -DPSK_SC_NT=$3 -c assoc_scan.hip -o /tmp/assoc_$1_$2_$3.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $OBJS /tmp/assoc_$1_$2_$3.o
  echo built $out
done
