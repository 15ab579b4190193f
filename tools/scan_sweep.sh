#!/bin/bash
# chi2 scan of a device-generated matrix far beyond the Infinity Cache (23.5 M rows x 2048 samples = 6 GB, config 3's
# slab shape): grid multiplier (env) x unroll (rebuild) sweep.  usage (GPU box): tools/scan_sweep.sh
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT/phenotypeseeker_amd/csrc"
run() { python3 "$ROOT/bench.py" --workload matrix --rows 23500000 --samples 2048 --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   kernel_ms %.4f  frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for unr in 4 2 8; do
  touch assoc_scan.hip; make -s EXTRA="-DPSK_SC_UNROLL=$unr" > /dev/null 2>&1 || { echo "build failed unroll $unr"; continue; }
  for gm in 4 8 16 32 64; do
    echo "unroll $unr grid_mult $gm"; PSK_GRID_MULT=$gm run
  done
done
touch assoc_scan.hip; make -s > /dev/null 2>&1
