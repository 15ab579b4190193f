#!/bin/bash
# A/B of the table-in-LDS moment scans on the GPU box: rebuilds assoc_scan.o with -D overrides and runs the moments workload
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd "$ROOT/phenotypeseeker_amd/csrc"
for v in "" "$@"; do
  touch assoc_scan.hip
  make -s EXTRA="$v" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== variant [$v]"
  python3 "$ROOT/tools/profile_workloads.py" moments | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v,3) for k,v in d['notes']['event_ms'].items()})"
done
touch assoc_scan.hip; make -s > /dev/null 2>&1
