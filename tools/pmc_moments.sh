#!/bin/bash
# SQ counters of the moment scans (tools/profile_workloads.py moments): where the waves' cycles go
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_moments
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o run -- python3 "$ROOT/tools/profile_workloads.py" moments > /dev/null 2> "$OUT/p$i.err"
done
python3 - "$OUT" <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "scan_kernel" not in k and "finalize" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
P
