#!/bin/bash
# rocprofv3 kernel-trace statistics (and, with PMC=1, separate FETCH_SIZE / WRITE_SIZE counter passes) of one python
# program, on the GPU box (through gpurun, from the repo root):
#   tools/prof.sh <tag> tools/cold_probe.py [args...]      ->  gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write}
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
PROG=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$PROG" "$@" > "$OUT/run.log" 2> "$OUT/trace.err"
if [ "${PMC:-0}" = "1" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 "$PROG" "$@" > /dev/null 2> "$OUT/pmc_fetch.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- python3 "$PROG" "$@" > /dev/null 2> "$OUT/pmc_write.err"
fi
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("%-60s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for r in rows[:40]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0][:60]
        print("%-60s %8s %12.1f %10.2f %6s" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    g = glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not g:
        continue
    acc = {}
    for r in csv.DictReader(open(g[0])):
        if r["Counter_Name"] == ctr:
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    print("\n%s (KB as reported; gfx950: double FETCH_SIZE for wide reads)" % ctr)
    for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%-60s %8d  mean %12.1f KB" % (k, c, s / c))
PY
