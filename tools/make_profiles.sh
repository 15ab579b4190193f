#!/bin/bash
# Regenerates the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh r03
# Writes gpurun_out/profiles_<tag>/<workload>/: kernel-trace stats, the program's own output, and (bench, cfg3slab,
# ingest) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; no tracing flags with counters).  tools/summarise_profiles.py
# turns that into the committed files under profiles/.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
prof() {   # name pmc(0/1) program args...
  local name=$1 pmc=$2; shift 2
  local d=$OUT/$name; mkdir -p "$d"
  # (every run under its own time limit: a hung profile must not hold the GPU box until gpurun's limit)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -o run -- python3 "$@" > "$d/stdout.txt" 2> "$d/trace.err"
  if [ "$pmc" = "1" ]; then
    timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/pmc_fetch" -o run -- python3 "$@" > /dev/null 2> "$d/pmc_fetch.err"
    timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/pmc_write" -o run -- python3 "$@" > /dev/null 2> "$d/pmc_write.err"
  fi
}
BENCH="--steps 50 --warmup 5 --no-cpu-baseline"
python3 "$ROOT/bench.py" $BENCH > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
prof bench 1 "$ROOT/bench.py" $BENCH
prof ingest 1 "$ROOT/tools/profile_workloads.py" ingest
prof cfg3slab 1 "$ROOT/tools/profile_workloads.py" cfg3slab
prof moments 0 "$ROOT/tools/profile_workloads.py" moments
prof fastq 0 "$ROOT/tools/profile_workloads.py" fastq
prof solver 0 "$ROOT/tools/profile_workloads.py" solver
prof solver4096 0 "$ROOT/tools/profile_workloads.py" solver4096
prof predict 0 "$ROOT/tools/profile_workloads.py" predict
prof weights 0 "$ROOT/tools/profile_workloads.py" weights
prof lasso 0 "$ROOT/tools/profile_workloads.py" lasso
du -sh "$OUT"
