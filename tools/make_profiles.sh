#!/bin/bash
# Regenerates the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh r05 [workload ...]
# Writes gpurun_out/profiles_<tag>/<workload>/: kernel-trace stats, the program's own output, and one directory per --pmc
# pass (counters are collected in passes of their own, never together with a tracing flag).  tools/summarise_profiles.py
# turns that into the committed files under profiles/.
#   pass names: fetch = FETCH_SIZE, write = WRITE_SIZE, rdreq / wrreq = the raw L2 -> fabric request counters FETCH_SIZE /
#   WRITE_SIZE are derived from (how many requests, how many of them short), hit = L2 hits / misses, lds = the SQ's LDS
#   counters (bank-conflict cycles, LDS-active cycles, LDS instructions, issue stalls on the LDS) with the wave-cycle split
set -u
TAG=${1:-r06}
shift || true
ONLY="$*"
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pmc_of() {
  case $1 in
    fetch) echo "FETCH_SIZE";;
    write) echo "WRITE_SIZE";;
    rdreq) echo "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum";;
    wrreq) echo "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum";;
    hit)   echo "TCC_HIT_sum TCC_MISS_sum";;
    lds)   echo "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY";;
  esac
}
wanted() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }
prof() {   # name "pass pass ..." program args...      (the program itself after `--`: no env / bash -c hop under the profiler)
  local name=$1 passes=$2; shift 2
  wanted "$name" || return 0
  local d=$OUT/$name; mkdir -p "$d"
  # (every run under its own time limit: a hung profile must not hold the GPU box until gpurun's limit)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -o run -- "$@" > "$d/stdout.txt" 2> "$d/trace.err"
  for p in $passes; do
    timeout 400 rocprofv3 --pmc $(pmc_of $p) --output-format csv -d "$d/pmc_$p" -o run -- "$@" > /dev/null 2> "$d/pmc_$p.err"
  done
}
BENCH="--steps 50 --warmup 5 --no-cpu-baseline --no-hbm-only"   # (--no-hbm-only: the headline kernel's average covers the headline matrix only; `hbmonly` below is the 4.3-GB matrix)
if wanted bench; then python3 "$ROOT/bench.py" $BENCH 2> "$OUT/bench_unprofiled.err" | tail -1 > "$OUT/bench_unprofiled.json"; fi   # (r06: the line is printed after every leg; the last one is the record)
prof calib "fetch write rdreq wrreq hit" "$ROOT/tools/calib/pmc_calib" 3
prof bench "fetch write" python3 "$ROOT/bench.py" $BENCH
prof hbmonly "fetch write" python3 "$ROOT/bench.py" --workload matrix --rows 134375000 --steps 50 --warmup 5 --no-cpu-baseline --no-hbm-only --no-e2e   # r06: the same kernel beyond the Infinity Cache
prof ingest "fetch write rdreq wrreq" python3 "$ROOT/tools/profile_workloads.py" ingest
prof cfg3slab "fetch write rdreq wrreq" python3 "$ROOT/tools/profile_workloads.py" cfg3slab
prof moments "fetch write lds" python3 "$ROOT/tools/profile_workloads.py" moments
prof fastq "" python3 "$ROOT/tools/profile_workloads.py" fastq
prof fastqwrapped "" python3 "$ROOT/tools/profile_workloads.py" fastqwrapped           # r06: FASTQ that is not four lines a record, device against host framing
prof gzinflate "fetch write lds" python3 "$ROOT/tools/gz_bench.py" fastq 64 128 6 3 noverify          # r05: 64 x 144 MB of .fastq.gz text inflated on the device
prof cfg5gz "" python3 "$ROOT/tools/profile_workloads.py" cfg5gz 64                      # ... and `phenotypeseeker modeling` on 64 of them, end to end (two minutes of generating first)
prof fastqgz "" python3 "$ROOT/tools/profile_workloads.py" fastqgz 16                   # ... and 16 config-5 samples counted from .fastq.gz files
prof solver "" python3 "$ROOT/tools/profile_workloads.py" solver
prof solver4096 "" python3 "$ROOT/tools/profile_workloads.py" solver4096
prof predict "" python3 "$ROOT/tools/profile_workloads.py" predict
prof weights "" python3 "$ROOT/tools/profile_workloads.py" weights
prof lasso "" python3 "$ROOT/tools/profile_workloads.py" lasso
for K in 16 21 31; do
  prof kslab$K "" python3 "$ROOT/tools/profile_workloads.py" kwide $K slab      # counting under a rank's 1/8 slab filter, 256 genomes
  prof kwhole$K "" python3 "$ROOT/tools/profile_workloads.py" kwide $K whole    # ... the whole word space, 64 genomes
done
prof cfg3slab21 "" python3 "$ROOT/tools/profile_workloads.py" cfg3slab 21
prof cfg3slab31 "" python3 "$ROOT/tools/profile_workloads.py" cfg3slab 31
du -sh "$OUT"
