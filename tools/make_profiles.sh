#!/bin/bash
# Regenerates the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh r01
# Writes gpurun_out/profiles_<tag>/: kernel-trace stats of `python3 bench.py`, the bench line under the profiler
# and unprofiled, and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; no tracing flags with counters).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline"
python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$ROOT/bench.py" $ARGS \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 "$ROOT/bench.py" $ARGS \
    > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python3 "$ROOT/bench.py" $ARGS \
    > /dev/null 2> "$OUT/pmc_write.err"
find "$OUT" -name "*.csv" | head -20
