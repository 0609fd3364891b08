#!/bin/bash
# Kernel-trace statistics and HBM traffic (PMC, separate passes as MI355X_MICROARCH.md prescribes) of the bench step.
#   bash tools/profile_round.sh <tag> <k> [extra bench args]       (on the GPU box; results under gpurun_out/prof_<tag>)
set -u
TAG=$1; K=$2; shift 2
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--k $K --no-cpu-baseline --no-gfa --no-e2e --no-k55 --no-meta $*"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 bench.py --steps 3 --warmup 1 $ARGS > "$OUT/bench_under_profiler.json" 2> "$OUT/stats.err"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch -- python3 bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write -- python3 bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> "$OUT/write.err"
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1); W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py "$F" "$W" "$OUT/pmc_traffic.json" > "$OUT/pmc_summary.txt" 2>&1
cat "$OUT/pmc_summary.txt"
head -12 "$OUT/kernel_stats.csv" | cut -c1-200
# keep the merge small: the raw traces stay on the box
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
