#!/bin/bash
# Profile the bench on the GPU box: one rocprofv3 --kernel-trace --stats run of the bench command of ONE configuration
# (--no-extras: the other configurations of the default run would mix their launches into the per-kernel table) plus separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a single step.
#   usage (through gpurun):  bash tools/profile_round.sh <tag> [config]
# Outputs under gpurun_out/<tag>/ ; tools/profile_collect.py turns them into profiles/<tag>_<config>.txt
set -u
TAG=${1:-r01_x}
CFG=${2:-c3}
shift 2 2>/dev/null || true   # anything else goes to bench.py (e.g. --genes 7500)
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- \
    python "$REPO/bench.py" --config "$CFG" --steps 5 --warmup 2 --no-extras "$@" > "$OUT/bench_prof.log" 2> "$OUT/bench_prof.err"
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -o run -- \
        python "$REPO/bench.py" --config "$CFG" --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > "$OUT/pmc_$C.log" 2>&1
done
cd "$REPO"
for f in $(find "$OUT" -name "*_results.db"); do echo "$f"; done
DB=$(find "$OUT/stats" -name "*_results.db" | head -1)
python tools/rocprof_summary.py stats "$DB" > "$OUT/stats.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
    DBC=$(find "$OUT/pmc_$C" -name "*_results.db" | head -1)
    python tools/rocprof_summary.py pmc "$DBC" $C "$OUT/pmc_$C.json" > "$OUT/pmc_$C.txt" 2>&1
done
# keep the merge-back small: the sqlite traces are not needed once summarised
find "$OUT" -name "*_results.db" -delete
tail -1 "$OUT/bench_prof.log" | cut -c1-400
