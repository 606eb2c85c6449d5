#!/bin/bash
# kernel-trace of a few bench steps + idle-gap table + timeline of one step
# (usage through gpurun: bash tools/gpu_gaps.sh <tag> [config] [extra bench args])
set -u
TAG=${1:-r03_gaps}; CFG=${2:-c3}; EXTRA=${3:-}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/trace" -o run -- \
    python "$REPO/bench.py" --config "$CFG" --steps 10 --warmup 2 --no-cpu-baseline --no-extras $EXTRA > "$OUT/bench.log" 2> "$OUT/bench.err"
cd "$REPO"
DB=$(find "$OUT/trace" -name "*_results.db" | head -1)
python tools/rocprof_gaps.py "$DB" 3 > "$OUT/gaps_$CFG.txt" 2>&1
python tools/rocprof_timeline.py "$DB" > "$OUT/timeline_$CFG.txt" 2>&1
find "$OUT" -name "*_results.db" -delete
head -30 "$OUT/gaps_$CFG.txt"
