#!/bin/bash
# SQ counter pass (separate from any trace domain): bash tools/pmc_sq.sh <tag> <config> [extra bench args]
set -u
REPO=$(pwd); TAG=$1; CFG=$2; shift 2
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d "$OUT/sq_$CFG" -o run -- \
    python "$REPO/bench.py" --config "$CFG" --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > "$OUT/sq_$CFG.log" 2>&1
cd "$REPO"
DB=$(find "$OUT/sq_$CFG" -name "*_results.db" | head -1)
for C in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT; do
  echo "## $C"; python tools/rocprof_summary.py pmc "$DB" $C | head -8
done > "$OUT/sq_$CFG.txt" 2>&1
find "$OUT" -name "*_results.db" -delete
cat "$OUT/sq_$CFG.txt"
