#!/bin/bash
# fp64 instruction-mix counter pass (its own run: --kernel-trace + --pmc only; 8 SQ slots per pass on gfx950):
#   bash tools/pmc_flops.sh <tag> <config> [extra bench args]   ->  gpurun_out/<tag>/flops_<config>.json
# tools/flops_json.py turns the rocpd database into the per-kernel table bench.py's companion roofline reads
# (profiles/flops_<config>.json): wave-level instruction counts of ADD / MUL / FMA / TRANS on float64, all VALU
# instructions, and the SQ busy / VALU-active cycles of the largest launch of every kernel.
set -u
REPO=$(pwd); TAG=$1; CFG=$2; shift 2
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d "$OUT/fl_$CFG" -o run -- \
    python "$REPO/bench.py" --config "$CFG" --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > "$OUT/fl_$CFG.log" 2>&1
cd "$REPO"
DB=$(find "$OUT/fl_$CFG" -name "*_results.db" | head -1)
python tools/flops_json.py "$DB" "$OUT/fl_$CFG.log" "$CFG" "$@" > "$OUT/flops_$CFG.json" 2> "$OUT/flops_$CFG.err"
find "$OUT" -name "*_results.db" -delete
head -c 1500 "$OUT/flops_$CFG.json"
