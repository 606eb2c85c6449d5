#!/bin/bash
# A/B of environment switches on one box: bash tools/gpu_ab_env.sh <tag> <config> "<ENV=val ...>" "<ENV=val ...>" ...
set -u
TAG=$1; CFG=$2; shift; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for E in "$@"; do
  env $E timeout 300 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/ab_$i.log" 2> "$OUT/ab_$i.err"
  python - "$OUT/ab_$i.log" "$E" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], "->", d["ms_per_step"], "ms/step; k_alpha", d["roofline"]["full_launch_ms"], d["roofline"]["kernel_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
  tail -2 "$OUT/ab_$i.err"
  i=$((i+1))
done
