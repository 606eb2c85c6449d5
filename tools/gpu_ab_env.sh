#!/bin/bash
# A/B of environment switches on the headline configuration (or "$CFG"): bash tools/gpu_ab_env.sh <tag> "ENV_A" "ENV_B" ... (each twice, interleaved)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout 600 python bench.py ${CFG:---config c3} --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $OUT/ab_${i}_$rep.log 2> $OUT/ab_${i}_$rep.err
    python - "$OUT/ab_${i}_$rep.log" "$e" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:40s} {d['ms_per_step']:8.3f} ms/step  full_launch {r['full_launch_ms']:.4f} (genewise {r['full_launch_ms_genewise_only']}) syncs/step {d.get('host_syncs_per_step')}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
  done
done
