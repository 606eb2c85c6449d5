#!/bin/bash
# A/B of library builds (DSQ_LIB) on one configuration: bash tools/gpu_ab_lib.sh <tag> "<bench args>" lib_a.so lib_b.so ... (each twice, interleaved)
TAG=$1; ARGS=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  i=0
  for lib in "$@"; do
    i=$((i+1))
    DSQ_LIB=$lib timeout 900 python bench.py $ARGS --no-extras --no-cpu-baseline > $OUT/lib_${i}_$rep.log 2> $OUT/lib_${i}_$rep.err
    python - "$OUT/lib_${i}_$rep.log" "$lib" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]; k = r["kernel_ms_per_step"]
    print(f"{sys.argv[2][-34:]:34s} {d['ms_per_step']:8.3f} ms/step  full_launch {r['full_launch_ms']:.4f}  " + " ".join(f"{a}={k[a]}" for a in ("irls_mu", "alpha_mle", "alpha_map", "lfc_fit", "robust_disp", "mom_lin_mu") if a in k))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
  done
done
