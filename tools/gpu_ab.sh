#!/bin/bash
# A/B of library variants on the bench configurations: bash tools/gpu_ab.sh <tag> "<configs>" <variant names...>  ("main" = the in-tree library)
set -u
REPO=$(pwd); TAG=$1; CFGS=$2; shift 2
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
for V in "$@"; do
  for CFG in $CFGS; do
    EXTRA=""; [ "$CFG" = "c5" ] && EXTRA="--genes 7500"
    LIB=$REPO/pydeseq2_amd/libdeseq_hip.so; [ "$V" != "main" ] && LIB=$REPO/build/libdeseq_hip_$V.so
    DSQ_LIB=$LIB timeout 300 python bench.py --config $CFG $EXTRA --steps 8 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/${V}_$CFG.log" 2> "$OUT/${V}_$CFG.err"
    python - "$OUT/${V}_$CFG.log" "$V $CFG" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["roofline"]["kernel_ms_per_step"]
    print(f"{sys.argv[2]:12s} {d['ms_per_step']:8.3f} ms/step  alpha {d['roofline']['full_launch_ms']:.3f}  irls_mu {k.get('irls_mu', 0):.3f} lfc_fit {k.get('lfc_fit', 0):.3f} robust {k.get('robust_disp', 0):.3f}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  done
done
