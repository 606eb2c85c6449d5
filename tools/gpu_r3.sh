#!/bin/bash
# round 3: GPU tests + bench lines (usage: bash tools/gpu_r3.sh <tag> [notest] [configs...])
set -u
REPO=$(pwd)
TAG=${1:-r03_a}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${2:-}" != "notest" ]; then
( time timeout 1500 python -m pytest tests -m gpu -q -n 4 ) > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | cut -c1-300
cp gpurun_out/parity_vs_reference_*.json "$OUT/" 2>/dev/null
fi
run() { # name, env, args
  env $2 timeout 900 python bench.py $3 > "$OUT/$1.log" 2> "$OUT/$1.err"
  echo "== $1"; python - "$OUT/$1.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["ms_per_step"], "ms/step  first", d.get("first_call_ms"), " h2d", d["h2d_ms"], " k_alpha full", d["roofline"]["full_launch_ms"], " parity", (d.get("parity") or {}).get("ok"), (d.get("parity") or {}).get("n_noise_genes"), (d.get("parity") or {}).get("max_rel"))
    print("  ", d["roofline"]["kernel_ms_per_step"])
    print("   cpu", d.get("cpu_baseline"))
    if "parity_c4" in d: print("   parity_c4", d["parity_c4"]["ok"], d["parity_c4"]["n_noise_genes"], d["parity_c4"]["max_rel"])
except Exception as e:
    print("no bench line:", e)
PY
  tail -3 "$OUT/$1.err"
}
shift; shift
for cfg in "$@"; do
  case $cfg in
    c3) run bench_c3 "A=1" "--config c3 --steps 20 --warmup 3" ;;
    c4) run bench_c4 "A=1" "--config c4 --steps 10 --warmup 2 --cpu-sample 2000 --no-extras" ;;
    c2) run bench_c2 "A=1" "--config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras" ;;
    c5s) run bench_c5shard "A=1" "--config c5 --genes 7500 --steps 5 --warmup 1 --no-cpu-baseline --no-extras" ;;
  esac
done
