#!/bin/bash
# round 2, GPU run 2: cell path + fused LFC epilogue + fused mu_hat + grid-kernel fix; tests, bench, profile
set -u
REPO=$(pwd)
TAG=${1:-r02_b}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -n 4 ) > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | cut -c1-200
timeout 900 python bench.py --config c3 --steps 20 --warmup 3 > "$OUT/bench_c3.log" 2> "$OUT/bench_c3.err"
timeout 600 python bench.py --config c4 --steps 10 --warmup 2 --cpu-sample 2000 --no-extras > "$OUT/bench_c4.log" 2> "$OUT/bench_c4.err"
DSQ_NO_CELL_PATH=1 timeout 600 python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/bench_c4_nocell.log" 2> "$OUT/bench_c4_nocell.err"
DSQ_NO_OVERLAP=1 timeout 600 python bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_c3_nooverlap.log" 2> "$OUT/bench_c3_nooverlap.err"
timeout 600 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_c2.log" 2> "$OUT/bench_c2.err"
timeout 600 python bench.py --config c5 --genes 7500 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5shard.log" 2> "$OUT/bench_c5shard.err"
for f in bench_c3 bench_c3_nooverlap bench_c4 bench_c4_nocell bench_c2 bench_c5shard; do echo "== $f"; python - "$OUT/$f.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["ms_per_step"], "ms/step  h2d", d["h2d_ms"], " k_alpha full", d["roofline"]["full_launch_ms"], " parity", (d.get("parity") or {}).get("ok"), (d.get("parity") or {}).get("n_noise_genes"), (d.get("parity") or {}).get("max_rel"))
    print("  ", d["roofline"]["kernel_ms_per_step"])
    if "parity_c4" in d: print("   parity_c4", d["parity_c4"]["ok"], d["parity_c4"]["n_noise_genes"], d["parity_c4"]["max_rel"])
except Exception as e:
    print("no bench line:", e)
PY
tail -3 "$OUT/$f.err"; done
bash tools/profile_round.sh $TAG c3 > "$OUT/profile_round.log" 2>&1
tail -3 "$OUT/profile_round.log"
