#!/bin/bash
# round 2, GPU run 4: tests (wide + sample-shard protocol), occupancy variants of the cell kernels at c4
set -u
REPO=$(pwd)
TAG=${1:-r02_d}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -n 4 ) > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | cut -c1-200
run() { # name, env, args
  env $2 timeout 600 python bench.py $3 > "$OUT/$1.log" 2> "$OUT/$1.err"
  echo "== $1"; python - "$OUT/$1.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["ms_per_step"], "ms/step  h2d", d["h2d_ms"], " k_alpha full", d["roofline"]["full_launch_ms"], " parity", (d.get("parity") or {}).get("ok"), (d.get("parity") or {}).get("n_noise_genes"), (d.get("parity") or {}).get("max_rel"))
    print("  ", d["roofline"]["kernel_ms_per_step"])
except Exception as e:
    print("no bench line:", e)
PY
  tail -3 "$OUT/$1.err"
}
run bench_c4 "A=1" "--config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
run bench_c4_cw3 "DSQ_LIB=$REPO/build/libdeseq_hip_cw3.so" "--config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
run bench_c4_cw4 "DSQ_LIB=$REPO/build/libdeseq_hip_cw4.so" "--config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c4" -o run -- python "$REPO/bench.py" --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/prof_c4.log" 2>&1
cd "$REPO"
DB=$(find "$OUT/stats_c4" -name "*_results.db" | head -1)
python tools/rocprof_summary.py stats "$DB" > "$OUT/stats_c4.txt" 2>&1
find "$OUT" -name "*_results.db" -delete
head -30 "$OUT/stats_c4.txt"
