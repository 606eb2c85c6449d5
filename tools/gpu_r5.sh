#!/bin/bash
# round 5: GPU tests + the default bench line (usage: bash tools/gpu_r5.sh <tag> [notest] [bench args...])
set -u
TAG=${1:-r05_a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${2:-}" != "notest" ]; then
( time timeout 1500 python -m pytest tests -m gpu -q -n 4 ) > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | cut -c1-400
else
shift
fi
shift
( time timeout 1500 python bench.py "$@" ) > "$OUT/bench.log" 2> "$OUT/bench.err"
tail -4 "$OUT/bench.err"
python - "$OUT/bench.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
r = d["roofline"]
print(d["config"]["workload"]); print(d["ms_per_step"], "ms/step first", d.get("first_call_ms"), "h2d", d["h2d_ms"], "full_launch", r["full_launch_ms"], "gw-only", r.get("full_launch_ms_genewise_only"), "frac", r["frac"])
print("  kernels", r["kernel_ms_per_step"])
p = d.get("parity") or {}
print("  parity", p.get("ok"), p.get("n_noise_genes"), p.get("max_rel"), p.get("raw_pvalue"))
print("  cpu", d.get("cpu_baseline", {}) and {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind")})
for k, v in (d.get("other_configs") or {}).items():
    print("  ==", k, {kk: v.get(kk) for kk in ("ms_per_step", "ms_per_step_wall", "ms_per_step_gpu_events", "ms_per_step_loop_mean", "generator_s", "error")})
    print("     stage", v.get("dispersion_stage"), "shrink", v.get("lfc_shrink"), "parity", (v.get("parity") or {}).get("ok"), (v.get("parity") or {}).get("n_noise_genes"), (v.get("parity") or {}).get("raw_pvalue"))
    print("     wall", v.get("stage_wall_ms_profiled_step"))
print("  plugin", json.dumps(d.get("plugin_path") or d.get("plugin_path_error")))
print("  shrink", d.get("lfc_shrink"), "summary", d.get("summary_tail"), d.get("extras_error"))
PY
