#!/bin/bash
# samples the GPU clock / power while a command runs: bash tools/probes/clock_watch.sh <out.txt> <cmd...>
OUT=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT &
W=$!
"$@"
kill $W
python - $OUT <<'PY'
import re, sys, statistics
s, p = [], []
for l in open(sys.argv[1]):
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if m: s.append(int(m.group(1)))
    m = re.search(r"Power \(W\): ([\d.]+)", l)
    if m: p.append(float(m.group(1)))
if s: print("sclk MHz: n", len(s), "min", min(s), "median", statistics.median(s), "max", max(s))
if p: print("power W : n", len(p), "min", min(p), "median", statistics.median(p), "max", max(p))
PY
