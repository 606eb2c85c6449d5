#!/bin/bash
# round 2, GPU run 1: full GPU test-suite, BYVAL A/B of the dispersion kernel, bench c3/c4, rocprof + PMC passes
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_a
mkdir -p "$OUT"
export TMPDIR=/tmp
nproc > "$OUT/nproc.txt"
( time timeout 1500 python -m pytest tests -m gpu -q -n 4 --deselect "tests/test_gpu_parity.py::test_inference_vs_reference_kats[p16]" --deselect "tests/test_gpu_parity.py::test_inference_vs_reference_kats[p24]" ) > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
# A/B: default vs BYVAL build, bit-for-bit
python tools/ab_dump.py "$OUT/ab_default.npz" 20000 1000 2level > "$OUT/ab.log" 2>&1
DSQ_LIB=$REPO/build/libdeseq_hip_byval.so python tools/ab_dump.py "$OUT/ab_byval.npz" 20000 1000 2level >> "$OUT/ab.log" 2>&1
python tools/ab_dump.py --cmp "$OUT/ab_default.npz" "$OUT/ab_byval.npz" >> "$OUT/ab.log" 2>&1
rm -f "$OUT"/ab_*.npz
cat "$OUT/ab.log"
# bench lines
timeout 900 python bench.py --config c3 --steps 20 --warmup 3 > "$OUT/bench_c3.log" 2> "$OUT/bench_c3.err"
DSQ_LIB=$REPO/build/libdeseq_hip_byval.so timeout 300 python bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_c3_byval.log" 2> "$OUT/bench_c3_byval.err"
timeout 600 python bench.py --config c4 --steps 10 --warmup 2 --cpu-sample 2000 --no-extras > "$OUT/bench_c4.log" 2> "$OUT/bench_c4.err"
timeout 600 python bench.py --config c5 --genes 7500 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5shard.log" 2> "$OUT/bench_c5shard.err"
for f in bench_c3 bench_c3_byval bench_c4 bench_c5shard; do echo "== $f"; tail -1 "$OUT/$f.log" | cut -c1-600; tail -2 "$OUT/$f.err"; done
# profiles (stats + PMC) of the default build, and the WRITE_SIZE pass of the BYVAL build
bash tools/profile_round.sh r02_a c3 > "$OUT/profile_round.log" 2>&1
cd /tmp
DSQ_LIB=$REPO/build/libdeseq_hip_byval.so timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_byval_WRITE" -o run -- \
    python "$REPO/bench.py" --config c3 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > "$OUT/pmc_byval_WRITE.log" 2>&1
cd "$REPO"
DBC=$(find "$OUT/pmc_byval_WRITE" -name "*_results.db" | head -1)
python tools/rocprof_summary.py pmc "$DBC" WRITE_SIZE "$OUT/pmc_byval_WRITE.json" > "$OUT/pmc_byval_WRITE.txt" 2>&1
find "$OUT" -name "*_results.db" -delete
grep -h "k_alpha" "$OUT/pmc_WRITE_SIZE.txt" "$OUT/pmc_byval_WRITE.txt" | head
