set -u
mkdir -p gpurun_out/r06b
python tests/tools/stage_diff.py c2 2000 > gpurun_out/r06b/stage_diff_c2_2000.json 2> gpurun_out/r06b/sd1.err; echo sd1 $?
python tests/tools/stage_diff.py c2 20000 > gpurun_out/r06b/stage_diff_c2_20000.json 2> gpurun_out/r06b/sd2.err; echo sd2 $?
python tests/tools/stage_diff.py c3 8000 > gpurun_out/r06b/stage_diff_c3_8000.json 2> gpurun_out/r06b/sd3.err; echo sd3 $?
for c in c2 c4; do bash tools/pmc_flops.sh r06b $c > /dev/null 2>&1; echo flops $c $?; done
bash tools/pmc_flops.sh r06b c5 --genes 7500 > /dev/null 2>&1; echo flops c5s $?
( time python bench.py ) > gpurun_out/r06b/bench_default.json 2> gpurun_out/r06b/bench_default.err; echo bench $?
tail -c 300 gpurun_out/r06b/bench_default.err
