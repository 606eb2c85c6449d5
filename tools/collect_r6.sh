#!/bin/bash
# gpurun_out/ of tools/gpu_final_r6.sh -> the files committed under profiles/ (run in the build container, after the GPU call)
set -eu
cd "$(dirname "$0")/.."
python tools/profile_collect.py r06 c3
python tools/profile_collect.py r06c4 c4 && mv profiles/r06c4_c4.txt profiles/r06_c4.txt
PROFILE_BENCH_ARGS="--genes 7500" python tools/profile_collect.py r06c5s c5 && mv profiles/r06c5s_c5.txt profiles/r06_c5_shard.txt
cp gpurun_out/r06_gaps/timeline_c3.txt profiles/r06_timeline_c3.txt
cp gpurun_out/r06_gaps/gaps_c3.txt profiles/r06_idle_gaps_c3.txt
cp gpurun_out/r06_gaps7500/timeline_c3.txt profiles/r06_timeline_c3_shard7500.txt
cp gpurun_out/r06_gapsc5s/timeline_c5.txt profiles/r06_timeline_c5_shard.txt
cp gpurun_out/r06_sq/sq_c3.txt profiles/r06_sq_counters_c3.txt
cp gpurun_out/r06_sq/sq_c5.txt profiles/r06_sq_counters_c5_shard.txt
for c in c2 c3 c4 c5; do cp gpurun_out/r06_fl/flops_$c.json profiles/flops_$c.json; done
for a in c2_2000 c2_20000 c3_8000; do cp gpurun_out/r06/stage_diff_$a.json profiles/r06_stage_diff_$a.json; done
cp gpurun_out/r06/plugin_c3.json profiles/r06_plugin_path_c3.json
cp gpurun_out/r06/plugin_c4.json profiles/r06_plugin_path_c4.json
head -1 gpurun_out/r06/bench_default.json > profiles/r06_bench_default.json
git status --short profiles | head -30
sed -i 's#profiles/r06c4_c4.txt#profiles/r06_c4.txt#; s#profiles/r06c5s_c5.txt#profiles/r06_c5_shard.txt#' profiles/traffic_c4.json profiles/traffic_c5.json
