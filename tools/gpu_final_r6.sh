#!/bin/bash
# round 6: everything committed under profiles/r06_* from ONE call: kernel stats + FETCH / WRITE per configuration, timelines / idle
# gaps, SQ counters, the fp64 instruction mix (flops_*.json), stage-by-stage parity, the drop-in path, the default bench line
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
bash tools/profile_round.sh r06 c3 > /dev/null 2>&1; echo c3 done
bash tools/profile_round.sh r06c4 c4 > /dev/null 2>&1; echo c4 done
bash tools/profile_round.sh r06c5s c5 --genes 7500 > /dev/null 2>&1; echo c5 shard done
bash tools/gpu_gaps.sh r06_gaps c3 > /dev/null 2>&1; echo gaps c3 done
bash tools/gpu_gaps.sh r06_gaps7500 c3 "--genes 7500" > /dev/null 2>&1; echo gaps c3 shard done
bash tools/gpu_gaps.sh r06_gapsc5s c5 "--genes 7500" > /dev/null 2>&1; echo gaps c5 shard done
bash tools/pmc_sq.sh r06_sq c3 > /dev/null 2>&1; bash tools/pmc_sq.sh r06_sq c5 --genes 7500 > /dev/null 2>&1; echo sq done
for c in c2 c3 c4; do bash tools/pmc_flops.sh r06_fl $c > /dev/null 2>&1; done; bash tools/pmc_flops.sh r06_fl c5 --genes 7500 > /dev/null 2>&1; echo flops done
for a in "c2 2000" "c2 20000" "c3 8000"; do set -- $a; python tests/tools/stage_diff.py $1 $2 > gpurun_out/r06/stage_diff_$1_$2.json 2> gpurun_out/r06/stage_diff_$1_$2.err; done; echo stage diff done
python tools/probes/plugin_probe.py c3 > gpurun_out/r06/plugin_c3.json 2> gpurun_out/r06/plugin_c3.err; tail -c 300 gpurun_out/r06/plugin_c3.json; echo
python tools/probes/plugin_probe.py c4 > gpurun_out/r06/plugin_c4.json 2> gpurun_out/r06/plugin_c4.err; tail -c 200 gpurun_out/r06/plugin_c4.json; echo
( time python bench.py ) > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err; tail -4 gpurun_out/r06/bench_default.err
