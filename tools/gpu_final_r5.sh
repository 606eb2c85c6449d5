#!/bin/bash
# round 5: the profiles committed under profiles/ (one call): kernel stats + FETCH/WRITE per configuration, timelines, SQ counters,
# phase probe of the mixed-design dispersion kernel, the drop-in path
set -u
export TMPDIR=/tmp
bash tools/profile_round.sh r05 c3 > /dev/null 2>&1; echo c3 done
bash tools/profile_round.sh r05c4 c4 > /dev/null 2>&1; echo c4 done
bash tools/profile_round.sh r05c5s c5 --genes 7500 > /dev/null 2>&1; echo c5 shard done
bash tools/gpu_gaps.sh r05_gaps c3 > /dev/null 2>&1; echo gaps c3 done
bash tools/gpu_gaps.sh r05_gaps7500 c3 "--genes 7500" > /dev/null 2>&1; echo gaps c3 shard done
bash tools/gpu_gaps.sh r05_gapsc5s c5 "--genes 7500" > /dev/null 2>&1; echo gaps c5 shard done
bash tools/pmc_sq.sh r05_sq c3 > /dev/null 2>&1; bash tools/pmc_sq.sh r05_sq c5 --genes 7500 > /dev/null 2>&1; echo sq done
DSQ_LIB=build/libdeseq_hip_mixph.so python tools/mix_phase_probe.py 7500 > gpurun_out/r05_mixphase.txt 2>&1; cat gpurun_out/r05_mixphase.txt
python tools/probes/plugin_probe.py c3 > gpurun_out/r05_plugin_c3.json 2>gpurun_out/r05_plugin_c3.err; tail -c 600 gpurun_out/r05_plugin_c3.json
python tools/probes/plugin_probe.py c4 > gpurun_out/r05_plugin_c4.json 2>gpurun_out/r05_plugin_c4.err; tail -c 400 gpurun_out/r05_plugin_c4.json
