set -u
mkdir -p gpurun_out/r06f
python -m pytest tests/ -q -m gpu -x > gpurun_out/r06f/gpu_tests.log 2>&1; tail -3 gpurun_out/r06f/gpu_tests.log
python tools/probes/plugin_probe.py c3 > gpurun_out/r06f/plugin_c3.json 2> gpurun_out/r06f/plugin_c3.err; python -c "
import json; d=json.load(open('gpurun_out/r06f/plugin_c3.json')); print('first', d['first']); print('repeat', d['repeat']); print(d['cache'])"
for i in 1 2; do python bench.py --config c5 --genes 7500 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c5 shard', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; done
