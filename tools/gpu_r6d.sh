set -u
mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "mixed or many_samples or benchmark_shapes or plugin_cache" > gpurun_out/r06d/gpu_tests.log 2>&1; tail -3 gpurun_out/r06d/gpu_tests.log
for i in 1 2; do python bench.py --config c5 --genes 7500 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c5 shard', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; done
python bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c5 full', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
