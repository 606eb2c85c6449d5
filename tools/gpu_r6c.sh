set -u
mkdir -p gpurun_out/r06c
python -m pytest tests/test_refsuite_facade.py tests/test_gpu_facade.py -q -m gpu > gpurun_out/r06c/facade.log 2>&1; tail -25 gpurun_out/r06c/facade.log
python -m pytest tests/ -q -m gpu --deselect tests/test_refsuite_facade.py --deselect tests/test_gpu_facade.py > gpurun_out/r06c/gpu_tests.log 2>&1; tail -15 gpurun_out/r06c/gpu_tests.log
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/r06c/bench_quick.json 2> gpurun_out/r06c/bench_quick.err; cut -c1-300 gpurun_out/r06c/bench_quick.json
