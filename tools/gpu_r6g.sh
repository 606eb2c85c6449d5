for m in "" "DSQ_CU_SPLIT_MODE=1" "DSQ_CU_SPLIT=16" "DSQ_CU_SPLIT_MODE=1 DSQ_CU_SPLIT=16" ""; do
  env $m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']; print('[$m]', d['ms_per_step'], 'trend', k['trend_fit'], 'prior', k['prior_mad'], 'robust', k['robust_disp'])"
done
python -m pytest tests/test_refsuite_facade.py tests/test_gpu_facade.py -q -m gpu 2>&1 | tail -2
