for g in 7500 20000 0; do for m in 0 1 2; do
  DSQ_TREND_GRID=$m python bench.py --genes $g --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']; print('genes $g grid $m', d['ms_per_step'], 'trend', k['trend_fit'], 'prior', k['prior_mad'])"
done; done
