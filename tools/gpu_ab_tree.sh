# A/B of two source trees on one box: the working tree vs build/wt_old (git worktree of an older commit with its own built library)
for rep in 1 2 3; do
  for t in . build/wt_old; do
    ( cd $t && timeout 300 python bench.py --config ${CFG:-c3} --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['roofline']['full_launch_ms'])" )
  done
done
