for v in build/libdeseq_hip_csu1.so build/libdeseq_hip_csu2.so build/libdeseq_hip_csu1.so build/libdeseq_hip_csu2.so; do
  for c in c3 c2; do DSQ_LIB=$v python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/err_$c.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']; print('$v $c', d['ms_per_step'], 'lfc', k['lfc_fit'])" || tail -3 gpurun_out/err_$c.txt; done
done
