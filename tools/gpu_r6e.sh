for v in pydeseq2_amd/libdeseq_hip.so build/libdeseq_hip_same.so build/libdeseq_hip_fexp.so pydeseq2_amd/libdeseq_hip.so build/libdeseq_hip_same.so build/libdeseq_hip_fexp.so; do
  DSQ_LIB=$v python bench.py --config c5 --genes 7500 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']; print('$v', d['ms_per_step'], k['irls_mu'], k['lfc_fit'], k['alpha_mle'], k['alpha_map'])"
done
python tools/probes/digest_bench.py
