#!/bin/bash
# A/B of library variants on the apeGLM shrinkage (bench extras): bash tools/shrink_ab.sh <config> <variant names...>  ("main" = in-tree)
CFG=$1; shift
for V in "$@"; do
  LIB=pydeseq2_amd/libdeseq_hip.so; [ "$V" != "main" ] && LIB=build/libdeseq_hip_$V.so
  DSQ_LIB=$LIB timeout 600 python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V $CFG', d['ms_per_step'], d.get('lfc_shrink', {}).get('ms'))"
done
