set -u
timeout 600 python -m pytest tests -m gpu -x -q -k "robust or trimmed or bucket or mix or benchmark_shapes or cooks or refit" 2>&1 | tail -3
bash tools/gpu_ab_lib.sh bufc3 "--config c3 --steps 30 --warmup 5" build/libdeseq_hip_base.so pydeseq2_amd/libdeseq_hip.so
