set -u
timeout 600 python -m pytest tests -m gpu -x -q -k "robust or trimmed or bucket or mix or benchmark_shapes or cooks or refit or replace" 2>&1 | tail -3
bash tools/gpu_ab_lib.sh cntc3 "--config c3 --steps 30 --warmup 5" build/libdeseq_hip_batched.so pydeseq2_amd/libdeseq_hip.so
bash tools/gpu_ab_lib.sh cntc5 "--config c5 --genes 7500 --steps 8 --warmup 3" build/libdeseq_hip_batched.so pydeseq2_amd/libdeseq_hip.so
