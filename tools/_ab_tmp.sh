set -u
timeout 900 python -m pytest tests -m gpu -x -q -k "grid or alpha or disp or kat or golden" 2>&1 | tail -3
bash tools/gpu_ab_lib.sh gridc5 "--config c5 --genes 7500 --steps 10 --warmup 3" build/libdeseq_hip_base2.so pydeseq2_amd/libdeseq_hip.so
bash tools/gpu_ab_lib.sh gridc3 "--config c3 --steps 30 --warmup 5" build/libdeseq_hip_base2.so pydeseq2_amd/libdeseq_hip.so
