mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_launches" 2>&1 | tail -5
CFG="--config c5" bash tools/gpu_ab_env.sh r06/lfc_c5 "DSQ_LFC_OVERLAP=1" "DSQ_LFC_OVERLAP=0" 2>&1 | grep -v workload
CFG="--config c4 --genes 7500" bash tools/gpu_ab_env.sh r06/lfc_c4s "DSQ_LFC_OVERLAP=1" "DSQ_LFC_OVERLAP=0" 2>&1 | grep -v workload
CFG="--config c5 --genes 7500" bash tools/gpu_ab_env.sh r06/lfc_c5g "DSQ_NO_ALPHA_MIX=1 DSQ_LFC_OVERLAP=1" "DSQ_NO_ALPHA_MIX=1 DSQ_LFC_OVERLAP=0" 2>&1 | grep -v workload
