# A/B of the LFC fit in two launches (DSQ_LFC_OVERLAP) on the configurations where it is on by default, and its tests
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_launches or reproducible or world1" 2>&1 | tail -4
CFG="--config c4" bash tools/gpu_ab_env.sh r06/lfc_c4 "DSQ_LFC_OVERLAP=1" "DSQ_LFC_OVERLAP=0" 2>&1 | grep -v workload
CFG="--config c5 --genes 7500" bash tools/gpu_ab_env.sh r06/lfc_c5s "DSQ_LFC_OVERLAP=1" "DSQ_LFC_OVERLAP=0" 2>&1 | grep -v workload
python tools/probes/repro_sweep2.py 2>&1 | tail -12
