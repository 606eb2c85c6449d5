set -u
mkdir -p gpurun_out/r06f
python tools/probes/plugin_probe.py c3 > gpurun_out/r06f/plugin_c3.json 2> gpurun_out/r06f/plugin_c3.err; python -c "
import json; d=json.load(open('gpurun_out/r06f/plugin_c3.json')); print('first', d['first']); print('second', d['second']['total_ms']); print('repeat', d['repeat']); print(d['cache'])"
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "plugin or hip_inference" 2>&1 | tail -2
