for rep in 1 2 3; do for ov in 1 0; do
DSQ_LFC_OVERLAP=$ov timeout 600 python bench.py --config c4 --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c4 overlap=$ov', d['ms_per_step'])"
done; done
for rep in 1 2; do for ov in 1 0; do
DSQ_LFC_OVERLAP=$ov timeout 600 python bench.py --config c4 --genes 15000 --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c4 15000 overlap=$ov', d['ms_per_step'])"
done; done
for rep in 1 2; do for ov in 1 0; do
DSQ_LFC_OVERLAP=$ov timeout 600 python bench.py --config c5 --genes 15000 --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c5 15000 overlap=$ov', d['ms_per_step'])"
done; done
for rep in 1 2; do for ov in 1 0; do
DSQ_LFC_OVERLAP=$ov timeout 600 python bench.py --config c5 --genes 30000 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c5 30000 overlap=$ov', d['ms_per_step'])"
done; done
