# Soak for rare hangs: the default bench, the forked c5 shard, the multi-rank launches, each under its own time limit
mkdir -p gpurun_out/r06
for i in $(seq 1 ${N_DEFAULT:-8}); do timeout 300 python bench.py > gpurun_out/r06/soak_d$i.log 2> gpurun_out/r06/soak_d$i.err; echo "default $i rc=$?"; done
for i in $(seq 1 ${N_SHARD:-15}); do timeout 120 python bench.py --config c5 --genes 7500 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06/soak_s$i.log 2> gpurun_out/r06/soak_s$i.err; echo "c5 shard $i rc=$?"; done
for i in $(seq 1 ${N_RANKS:-4}); do DSQ_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 8 --config c5 --genes 24000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r06/soak_r$i.log 2> gpurun_out/r06/soak_r$i.err; echo "8 ranks $i rc=$?"; done
