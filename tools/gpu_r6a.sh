set -u
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sixteen_bit_boundary or mixed_design_pipeline_every_shape" > gpurun_out/r06a/test_boundary.log 2>&1; tail -3 gpurun_out/r06a/test_boundary.log
python tests/tools/stage_diff.py c2 2000 > gpurun_out/r06a/stage_diff_c2_2000.json 2> gpurun_out/r06a/sd1.err; echo sd1 $?
python tests/tools/stage_diff.py c2 20000 > gpurun_out/r06a/stage_diff_c2_20000.json 2> gpurun_out/r06a/sd2.err; echo sd2 $?
python tests/tools/stage_diff.py c3 8000 > gpurun_out/r06a/stage_diff_c3_8000.json 2> gpurun_out/r06a/sd3.err; echo sd3 $?
bash tools/pmc_flops.sh r06a c3 > gpurun_out/r06a/pmc_flops_c3.out 2>&1; echo flops $?
