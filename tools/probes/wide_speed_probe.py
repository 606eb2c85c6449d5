"""Time per deseq2() pass for designs beyond 12 columns (the LDS / matrix-core kernels): python tools/probes/wide_speed_probe.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _wide_case
from pydeseq2_amd import DeseqPipeline

for kind, G, N in (("factor16", 20000, 480), ("mixed14", 20000, 480), ("factor40", 20000, 960), ("mixed44", 20000, 960)):
    counts, X = _wide_case(kind, G, N, 3)
    pipe = DeseqPipeline(counts, X, device=0)
    pipe.deseq2()
    t0 = time.perf_counter()
    for _ in range(3):
        r = pipe.deseq2()
    dt = (time.perf_counter() - t0) / 3
    prof = pipe.deseq2(profile=True).timings
    print(f"{kind:9s} p={X.shape[1]:2d} {G} x {N}: {dt * 1e3:8.1f} ms per pass  ", {k: round(v * 1e3, 1) for k, v in prof.items() if v > 1e-3}, flush=True)
    pipe.close()
