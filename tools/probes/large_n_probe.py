"""A data set with many samples: python tools/probes/large_n_probe.py <genes> <samples>  (two-level design, time per pass)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pydeseq2_amd import DeseqPipeline
from pydeseq2_amd.synth import synth_counts_block

G, N = int(sys.argv[1]), int(sys.argv[2])
t0 = time.perf_counter()
counts, X = synth_counts_block(G, N, "2level", 3)
print(f"generated {counts.shape} in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
pipe = DeseqPipeline(counts, X, device=0)
print(f"upload {time.perf_counter() - t0:.2f} s, row_mode {pipe._row_mode}", flush=True)
for it in range(3):
    t0 = time.perf_counter()
    r = pipe.deseq2()
    print(f"pass {it}: {1e3 * (time.perf_counter() - t0):.1f} ms; refitted {int(r.refitted.sum())}, "
          f"converged LFC {np.nanmean(r.LFC_converged):.4f}, disp median {np.nanmedian(r.dispersions):.4f}", flush=True)
print({k: round(v * 1e3, 2) for k, v in pipe.deseq2(profile=True).timings.items()})
