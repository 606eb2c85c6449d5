"""Whole-pass reproducibility on the general kernels under different overlap settings."""
import sys, os
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _mixed_case
from pydeseq2_amd import DeseqPipeline

counts, X = _mixed_case(5, 4, 200, 3000, 12, ())
counts = counts.copy(); counts[:, 17] = 0; counts[3, 40:44] = 150000
for label, setup in (("default", lambda p: None), ("no overlap", lambda p: setattr(p, "overlap", False)),
                     ("robust late", lambda p: setattr(p, "_robust_late", True)),
                     ("robust early", lambda p: setattr(p, "_robust_early", True))):
    pipe = DeseqPipeline(counts, X, device=0)
    pipe._lfc_overlap = False
    setup(pipe)
    vals = []
    for it in range(10):
        r = pipe.deseq2()
        vals.append((float(r.genewise_dispersions[40]), float(r.mom_dispersions[40]), float(r.genewise_converged[40])))
    print(label, sorted(set(vals)))
    pipe.close()
