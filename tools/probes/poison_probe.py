"""Does the rescued gene of test_rescued_gene_is_reproducible_from_pass_to_pass read workspace entries it has not written?
Run with DSQ_LIB=build/libdeseq_hip_poison.so (make variant NAME=poison SRC=dsq_k_irls DEFS=-DDSQ_LBFGSB_POISON: the
workspace starts as NaNs instead of zeros) and with the production library, and compare."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _mixed_case
from pydeseq2_amd import DeseqPipeline

counts, X = _mixed_case(5, 4, 200, 3000, 12, ())
counts = counts.copy(); counts[3, 40:44] = 150000
pipe = DeseqPipeline(counts, X, device=0)
r = pipe.deseq2()
print("genewise", r.genewise_dispersions[38:45])
print("LFC[40]", r.LFC[40], "converged", r.LFC_converged[38:45])
