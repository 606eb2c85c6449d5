"""Developer tool (test infrastructure): stage-by-stage max relative differences GPU pipeline vs oracle.

    python tests/debug_parity.py <genes> <samples> <2level|3factor|mixed> <seed>
"""
import sys

import numpy as np

sys.path.insert(0, ".")
import pydeseq2_amd
from oracle import nbglm_oracle as orc

G, N, design, seed = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
counts, X = orc.synth_counts(G, N, design, seed)
pipe = pydeseq2_amd.DeseqPipeline(counts, X, device=0)
res = pipe.deseq2(profile=True)
ref = orc.deseq2(counts, X, n_jobs=8)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    ok = ~np.isnan(b) & ~np.isnan(a)
    d = np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1e-300)
    return f"max {d.max():.2e} p99 {np.quantile(d, 0.99):.2e} median {np.median(d):.2e} n>1e-5 {(d > 1e-5).sum()}"


for f in ["size_factors", "normed_means", "mom_dispersions", "genewise_dispersions", "trend_coeffs",
          "fitted_dispersions", "MAP_dispersions", "dispersions", "LFC", "lfcSE", "stat", "pvalue"]:
    print(f"{f:22s}", rel(getattr(res, f), getattr(ref, f)))
print("squared_logres", res.squared_logres, ref.squared_logres, "prior_var", res.prior_disp_var, ref.prior_disp_var)
for f in ["genewise_converged", "MAP_converged", "LFC_converged", "replaced", "refitted", "cooks_outlier"]:
    a, b = np.nan_to_num(getattr(res, f)), np.nan_to_num(getattr(ref, f))
    print(f"{f:22s} mismatches {(a != b).sum()}  (gpu false: {(a == 0).sum()}, oracle false: {(b == 0).sum()})")
print("timings", {k: round(v * 1e3, 2) for k, v in res.timings.items()})
