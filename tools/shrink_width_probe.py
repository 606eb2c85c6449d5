#!/usr/bin/env python3
"""apeGLM shrinkage time by design width (no bench configuration has 9 ... 12 columns):

    python tools/shrink_width_probe.py [G] [N]          # DSQ_LIB=build/libdeseq_hip_<variant>.so for an A/B

A one-factor + one-factor + continuous-covariate design of p columns for p in (5, 8, 10, 12), G genes x N samples of NB
counts; the pipeline's lfc_shrink of the last coefficient (all genes), second call timed."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import pydeseq2_amd  # noqa: E402
from pydeseq2_amd.summary import lfc_shrink  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    rng = np.random.default_rng(5)
    for p in ([int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else (5, 8, 10, 12)):
        i = np.arange(N)
        a, b = i % 2, (i // 2) % 4
        cols = [np.ones(N), a == 1] + [(b == k) for k in (1, 2, 3)]
        while len(cols) < p:
            cols.append(rng.normal(0, 0.6, N))
        X = np.column_stack([np.asarray(v, dtype=float) for v in cols])
        mu = np.exp(rng.normal(4.0, 1.2, G))[None, :] * np.exp(0.3 * X[:, 1:2] * rng.normal(0, 1, G)[None, :])
        disp = 0.05 + 2.0 / mu.mean(0)
        counts = rng.negative_binomial(1.0 / disp[None, :], 1.0 / (1.0 + mu * disp[None, :])).astype(np.int64)
        pipe = pydeseq2_amd.DeseqPipeline(counts, X, device=0)
        res = pipe.deseq2()
        lfc_shrink(pipe, res, p - 1)
        pipe.ctx.sync()
        t0 = time.perf_counter()
        out = lfc_shrink(pipe, res, p - 1)
        pipe.ctx.sync()
        dt = time.perf_counter() - t0
        print(f"p={p:2d}  {G} genes x {N} samples  lfc_shrink {1e3 * dt:8.2f} ms  converged {np.nanmean(out[2]):.4f}", flush=True)
        pipe.close()


if __name__ == "__main__":
    main()
