"""Developer aid for A/B builds of the library (DSQ_LIB=...): run deseq2() on a seeded matrix and dump the
result vectors, so that two builds can be compared bit for bit.
    DSQ_LIB=build/libdeseq_hip_x.so python tools/ab_dump.py out.npz [G N design]
    python tools/ab_dump.py --cmp a.npz b.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if sys.argv[1] == "--cmp":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
        print("identical" if not bad else f"DIFFERENT: {bad}")
        return
    import pydeseq2_amd
    from pydeseq2_amd.synth import synth_counts

    G, N, design = (int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else (3000, 120, "3factor")
    counts, X = synth_counts(G, N, design, 5)
    r = pydeseq2_amd.DeseqPipeline(counts, X, device=0).deseq2()
    np.savez(sys.argv[1], sf=r.size_factors, gw=r.genewise_dispersions, disp=r.dispersions, lfc=r.LFC, p=r.pvalue,
             se=r.lfcSE, trend=np.asarray(r.trend_coeffs), prior=np.array([r.prior_disp_var]))


if __name__ == "__main__":
    main()
