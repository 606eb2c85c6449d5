"""Stress of the LFC fit in two launches: fork vs single launch, bit for bit, over mixed-design shapes, seeds and repeated passes."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _mixed_case
from pydeseq2_amd import DeseqPipeline

F = ("genewise_dispersions", "MAP_dispersions", "MAP_converged", "dispersions", "outlier_genes", "LFC", "LFC_converged", "lfcSE",
     "stat", "pvalue", "cooks_outlier", "replaced", "refitted")
shapes = [(8, 3, (2, 4), 1400, 3000), (5, 2, (3,), 600, 5000), (3, 1, (2,), 300, 9000), (7, 1, (2, 5), 900, 2500),
          (6, 2, (4,), 2560, 2100), (4, 3, (), 260, 4000)]
n_bad = 0
for (P, Q, lv, N, G) in shapes:
    for seed in (1, 2, 3):
        counts, X = _mixed_case(P, Q, N, G, 1000 * P + seed, lv)
        counts = counts.copy()
        counts[:, seed] = 0
        counts[seed, 10:14] = 70000 + seed  # a few genes beyond the 16-bit staging: general kernel beside the row kernel
        counts[N // 2, 20:22] = 40000
        pipe = DeseqPipeline(counts, X, device=0)
        assert pipe._row_mode == 3
        pipe._lfc_overlap = False
        ref = pipe.deseq2()
        ref = {f: np.array(getattr(ref, f), copy=True) for f in F}
        pipe._lfc_overlap = True
        f0 = pipe.lfc_forks
        for it in range(4):
            r = pipe.deseq2()
            for f in F:
                a, b = np.asarray(getattr(r, f), float), np.asarray(ref[f], float)
                if not ((a == b) | (np.isnan(a) & np.isnan(b))).all():
                    n_bad += 1
                    print("MISMATCH", (P, Q, lv, N, G), seed, it, f)
        assert pipe.lfc_forks == f0 + 4
        mc = np.asarray(r.MAP_converged)
        assert set(np.unique(mc[~np.isnan(mc)])) <= {0.0, 1.0}, np.unique(mc)
        pipe.close()
    print("shape", (P, Q, lv, N, G), "done")
print("mismatches:", n_bad)
