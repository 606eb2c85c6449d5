"""Sweep of the `ties` case of tests/test_gpu_summary.py::test_adjusted_pvalues_vs_oracle over its 1000 possible seeds (the
test used hash(case), salted per process): which seeds disagree with the oracle, and where."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import nbglm_oracle as orc  # noqa: E402
from pydeseq2_amd import summary as sm  # noqa: E402
from pydeseq2_amd._lib import Context  # noqa: E402

ctx = Context(0)
bad = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1000):
    rng = np.random.default_rng(seed)
    G = 5000
    bm = 10 ** rng.uniform(-1, 4, G)
    p = rng.uniform(0, 1, G) ** np.where(bm > 50, 6, 1.2)
    p[rng.random(G) < 0.03] = np.nan
    p = np.round(p, 3)
    bm = np.round(bm, 0)
    padj, info = sm.adjusted_pvalues(ctx, bm, p, 0.05, True)
    ref, rinfo = orc.independent_filtering(bm, p, 0.05)
    dc = np.nonzero(info["cutoffs"] != rinfo["cutoffs"])[0]
    dn = np.nonzero(info["num_rej"] != rinfo["num_rej"])[0]
    dt = np.nonzero(info["theta"] != rinfo["theta"])[0]
    if len(dc) or len(dn) or len(dt):
        bad.append(seed)
        print("seed", seed, "cutoffs differ at", dc[:5], [(float(info["cutoffs"][i]).hex(), float(rinfo["cutoffs"][i]).hex()) for i in dc[:3]],
              "theta differ at", dt[:5], "num_rej differ at", dn[:5], [(int(info["num_rej"][i]), int(rinfo["num_rej"][i])) for i in dn[:3]],
              flush=True)
print("bad seeds:", bad)
