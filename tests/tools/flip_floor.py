"""How many dispersion fits change their L-BFGS-B ``success`` flag when mu_hat moves by an ulp?  (build container or GPU box;
CPU only)

The reference returns whatever iterate scipy's L-BFGS-B stops at, and the grid-quantised value when the line search
ends in the rounding noise of the loss (utils.py:556-564).  This script measures that sensitivity on the reference's own
arithmetic (the oracle's per-gene fit is bit-identical to utils.fit_alpha_mle, tests/test_oracle_golden.py): the
genewise fits of the first genes of the c3 golden case are repeated with mu_hat * (1 + eps * N(0,1)) for several eps,
and the genes whose flag or value changes are counted.  Any implementation whose loss differs from the reference's in
the last bits - another summation order, another lgamma - sits at some point of this curve.

    python tests/tools/flip_floor.py [genes]      ->  profiles/r03_flip_floor.json      (c3 shape: 1000 samples, two groups)
    python tests/tools/flip_floor.py 1000 c5      ->  profiles/r03_flip_floor_c5.json   (c5 shape: 5000 samples, continuous
                                                covariates, mu_hat from the IRLS fit)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nbglm_oracle as orc  # noqa: E402
from pydeseq2_amd.synth import synth_counts  # noqa: E402


def main():
    genes = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    n_jobs = min(os.cpu_count() or 1, 64)
    case = sys.argv[2] if len(sys.argv) > 2 else "c3"
    samples = 5000 if case == "c5" else 1000
    counts, X = synth_counts(genes, samples, "mixed", 4) if case == "c5" else synth_counts(genes, 1000, "2level", 2)
    sf, normed, _, _ = orc.size_factors_ratio(counts)
    mom = orc.mom_dispersions(normed, X, sf, 1e-8, 1000.0)
    if case == "c5":
        mu = np.maximum(orc.irls(counts, sf, X, mom, n_jobs=n_jobs)[1], 0.5)
    else:
        mu = orc.lin_reg_mu(counts, sf, X, 0.5)
    a0, c0 = orc.alpha_mle(counts, X, mu, mom, 1e-8, 1000.0, n_jobs=n_jobs)
    out = {"_note": "genewise dispersion fits (utils.fit_alpha_mle arithmetic) of the golden case repeated with "
                    "mu_hat * (1 + eps * N(0,1)); flips = genes whose scipy success flag changed",
           "case": case, "genes": genes, "samples": samples, "non_converged_unperturbed": int((~c0).sum()), "levels": []}
    rng = np.random.default_rng(0)
    for eps in (1.1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-10):
        mu_p = mu * (1.0 + eps * rng.standard_normal(mu.shape))
        a, c = orc.alpha_mle(counts, X, mu_p, mom, 1e-8, 1000.0, n_jobs=n_jobs)
        same = c == c0
        rel = np.abs(a - a0) / a0
        lev = {"eps": eps, "flips": int((~same).sum()), "flip_rate": round(float((~same).mean()), 6),
               "max_rel_same_flag": float(rel[same].max()), "genes_beyond_1e-5_same_flag": int((rel[same] > 1e-5).sum()),
               "max_rel_flipped": float(rel[~same].max()) if (~same).any() else 0.0}
        out["levels"].append(lev)
        print(lev, flush=True)
    path = os.path.join(ROOT, "profiles", "r03_flip_floor.json" if case == "c3" else f"r03_flip_floor_{case}.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
