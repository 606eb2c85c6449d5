"""The reference against ITSELF: how far do the final outputs of the whole path move when mu_hat moves by one ulp?

Test infrastructure, CPU only.  The oracle (bit-identical to the unmodified reference at these shapes, tests/golden/
kat_e2e_*.npz) runs deseq2() + Wald twice on the same matrix: as is, and with the mu_hat of the genewise stage multiplied
by (1 + eps N(0,1)), eps = 1.1e-16 - the smallest disagreement two correct implementations of `lin_reg_mu` can have.
The two results are compared with bench.py's own `parity_report`, i.e. exactly as the engine is compared with the
oracle: success-flag flips, max relative differences of dispersions / LFC / statistic on the genes whose flags agree,
the RAW Wald p-value difference and the number of genes beyond a raw 1e-5.  Whatever these numbers are, no engine can
be asked to agree with the reference more closely than the reference agrees with itself.

    python tests/tools/pvalue_floor.py c3 8000      ->  profiles/r06_pvalue_floor_c3.json
    python tests/tools/pvalue_floor.py c2 20000     ->  profiles/r06_pvalue_floor_c2.json
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (parity_report: the comparison the bench line and the tests use)
from oracle import nbglm_oracle as orc  # noqa: E402
from pydeseq2_amd.synth import synth_counts  # noqa: E402


class PerturbedInference(orc._OracleInference):
    def __init__(self, n_jobs, eps, seed):
        super().__init__(n_jobs)
        self.eps, self.rng = eps, np.random.default_rng(seed)

    def lin_reg_mu(self, counts, size_factors, design_matrix, min_mu):
        mu = super().lin_reg_mu(counts, size_factors, design_matrix, min_mu)
        return mu * (1.0 + self.eps * self.rng.standard_normal(mu.shape))

    def irls(self, counts, size_factors, design_matrix, disp, min_mu, beta_tol, **kw):
        b, mu, h, c = super().irls(counts, size_factors, design_matrix, disp, min_mu, beta_tol, **kw)
        if not getattr(self, "_mu_hat_done", False):  # only the mu_hat fit of the genewise stage (dds.py:757-765)
            self._mu_hat_done = True
            mu = mu * (1.0 + self.eps * self.rng.standard_normal(mu.shape))
        return b, mu, h, c


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    genes = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    shapes = {"c2": (20000, 200, "2level", 1), "c3": (60000, 1000, "2level", 2), "c4": (60000, 500, "3factor", 3),
              "c5": (60000, 5000, "mixed", 4)}
    G_cfg, N, design, seed = shapes[cfg]
    n_jobs = min(os.cpu_count() or 1, 64)
    if cfg == "c5":
        from pydeseq2_amd.synth import synth_counts_block

        counts, X = synth_counts_block(G_cfg, N, design, seed, genes=(0, genes))
    else:
        counts, X = synth_counts(G_cfg, N, design, seed)
        counts = np.ascontiguousarray(counts[:, :genes])
    out = {"config": cfg, "genes": genes, "samples": N,
           "what": "oracle (= the reference, bit for bit) vs the oracle with mu_hat * (1 + eps N(0,1)): bench.parity_report of the two",
           "levels": []}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = orc.deseq2(counts, X, n_jobs=n_jobs, keep_layers=False)
        for eps, sd in ((1.1e-16, 11), (1.1e-16, 12), (1e-14, 13)):
            pert = orc.deseq2(counts, X, n_jobs=n_jobs, keep_layers=False, inference=PerturbedInference(n_jobs, eps, sd))
            rep = bench.parity_report(pert, base)
            lev = {"eps": eps, "seed": sd, "n_flag_flips": rep["n_noise_genes"], "max_rel_same_flags": rep["max_rel"],
                   "raw_pvalue": rep["raw_pvalue"], "trend_coeffs_max_rel": rep.get("trend_coeffs_max_rel"),
                   "would_pass_bench_parity": rep["ok"]}
            out["levels"].append(lev)
            print(json.dumps(lev), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
