"""Stage-by-stage disagreement between the engine (GPU) and the oracle on one configuration's gene slice.

Test infrastructure (needs a GPU; the oracle is the checker).  For every stage of the path the largest and median
relative difference over the genes whose convergence flags agree, then the decisive experiment for the MAP stage:
the ORACLE's MAP fit repeated from the ENGINE's trend values and prior (so that both sides start every per-gene fit
from the same numbers) - what remains is the per-gene kernels' own disagreement, what disappears was inherited
from the cross-gene trend fit.

    python tests/tools/stage_diff.py c2 2000 > gpurun_out/stage_diff_c2.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nbglm_oracle as orc  # noqa: E402


def rel(a, b, floor=1e-300):
    a, b = np.atleast_1d(np.asarray(a, float)), np.atleast_1d(np.asarray(b, float))
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.abs(a - b) / np.maximum(np.abs(b), floor)
    d[np.isnan(a) & np.isnan(b)] = 0.0
    return d


def stats(d, ok):
    d = d[ok]
    return {"max": float(d.max()), "median": float(np.median(d)), "p99": float(np.quantile(d, 0.99)),
            "n_beyond_1e-7": int((d > 1e-7).sum()), "n_beyond_1e-6": int((d > 1e-6).sum())}


def main():
    import pydeseq2_amd
    from pydeseq2_amd.synth import synth_counts

    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n_genes = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    shapes = {"c2": (20000, 200, "2level", 1), "c3": (60000, 1000, "2level", 2), "c4": (60000, 500, "3factor", 3)}
    G_cfg, N, design, seed = shapes[cfg]
    counts, X = synth_counts(G_cfg, N, design, seed)
    counts = np.ascontiguousarray(counts[:, :n_genes])
    n_jobs = min(os.cpu_count() or 1, 64)
    ref = orc.deseq2(counts, X, n_jobs=n_jobs, keep_layers=True)
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, device=0)
    pipe.collect_nfev = True
    res = pipe.deseq2()
    nz = ref.non_zero
    with np.errstate(invalid="ignore"):
        same = nz & (res.genewise_converged == ref.genewise_converged) & (res.MAP_converged == ref.MAP_converged) \
            & (res.refitted == ref.refitted)
    out = {"config": cfg, "genes": int(n_genes), "samples": N, "n_flag_or_refit_disagreements": int((nz & ~same).sum()),
           "stages": {
               "size_factors": {"max": float(rel(res.size_factors, ref.size_factors).max())},
               "normed_means": stats(rel(res.normed_means, ref.normed_means), same),
               "mom_dispersions": stats(rel(res.mom_dispersions, ref.mom_dispersions), same),
               "genewise_dispersions": stats(rel(res.genewise_dispersions, ref.genewise_dispersions), same),
               "trend_coeffs": [float(x) for x in rel(res.trend_coeffs, ref.trend_coeffs)],
               "prior_disp_var": float(rel(res.prior_disp_var, ref.prior_disp_var)[0]),
               "squared_logres": float(rel(res.squared_logres, ref.squared_logres)[0]),
               "fitted_dispersions": stats(rel(res.fitted_dispersions, ref.fitted_dispersions), same),
               "MAP_dispersions": stats(rel(res.MAP_dispersions, ref.MAP_dispersions), same),
               "dispersions": stats(rel(res.dispersions, ref.dispersions), same),
               "LFC": stats(rel(res.LFC, ref.LFC, 1e-3).max(axis=1), same),
               "lfcSE": stats(rel(res.lfcSE, ref.lfcSE), same),
               "stat": stats(rel(res.stat, ref.stat, 1e-3), same),
               "pvalue": stats(rel(res.pvalue, ref.pvalue), same)}}
    # ---- the oracle's MAP fit from the ENGINE's trend values / prior
    nzi = np.nonzero(nz)[0]
    c_nz = counts[:, nzi]
    max_disp = float(max(10.0, N))
    mu_hat = ref.mu_hat[:, nzi]
    mp, mconv = orc.alpha_mle(c_nz, X, mu_hat, res.fitted_dispersions[nzi], 1e-8, max_disp,
                              prior_disp_var=float(res.prior_disp_var), cr_reg=True, prior_reg=True, n_jobs=n_jobs)
    mp_full = np.full(len(nz), np.nan)
    mp_full[nzi] = np.clip(mp, 1e-8, max_disp)
    mc_full = np.full(len(nz), np.nan)
    mc_full[nzi] = mconv
    with np.errstate(invalid="ignore"):
        same2 = same & (mc_full == res.MAP_converged)
    out["oracle_MAP_from_engine_trend_and_prior"] = {
        "what": "oracle.alpha_mle(prior_reg=True) started from the engine's fitted dispersions with the engine's prior variance, "
                "against the engine's MAP dispersions: the per-gene kernels on identical inputs",
        "MAP_dispersions": stats(rel(res.MAP_dispersions, mp_full), same2),
        "n_flag_disagreements": int((same & ~same2).sum())}
    d = rel(res.MAP_dispersions, ref.MAP_dispersions)
    d[~same] = 0
    k = int(np.argmax(d))
    out["worst_MAP_gene"] = {"gene": k, "rel": float(d[k]), "engine": float(res.MAP_dispersions[k]),
                             "oracle": float(ref.MAP_dispersions[k]), "oracle_from_engine_trend": float(mp_full[k]),
                             "fitted_engine": float(res.fitted_dispersions[k]), "fitted_oracle": float(ref.fitted_dispersions[k]),
                             "genewise_engine": float(res.genewise_dispersions[k]), "genewise_oracle": float(ref.genewise_dispersions[k]),
                             "abs_z": float(abs(ref.stat[k]))}
    pv = rel(res.pvalue, ref.pvalue)
    pv[~same] = 0
    k = int(np.nanargmax(pv))
    out["worst_pvalue_gene"] = {"gene": k, "rel_p": float(pv[k]), "abs_z": float(abs(ref.stat[k])),
                                "rel_dispersion": float(rel(res.dispersions, ref.dispersions)[k]),
                                "rel_MAP_vs_oracle_from_engine_trend": float(rel(res.MAP_dispersions, mp_full)[k]),
                                "rel_stat": float(rel(res.stat, ref.stat)[k]), "rel_lfc": float(rel(res.LFC, ref.LFC, 1e-3)[k].max())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
