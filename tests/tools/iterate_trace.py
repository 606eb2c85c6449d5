"""Where does the ~1e-6 dispersion disagreement between the engine and the reference come from?

Test infrastructure (CPU only): runs the genewise dispersion fit of a slice of a benchmark configuration through
  * the oracle's `alpha_mle_gene` (scipy L-BFGS-B, the reference's arithmetic, utils.py:441-564) and
  * the HOST instantiation of the engine's per-gene templates (tests/hostsim: the same `fit_alpha_gene` +
    `dsq_lbfgsb1d.h` the device kernels instantiate),
keeps every objective evaluation of both sides (x, f, g), and classifies each gene:

  same_path      same number of evaluations, every evaluation point agrees to 1e-9: the two sides walked the same
                 iterates and stop at the same one; the final difference is the propagated rounding of f and g
  eval_apart     the evaluation counts differ (one side stopped an iteration earlier / took another line-search
                 trial): the stopping test or a line-search test was decided inside the rounding noise of f
  flag_differs   the `success` flags differ (one side went to the grid)

and prints the distribution of |d alpha| / alpha per class, plus the full trace of the worst gene.

    python tests/tools/iterate_trace.py [c2|c3] [genes] > profiles/r06_iterate_trace_c2.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nbglm_oracle as orc  # noqa: E402
from scipy.optimize import minimize  # noqa: E402
from tests import hostsim as hs  # noqa: E402


def scipy_trace(y, X, mu, ah, min_disp, max_disp):
    """alpha_mle_gene's objective with every evaluation recorded."""
    la_hat = np.log(ah)
    ev = []

    def loss(la):
        alpha = np.exp(la)
        W = mu / (1 + mu * alpha)
        return orc.nb_nll(y, mu, alpha) + 0.5 * np.linalg.slogdet((X.T * W) @ X)[1]

    def dloss(la):
        alpha = np.exp(la)
        W = mu / (1 + mu * alpha)
        dW = -(W ** 2)
        rg = (0.5 * (np.linalg.inv((X.T * W) @ X) * ((X.T * dW) @ X)).sum()) * alpha
        return alpha * orc.dnb_nll(y, mu, alpha) + rg

    def f(x):
        v = loss(x[0])
        ev.append([float(x[0]), float(v), None])
        return v

    def g(x):
        v = dloss(x[0])
        if ev and ev[-1][0] == float(x[0]) and ev[-1][2] is None:
            ev[-1][2] = float(v)
        return np.asarray([v])

    with np.errstate(all="ignore"):
        res = minimize(f, x0=np.asarray([la_hat]), jac=g, method="L-BFGS-B",
                       bounds=[(np.log(min_disp), np.log(max_disp))])
    return float(res.x[0]), bool(res.success), int(res.nfev), int(res.nit), ev


def host_trace(y, X, mu, ah, min_disp, max_disp):
    """The engine's optimiser (dsq_lbfgsb1d.h, host build) on the ENGINE's objective (hs.alpha_eval), recorded."""
    ev = []

    def fg(la):
        f, g = hs.alpha_eval(y, mu, X, la)
        ev.append([float(la), float(f), float(g)])
        return f, g

    x, f, ok, nfev, nit, st = hs.lbfgsb1d(fg, float(np.log(ah)), float(np.log(min_disp)), float(np.log(max_disp)))
    return float(x), bool(ok), int(nfev), int(nit), ev


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n_genes = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    shapes = {"c2": (20000, 200, 1), "c3": (60000, 1000, 2)}
    G_cfg, N, seed = shapes[cfg]
    from pydeseq2_amd.synth import synth_counts

    counts, X = synth_counts(G_cfg, N, "2level", seed)
    counts = np.ascontiguousarray(counts[:, :n_genes])
    sf = orc.size_factors_ratio(counts)[0]
    nz = counts.sum(0) > 0
    c = counts[:, nz]
    normed = c / sf[:, None]
    min_disp, max_disp = 1e-8, float(max(10.0, N))
    mom = orc.mom_dispersions(normed, X, sf, min_disp, max_disp)
    mu = orc.lin_reg_mu(c, sf, X, 0.5)
    rows = []
    for j in range(c.shape[1]):
        y = c[:, j].astype(float)
        xs, oks, nfs, nits, evs = scipy_trace(y, X, mu[:, j], mom[j], min_disp, max_disp)
        xh, okh, nfh, nith, evh = host_trace(c[:, j], X, mu[:, j], mom[j], min_disp, max_disp)
        rel = abs(np.exp(xh) - np.exp(xs)) / np.exp(xs)
        if oks != okh:
            cls = "flag_differs"
        elif nfs != nfh or len(evs) != len(evh):
            cls = "eval_apart"
        else:
            dx = max(abs(a[0] - b[0]) for a, b in zip(evs, evh))
            cls = "same_path" if dx <= 1e-9 else "eval_apart"
        rows.append((j, cls, rel, nfs, nfh, nits, nith, evs, evh, oks, okh))
    out = {"config": cfg, "genes": int(c.shape[1]), "samples": N,
           "what": "genewise dispersion fit: scipy L-BFGS-B on the reference's objective vs the engine's optimiser and objective "
                   "(host instantiation of the device templates), both from the same start and the same mu_hat",
           "classes": {}}
    for cls in ("same_path", "eval_apart", "flag_differs"):
        r = np.array([x[2] for x in rows if x[1] == cls])
        out["classes"][cls] = {"genes": int(len(r)),
                               "max_rel_alpha": float(r.max()) if len(r) else None,
                               "median_rel_alpha": float(np.median(r)) if len(r) else None,
                               "n_beyond_1e-7": int((r > 1e-7).sum()), "n_beyond_1e-6": int((r > 1e-6).sum())}
    conv = [x for x in rows if x[1] != "flag_differs"]
    worst = max(conv, key=lambda x: x[2])
    j, cls, rel, nfs, nfh, nits, nith, evs, evh, oks, okh = worst
    out["worst_converged_gene"] = {
        "index_among_nonzero": int(j), "class": cls, "rel_alpha": float(rel), "nfev_scipy": nfs, "nfev_engine": nfh,
        "nit_scipy": nits, "nit_engine": nith,
        "evaluations_scipy_x_f_g": evs, "evaluations_engine_x_f_g": evh,
        "f_difference_at_common_points": [float(a[1] - b[1]) for a, b in zip(evs, evh) if abs(a[0] - b[0]) < 1e-12]}
    worst_same = max((x for x in conv if x[1] == "same_path"), key=lambda x: x[2], default=None)
    if worst_same is not None:
        j, cls, rel, nfs, nfh, nits, nith, evs, evh, oks, okh = worst_same
        out["worst_same_path_gene"] = {
            "index_among_nonzero": int(j), "rel_alpha": float(rel), "nfev": nfs,
            "x_differences_per_evaluation": [float(a[0] - b[0]) for a, b in zip(evs, evh)],
            "g_relative_difference_per_evaluation": [float((a[2] - b[2]) / max(abs(a[2]), 1e-300)) if a[2] is not None else None
                                                     for a, b in zip(evs, evh)],
            "g_scipy": [a[2] for a in evs], "x_scipy": [a[0] for a in evs]}
    print(json.dumps(out, indent=1))


def propagate(cfg="c2", n_genes=2000, n_jobs=8, n_flip=1, flip_rel=2.0e-3, noise=0.0):
    """Second experiment: how far does ONE success-flag flip (a genewise dispersion replaced by its grid-quantised value,
    ~2e-3 away) move every OTHER gene's final dispersion and Wald p-value through the all-gene trend fit and prior?
    Oracle only: the genewise dispersions of `n_flip` genes are multiplied by (1 + flip_rel) before the trend fit and the
    rest of the path (trend -> prior -> MAP -> IRLS -> Wald) is repeated."""
    shapes = {"c2": (20000, 200, 1), "c3": (60000, 1000, 2)}
    G_cfg, N, seed = shapes[cfg]
    from pydeseq2_amd.synth import synth_counts

    counts, X = synth_counts(G_cfg, N, "2level", seed)
    counts = np.ascontiguousarray(counts[:, :n_genes])
    p = X.shape[1]
    min_mu, min_disp, max_disp, beta_tol = 0.5, 1e-8, float(max(10.0, N)), 1e-8
    sf, normed, _, _ = orc.size_factors_ratio(counts)
    nz = ~(counts == 0).all(0)
    c = counts[:, nz]
    nm = normed[:, nz].mean(0)
    inf = orc._OracleInference(n_jobs)
    mom, mu_hat, gw, gconv = orc._fit_genewise(c, normed[:, nz], sf, X, min_mu, min_disp, max_disp, beta_tol, n_jobs, inf)

    def tail(gw_in):
        coeffs, _ = orc.fit_parametric_trend(gw_in, nm, glm=inf.dispersion_trend_gamma_glm)
        fitted = coeffs[0] + coeffs[1] / nm
        sq, pv = orc.dispersion_prior(gw_in, fitted, N, p, min_disp)
        mp, _ = inf.alpha_mle(counts=c, design_matrix=X, mu=mu_hat, alpha_hat=fitted, min_disp=min_disp, max_disp=max_disp,
                              prior_disp_var=float(pv), cr_reg=True, prior_reg=True)
        disp = np.clip(mp, min_disp, max_disp)
        out_g = np.log(gw_in) > np.log(fitted) + 2 * np.sqrt(sq)
        disp[out_g] = gw_in[out_g]
        beta, mu, _, _ = inf.irls(counts=c, size_factors=sf, design_matrix=X, disp=disp, min_mu=min_mu, beta_tol=beta_tol)
        con = np.zeros(p)
        con[-1] = 1.0
        pval, st, se = inf.wald_test(design_matrix=X, disp=disp, lfc=beta, mu=mu, ridge_factor=np.diag(np.repeat(1e-6, p)),
                                     contrast=con, lfc_null=0.0, alt_hypothesis=None)
        return coeffs, float(pv), disp, st, pval

    base = tail(gw)
    rng = np.random.default_rng(7)
    # a flip happens on converged mid-range genes; pick genes near the median dispersion
    cand = np.argsort(np.abs(np.log(gw) - np.median(np.log(gw))))[:200]
    picks = rng.choice(cand, n_flip, replace=False) if n_flip else np.zeros(0, int)
    gw2 = gw.copy()
    gw2[picks] *= 1.0 + flip_rel
    if noise > 0:  # rounding-level disagreement of every genewise dispersion (what two correct implementations differ by)
        gw2 *= 1.0 + noise * rng.standard_normal(len(gw2))
    pert = tail(gw2)
    others = np.ones(len(gw), bool)
    others[picks] = False
    rel_d = np.abs(pert[2] - base[2]) / base[2]
    with np.errstate(invalid="ignore", divide="ignore"):
        rel_p = np.abs(pert[4] - base[4]) / np.maximum(base[4], 1e-300)
    z = np.abs(base[3])
    k = int(np.nanargmax(np.where(others, rel_p, 0)))
    return {"experiment": "one flip propagated through trend and prior (oracle only)", "config": cfg,
            "genes": int(len(gw)), "n_flipped": int(n_flip), "flip_rel": flip_rel, "noise_on_every_genewise_dispersion": noise,
            "trend_coeffs_rel_change": [float(abs(a - b) / abs(b)) for a, b in zip(pert[0], base[0])],
            "prior_var_rel_change": float(abs(pert[1] - base[1]) / base[1]),
            "other_genes_max_rel_dispersion_change": float(rel_d[others].max()),
            "other_genes_median_rel_dispersion_change": float(np.median(rel_d[others])),
            "other_genes_max_rel_pvalue_change": float(np.nanmax(rel_p[others])),
            "abs_z_of_that_gene": float(z[k]),
            "other_genes_n_pvalue_beyond_1e-5": int(np.nansum(rel_p[others] > 1e-5)),
            "other_genes_max_pvalue_change_per_z2": float(np.nanmax((rel_p / np.maximum(1.0, z ** 2))[others]))}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "propagate":
        cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
        ng = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
        nf = int(sys.argv[4]) if len(sys.argv) > 4 else 1
        noise = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
        print(json.dumps(propagate(cfg, ng, n_flip=nf, noise=noise), indent=1))
    else:
        main()
