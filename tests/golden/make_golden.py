"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

* per-gene known-answer vectors: the reference's own kernels
  (pydeseq2.utils / grid_search / preprocessing / default_inference, imported
  through a 3-line on-disk shim because ``import pydeseq2`` needs anndata)
  are run on seeded inputs; inputs and outputs go to ``kat_*.npz``;
* the reference's R-DESeq2 (v1.34.0) fixtures and the shipped synthetic
  dataset are copied verbatim (they are data, not source) to ``r_*/``.

The GPU box has no /root/reference: tests only read the committed files.
"""

import os
import shutil
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    shim = tempfile.mkdtemp(prefix="refshim_")
    os.makedirs(os.path.join(shim, "pydeseq2"))
    with open(os.path.join(shim, "pydeseq2", "__init__.py"), "w") as fh:
        fh.write(
            f"__path__.append('{REF}/pydeseq2')\n"
            f"__file__ = '{REF}/pydeseq2/__init__.py'\n"
            "__version__ = '0.5.3'\n"
        )
    sys.path.insert(0, shim)
    import pydeseq2.default_inference as di
    import pydeseq2.grid_search as gs
    import pydeseq2.preprocessing as pp
    import pydeseq2.utils as ut

    return ut, gs, pp, di


def synth(G, N, X, seed):
    rng = np.random.default_rng(seed)
    p = X.shape[1]
    beta = np.zeros((p, G))
    beta[0] = rng.normal(4, 2, G)
    for j in range(1, p):
        beta[j] = rng.normal(0, 0.7, G)
    disp = 4 / np.maximum(2.0 ** beta[0], 1e-3) + 0.1
    sf = np.exp(rng.normal(0, 0.2, N))
    mu = sf[:, None] * 2.0 ** (X @ beta)
    size = 1 / disp
    return rng.negative_binomial(size[None, :], size[None, :] / (size[None, :] + mu)).astype(np.int64)


def kat_case(name, counts, X, ut, gs, pp, di, n_grid=3):
    """Run every reference routine on the hot path and dump inputs + outputs."""
    N, G = counts.shape
    p = X.shape[1]
    out = {"counts": counts, "X": X}
    normed, sf = pp.deseq2_norm(counts)
    out["sf"], out["normed"] = sf, normed
    lm, filt = pp.deseq2_norm_fit(counts)
    out["logmeans"], out["filtered"] = lm, filt
    nz = ~(counts == 0).all(axis=0)
    assert nz.all(), "generator should not emit all-zero genes for KAT cases"
    min_disp, max_disp, min_mu = 1e-8, float(max(10, N)), 0.5
    rde = ut.fit_rough_dispersions(normed, X)
    mde = ut.fit_moments_dispersions(normed, sf)
    out["rough"], out["moments"] = rde, mde
    mom = np.clip(np.minimum(rde, mde), min_disp, max_disp)
    out["mom"] = mom
    # linear mu
    out["lin_mu"] = np.stack(
        [ut.fit_lin_mu(counts[:, g], sf, X, min_mu) for g in range(G)], axis=1
    )
    # IRLS with MoM dispersions
    r = [ut.irls_solver(counts[:, g], sf, X, mom[g], min_mu, 1e-8) for g in range(G)]
    out["irls_beta"] = np.stack([x[0] for x in r])
    out["irls_mu"] = np.stack([x[1] for x in r], axis=1)
    out["irls_H"] = np.stack([x[2] for x in r], axis=1)
    out["irls_conv"] = np.array([x[3] for x in r], dtype=bool)
    n_cells = len(np.unique(X, axis=0))
    mu_hat = out["lin_mu"] if n_cells == p else out["irls_mu"]
    out["mu_hat"] = mu_hat
    # genewise alpha
    r = [
        ut.fit_alpha_mle(counts[:, g], X, mu_hat[:, g], mom[g], min_disp, max_disp)
        for g in range(G)
    ]
    gw = np.array([x[0] for x in r])
    out["gw_alpha"], out["gw_conv"] = gw, np.array([x[1] for x in r], dtype=bool)
    gwc = np.clip(gw, min_disp, max_disp)
    # trend (one call of the gamma GLM on all genes)
    means = normed.mean(0)
    infer = di.DefaultInference(n_cpus=1)
    coeffs, pred, conv = infer.dispersion_trend_gamma_glm(
        pd.Series(1 / means), pd.Series(gwc)
    )
    out["trend_coeffs"], out["trend_pred"], out["trend_conv"] = coeffs, pred, conv
    fitted = coeffs[0] + coeffs[1] / means
    out["fitted"] = fitted
    prior_var = 0.7
    out["prior_var"] = prior_var
    r = [
        ut.fit_alpha_mle(counts[:, g], X, mu_hat[:, g], fitted[g], min_disp, max_disp,
                         prior_var, True, True)
        for g in range(G)
    ]
    mp = np.array([x[0] for x in r])
    out["map_alpha"], out["map_conv"] = mp, np.array([x[1] for x in r], dtype=bool)
    disp = np.clip(mp, min_disp, max_disp)
    # final IRLS
    r = [ut.irls_solver(counts[:, g], sf, X, disp[g], min_mu, 1e-8) for g in range(G)]
    beta = np.stack([x[0] for x in r])
    out["lfc_beta"] = beta
    out["lfc_mu"] = np.stack([x[1] for x in r], axis=1)
    out["lfc_H"] = np.stack([x[2] for x in r], axis=1)
    out["lfc_conv"] = np.array([x[3] for x in r], dtype=bool)
    # grid searches on the first few genes
    out["grid_alpha"] = np.array(
        [gs.grid_fit_alpha(counts[:, g], X, mu_hat[:, g], mom[g], min_disp, max_disp)
         for g in range(n_grid)]
    )
    if p == 2:
        out["grid_beta"] = np.stack(
            [gs.grid_fit_beta(counts[:, g], sf, X, disp[g]) for g in range(n_grid)]
        )
    # robust dispersion / cooks ingredients
    ddf = pd.DataFrame(X, columns=[f"c{j}" for j in range(p)])
    out["robust_disp"] = ut.robust_method_of_moments_disp(normed, ddf)
    out["trim_mean_02"] = ut.trimmed_mean(normed, trim=0.2, axis=0)
    out["mad"] = ut.mean_absolute_deviation(np.log(gwc) - np.log(fitted))
    # nll / gradient samples
    out["nll"] = np.array([ut.nb_nll(counts[:, g], mu_hat[:, g], gwc[g]) for g in range(G)])
    out["dnll"] = np.array([ut.dnb_nll(counts[:, g], mu_hat[:, g], gwc[g]) for g in range(G)])
    # Wald, every alternative
    mu_w = np.exp(X @ beta.T) * sf[:, None]
    ridge = np.diag(np.repeat(1e-6, p))
    contrast = np.zeros(p)
    contrast[1] = 1.0
    out["contrast"] = contrast
    for alt, null in ((None, 0.0), ("greater", 0.5), ("less", -0.5), ("greaterAbs", 0.5),
                      ("lessAbs", 0.5)):
        r = [
            ut.wald_test(X, disp[g], beta[g], mu_w[:, g], ridge, contrast,
                         np.log(2) * null, alt)
            for g in range(G)
        ]
        tag = alt or "none"
        out[f"wald_p_{tag}"] = np.array([x[0] for x in r], dtype=float)
        out[f"wald_stat_{tag}"] = np.array([x[1] for x in r], dtype=float)
        out[f"wald_se_{tag}"] = np.array([x[2] for x in r], dtype=float)
    np.savez_compressed(os.path.join(HERE, f"kat_{name}.npz"), **out)
    print(name, "genes", G, "samples", N, "p", p,
          "gw non-converged", int((~out["gw_conv"]).sum()),
          "map non-converged", int((~out["map_conv"]).sum()))


def main():
    ut, gs, pp, di = _import_reference()
    # case A: 2-level factor (linear-mu route), p = 2
    N = 40
    X = np.column_stack([np.ones(N), (np.arange(N) % 2).astype(float)])
    kat_case("p2", synth(96, N, X, 11), X, ut, gs, pp, di)
    # case B: two factors + one continuous covariate (IRLS-mu route), p = 4
    N = 60
    rng = np.random.default_rng(5)
    X = np.column_stack([
        np.ones(N), (np.arange(N) % 2).astype(float), ((np.arange(N) // 2) % 2).astype(float),
        rng.normal(size=N),
    ])
    kat_case("p4", synth(64, N, X, 12), X, ut, gs, pp, di)
    # case C: three categorical factors, p = 8, 30 cells
    N = 120
    a, b, c = np.arange(N) % 2, (np.arange(N) // 2) % 3, (np.arange(N) // 6) % 5
    cols = [np.ones(N), (a == 1)] + [(b == k) for k in (1, 2)] + [(c == k) for k in (1, 2, 3, 4)]
    X = np.column_stack([np.asarray(v, dtype=float) for v in cols])
    kat_case("p8", synth(48, N, X, 13), X, ut, gs, pp, di)

    # apeGLM MAP LFC known answers: the reference's utils.nbinomGLM per gene
    sh = {}
    for case, sidx in (("p2", 1), ("p4", 3), ("p8", 1)):
        k = np.load(os.path.join(HERE, f"kat_{case}.npz"))
        counts, X, sf = k["counts"], k["X"], k["sf"]
        size = 1.0 / np.clip(k["map_alpha"], 1e-8, max(10, counts.shape[0]))
        G = min(counts.shape[1], 40)
        for tag, ps in (("a", 1.0), ("b", 0.3)):
            r = [ut.nbinomGLM(X, counts[:, i], size[i], np.log(sf), 15, ps, "L-BFGS-B", sidx) for i in range(G)]
            sh[f"{case}{tag}_beta"] = np.stack([x[0] for x in r])
            sh[f"{case}{tag}_invh"] = np.stack([x[1] for x in r])
            sh[f"{case}{tag}_conv"] = np.array([x[2] for x in r], dtype=bool)
            sh[f"{case}{tag}_scale"] = np.array(ps)
        sh[f"{case}_sidx"], sh[f"{case}_size"], sh[f"{case}_G"] = np.array(sidx), size[:G], np.array(G)
    np.savez(os.path.join(HERE, "kat_shrink.npz"), **sh)

    # lowess known answers (summary tail): the reference's utils.lowess on the shapes
    # _independent_filtering feeds it (50 thetas vs rejection counts) plus generic cases
    from scipy.stats import false_discovery_control

    rng = np.random.default_rng(21)
    lw = {}
    for k in range(6):
        n = 50 if k < 4 else 37
        x = np.linspace(rng.uniform(0, 0.2), 0.95, n) if k < 4 else np.sort(rng.uniform(0, 1, n))
        if k < 4:
            y = np.round(np.maximum(800 * np.exp(-((x - 0.4) / 0.5) ** 2) + rng.normal(0, 15, n), 0))
            if k == 3:
                y[:] = 7.0
        else:
            y = np.sin(4 * x) + rng.normal(0, 0.2, n)
        lw[f"x{k}"], lw[f"y{k}"] = x, y
        lw[f"f{k}"] = np.array(1 / 5 if k < 4 else 2 / 3)
        lw[f"out{k}"] = ut.lowess(x, y, frac=float(lw[f"f{k}"]))
    pv = rng.uniform(0, 1, 500) ** 3
    lw["bh_in"], lw["bh_out"] = pv, false_discovery_control(pv, method="bh")
    np.savez(os.path.join(HERE, "kat_lowess.npz"), **lw)

    # R fixtures (data files) copied verbatim
    for sub, files in {
        "single_factor": ["r_test_size_factors.csv", "r_test_dispersions.csv", "r_test_res.csv",
                          "r_test_res_mean_curve.csv", "r_test_res_no_independent_filtering.csv",
                          "r_test_res_greater.csv", "r_test_res_less.csv",
                          "r_test_res_greaterAbs.csv", "r_test_res_lessAbs.csv",
                          "r_test_lfc_shrink_res.csv", "r_test_lfc_shrink_no_apeAdapt_res.csv",
                          "r_vst.csv", "r_vst_with_design.csv", "r_mean_vst.csv",
                          "r_test_size_factors_poscount.csv", "r_iterative_size_factors.csv"],
        "multi_factor": ["r_test_size_factors.csv", "r_test_dispersions.csv", "r_test_res.csv",
                         "r_test_res_outliers.csv", "r_test_lfc_shrink_res.csv"],
        "continuous": ["r_test_size_factors.csv", "r_test_dispersions.csv", "r_test_res.csv",
                       "r_test_res_outliers.csv", "r_test_lfc_shrink_res.csv", "test_counts.csv",
                       "test_metadata.csv"],
        "wide": ["r_test_size_factors.csv", "r_test_dispersions.csv", "r_test_res.csv",
                 "test_counts.csv", "test_metadata.csv"],
        "large_counts": ["r_test_size_factors.csv", "r_test_dispersions.csv", "r_test_res.csv",
                         "r_test_lfc_shrink_res.csv"],
    }.items():
        dst = os.path.join(HERE, f"r_{sub}")
        os.makedirs(dst, exist_ok=True)
        for fn in files:
            shutil.copy(os.path.join(REF, "tests", "data", sub, fn), os.path.join(dst, fn))
    dst = os.path.join(HERE, "synthetic")
    os.makedirs(dst, exist_ok=True)
    for fn in ("test_counts.csv", "test_metadata.csv"):
        shutil.copy(os.path.join(REF, "datasets", "synthetic", fn), os.path.join(dst, fn))


if __name__ == "__main__":
    main()
