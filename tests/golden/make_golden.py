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


def synth(G, N, X, seed, eff=0.7):
    rng = np.random.default_rng(seed)
    p = X.shape[1]
    beta = np.zeros((p, G))
    beta[0] = rng.normal(4, 2, G)
    for j in range(1, p):
        beta[j] = rng.normal(0, eff, G)
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
    contrast[min(1, p - 1)] = 1.0
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



def hard_cases(ut, gs):
    """Genes on which the reference leaves its main optimiser and takes a grid search (kat_hard.npz).

    * dispersion (fit_alpha_mle, utils.py:546-564): very large counts make scipy's L-BFGS-B line search end in
      the rounding noise of the loss (success = False) -> exp(grid_fit_alpha(...));
    * LFC (irls_solver, utils.py:374-413): IRLS diverges (|beta| > 30: a group without counts, a huge outlier)
      and the bounded L-BFGS-B rescue terminates ABNORMALly on the min_mu kink -> grid_fit_beta(...).
    Every candidate gene of the seeded search is kept (no selection on what any other implementation does)."""
    import warnings

    out = {}
    N = 40
    X = np.column_stack([np.ones(N), (np.arange(N) % 2).astype(float)])
    rng = np.random.default_rng(77)
    sf = np.exp(rng.normal(0, 0.2, N))
    out["X"], out["sf"] = X, sf
    # ---- dispersion
    G = 600
    b0, lfc = rng.uniform(12, 22, G), rng.normal(0, 1, G)
    disp = 10 ** rng.uniform(-2.5, 0.0, G)
    mu = sf[:, None] * 2.0 ** (b0[None, :] + X[:, 1:2] * lfc[None, :])
    Y = rng.negative_binomial(1 / disp[None, :], 1 / (1 + mu * disp[None, :])).astype(np.int64)
    Y = Y[:, Y.max(0) < 2**30]
    normed = Y / sf[:, None]
    mom = np.clip(np.minimum(ut.fit_rough_dispersions(normed, X), ut.fit_moments_dispersions(normed, sf)), 1e-8, 40.0)
    mu_hat = np.stack([ut.fit_lin_mu(Y[:, g], sf, X, 0.5) for g in range(Y.shape[1])], axis=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = [ut.fit_alpha_mle(Y[:, g], X, mu_hat[:, g], mom[g], 1e-8, 40.0) for g in range(Y.shape[1])]
    conv = np.array([x[1] for x in r], dtype=bool)
    sel = np.r_[np.nonzero(~conv)[0][:24], np.nonzero(conv)[0][:8]]
    out["a_counts"], out["a_mu_hat"], out["a_mom"] = Y[:, sel], mu_hat[:, sel], mom[sel]
    out["a_alpha"] = np.array([r[g][0] for g in sel])
    out["a_conv"] = conv[sel]
    out["a_grid_log_alpha"] = np.array(
        [gs.grid_fit_alpha(Y[:, g], X, mu_hat[:, g], mom[g], 1e-8, 40.0) for g in sel])
    # ---- LFC
    found = []
    state = {}
    orig = ut.minimize

    def spy(*a, **k):
        state["res"] = orig(*a, **k)
        return state["res"]

    ut.minimize = spy
    try:
        for trial in range(1200):
            kind = trial % 4
            base, lf = 2.0 ** rng.uniform(-3, 6), rng.normal(0, 3)
            m = sf * base * np.exp(X[:, 1] * lf)
            dtrue = 10 ** rng.uniform(-3, 1.3)
            y = rng.negative_binomial(1 / dtrue, 1 / (1 + m * dtrue)).astype(np.int64)
            if kind == 1:
                y[X[:, 1] == 1] = 0
            if kind == 2:
                y[X[:, 1] == 0] = 0
            if kind == 3:
                y[rng.integers(0, N)] = int(10 ** rng.uniform(3, 7))
            if y.sum() == 0:
                continue
            d = 10 ** rng.uniform(-8, 1.6)
            state.pop("res", None)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                b, mu_, H, cv = ut.irls_solver(y, sf, X, d, 0.5, 1e-8)
            if "res" in state and not state["res"].success:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    gb = gs.grid_fit_beta(y, sf, X, d)
                found.append((y, d, b, mu_, H, cv, state["res"].x.copy(), gb))
            if len(found) >= 32:
                break
    finally:
        ut.minimize = orig
    out["b_counts"] = np.stack([f[0] for f in found], axis=1)
    out["b_disp"] = np.array([f[1] for f in found])
    out["b_beta"] = np.stack([f[2] for f in found])
    out["b_mu"] = np.stack([f[3] for f in found], axis=1)
    out["b_H"] = np.stack([f[4] for f in found], axis=1)
    out["b_conv"] = np.array([f[5] for f in found], dtype=bool)
    out["b_rescue_x"] = np.stack([f[6] for f in found])
    out["b_grid_beta"] = np.stack([f[7] for f in found])
    np.savez_compressed(os.path.join(HERE, "kat_hard.npz"), **out)
    print("hard: dispersion genes", len(sel), "non-converged", int((~out["a_conv"]).sum()),
          "| LFC genes with a failed rescue", len(found))


def bfgs_cases(ut):
    """optimizer="BFGS" of the reference's per-gene kernels (utils.py:343, 389-399, 546-554): kat_bfgs.npz.

    * dispersion: fit_alpha_mle(..., optimizer="BFGS") on the inputs of kat_p2 / kat_p8 (genewise and MAP fits) and on
      the huge-count genes of kat_hard (unbounded search: several end in "precision loss" -> grid value);
    * LFC: irls_solver(..., optimizer="BFGS") on genes whose IRLS diverges (a group without counts, a huge outlier,
      low counts on the 30-cell design): the unbounded BFGS rescue, grid_fit_beta when it fails at p = 2."""
    import warnings

    out = {}
    for case in ("p2", "p8"):
        k = np.load(os.path.join(HERE, f"kat_{case}.npz"))
        Y, X, mu, mom, fit = k["counts"], k["X"], k["mu_hat"], k["mom"], k["fitted"]
        N = Y.shape[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r1 = [ut.fit_alpha_mle(Y[:, g], X, mu[:, g], mom[g], 1e-8, max(10, N), optimizer="BFGS")
                  for g in range(Y.shape[1])]
            r2 = [ut.fit_alpha_mle(Y[:, g], X, mu[:, g], fit[g], 1e-8, max(10, N), float(k["prior_var"]), True, True,
                                   optimizer="BFGS") for g in range(Y.shape[1])]
        out[f"{case}_gw_alpha"] = np.array([r[0] for r in r1])
        out[f"{case}_gw_conv"] = np.array([r[1] for r in r1], dtype=bool)
        out[f"{case}_map_alpha"] = np.array([r[0] for r in r2])
        out[f"{case}_map_conv"] = np.array([r[1] for r in r2], dtype=bool)
    h = np.load(os.path.join(HERE, "kat_hard.npz"))
    Y, X, mu, mom = h["a_counts"], h["X"], h["a_mu_hat"], h["a_mom"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = [ut.fit_alpha_mle(Y[:, g], X, mu[:, g], mom[g], 1e-8, 40.0, optimizer="BFGS") for g in range(Y.shape[1])]
    out["hard_alpha"] = np.array([x[0] for x in r])
    out["hard_conv"] = np.array([x[1] for x in r], dtype=bool)
    # ---- LFC rescue with BFGS: p = 2 (the failed-rescue genes of kat_hard + fresh divergent genes) and p = 8
    sf = h["sf"]
    Yb, db = h["b_counts"], h["b_disp"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rb = [ut.irls_solver(Yb[:, g], sf, X, db[g], 0.5, 1e-8, optimizer="BFGS") for g in range(Yb.shape[1])]
    out["b2_beta"] = np.stack([x[0] for x in rb])
    out["b2_mu"] = np.stack([x[1] for x in rb], axis=1)
    out["b2_H"] = np.stack([x[2] for x in rb], axis=1)
    out["b2_conv"] = np.array([x[3] for x in rb], dtype=bool)
    k8 = np.load(os.path.join(HERE, "kat_p8.npz"))
    X8 = k8["X"]
    N8 = X8.shape[0]
    rng = np.random.default_rng(91)
    sf8 = np.exp(rng.normal(0, 0.2, N8))
    found = []
    state = {}
    orig = ut.minimize

    def spy(*a, **kw):
        state["res"] = orig(*a, **kw)
        return state["res"]

    ut.minimize = spy
    try:
        for trial in range(4000):
            base = 2.0 ** rng.uniform(-4, 1)
            y = rng.negative_binomial(2.0, 2.0 / (2.0 + sf8 * base)).astype(np.int64)
            if trial % 3 == 0:
                y[rng.integers(0, N8)] = int(10 ** rng.uniform(2, 5))
            if y.sum() == 0:
                continue
            d = 10 ** rng.uniform(-3, 1)
            state.pop("res", None)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                b, mu_, H, cv = ut.irls_solver(y, sf8, X8, d, 0.5, 1e-8, optimizer="BFGS")
            if "res" in state:
                found.append((y, d, b, mu_, H, cv))
            if len(found) >= 24:
                break
    finally:
        ut.minimize = orig
    out["b8_X"], out["b8_sf"] = X8, sf8
    out["b8_counts"] = np.stack([f[0] for f in found], axis=1)
    out["b8_disp"] = np.array([f[1] for f in found])
    out["b8_beta"] = np.stack([f[2] for f in found])
    out["b8_mu"] = np.stack([f[3] for f in found], axis=1)
    out["b8_H"] = np.stack([f[4] for f in found], axis=1)
    out["b8_conv"] = np.array([f[5] for f in found], dtype=bool)
    np.savez_compressed(os.path.join(HERE, "kat_bfgs.npz"), **out)
    print("bfgs: dispersion non-converged p2/p8/hard", int((~out["p2_gw_conv"]).sum()), int((~out["p8_gw_conv"]).sum()),
          int((~out["hard_conv"]).sum()), "| p=2 rescues converged", int(out["b2_conv"].sum()), "of", len(rb),
          "| p=8 rescues", len(found), "converged", int(out["b8_conv"].sum()))


# end-to-end cases at the benchmark SHAPES (BASELINE configs[2..4]: N = 1000 / 500 / 5000): gene count, samples, design
# kind and seed of pydeseq2_amd.synth.synth_counts - the counts regenerate from these, only outputs are committed
E2E_CASES = {"c3": (8000, 1000, "2level", 2), "c4": (4000, 500, "3factor", 3), "c5": (4000, 5000, "mixed", 4)}
E2E_FIELDS = ("size_factors", "normed_means", "non_zero", "mom_dispersions", "genewise_dispersions",
              "genewise_converged", "trend_coeffs", "fitted_dispersions", "squared_logres", "prior_disp_var",
              "MAP_dispersions", "MAP_converged", "outlier_genes", "dispersions", "LFC", "LFC_converged", "replaced",
              "refitted", "new_all_zeroes", "cooks_outlier", "pvalue", "stat", "lfcSE")


def e2e_cases(di, which=None):
    """kat_e2e_{c3,c4,c5}.npz: the UNMODIFIED reference kernels end to end at the benchmark shapes.

    ``DefaultInference`` (default_inference.py:14-264: utils.fit_alpha_mle / irls_solver / wald_test / fit_lin_mu /
    fit_rough_dispersions / fit_moments_dispersions / dispersion_trend_gamma_glm per gene through joblib) is driven in
    dds.py / ds.py call order by the oracle's orchestrator (pydeseq2.dds itself needs anndata).  What the orchestrator
    adds around the reference's kernels - size factors, trend iteration, prior, Cook's, refit bookkeeping - is the part
    of the oracle that the R fixtures pin; every per-gene number in these files was produced by the reference's own
    code.  The counts are not stored: tests regenerate them with pydeseq2_amd.synth.synth_counts(G, N, design, seed)."""
    import time
    import warnings

    root = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, root)
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.synth import synth_counts

    inf = di.DefaultInference(n_cpus=os.cpu_count())
    for name, (G, N, design, seed) in E2E_CASES.items():
        if which and name not in which:
            continue
        counts, X = synth_counts(G, N, design, seed)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = orc.deseq2(counts, X, inference=inf, keep_layers=False)
        dt = time.perf_counter() - t0
        out = {"G": np.array(G), "N": np.array(N), "seed": np.array(seed), "design": np.array(design),
               "counts_sha256": np.array(__import__("hashlib").sha256(np.ascontiguousarray(counts)).hexdigest()),
               "X": X}
        for f in E2E_FIELDS:
            out[f] = np.asarray(getattr(r, f))
        np.savez_compressed(os.path.join(HERE, f"kat_e2e_{name}.npz"), **out)
        print(f"e2e {name}: {G} genes x {N} samples p={X.shape[1]}: {dt:.1f} s, genewise non-converged "
              f"{int(np.nansum(r.genewise_converged == 0))}, MAP non-converged {int(np.nansum(r.MAP_converged == 0))}, "
              f"refitted {int(r.refitted.sum())}, cooks_outlier {int(r.cooks_outlier.sum())}", flush=True)


def main():
    ut, gs, pp, di = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "e2e":  # round 3: end-to-end outputs at the benchmark shapes
        e2e_cases(di, sys.argv[2:])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "narrow":  # round 3: design widths 3, 5, 6, 7
        narrow_cases(ut, gs, pp, di)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "widths":  # round 4: design widths 1, 9, 11 and the mixed p = 8 design
        width_cases(ut, gs, pp, di)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "wider":  # round 6: design widths 40 and 48 (the engine's limit moved 32 -> 48)
        wider_cases(ut, gs, pp, di)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round2":  # only the files added in round 2
        wide_cases(ut, gs, pp, di)
        hard_cases(ut, gs)
        bfgs_cases(ut)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "bfgs":
        bfgs_cases(ut)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "shrink_opt":  # round 4: apeGLM shrinkage with the two other optimisers
        shrink_optimizer_cases(ut)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "shrink_mid":  # round 4: apeGLM shrinkage at the widths 5-7 and 9-12
        shrink_mid_cases(ut)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "shrink_wider":  # round 6: apeGLM shrinkage for designs of 33 ... 48 columns
        shrink_wide_cases(ut, (("p40", 2), ("p48", 5)), "kat_shrink_wider.npz", 16)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "shrink_wide":  # round 3: apeGLM shrinkage for designs of 13 ... 32 columns
        shrink_wide_cases(ut)
        return
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

    wide_cases(ut, gs, pp, di)
    narrow_cases(ut, gs, pp, di)
    width_cases(ut, gs, pp, di)
    hard_cases(ut, gs)
    bfgs_cases(ut)
    rest_of_main(ut)
    e2e_cases(di)


def wide_cases(ut, gs, pp, di):
    # cases D..G: wide designs (two categorical factors + continuous covariates): p = 10, 12 exercise the
    # split second sweep of the register path, p = 16 and 24 the LDS path for designs beyond 12 columns
    for pw, N, seed in ((10, 80, 14), (12, 96, 15), (16, 128, 16), (24, 168, 17)):
        rng = np.random.default_rng(100 + pw)
        a, b = np.arange(N) % 2, (np.arange(N) // 2) % 4
        cols = [np.ones(N), (a == 1)] + [(b == k) for k in (1, 2, 3)]
        while len(cols) < pw:
            cols.append(rng.normal(0, 0.6, N))
        X = np.column_stack([np.asarray(v, dtype=float) for v in cols])
        kat_case(f"p{pw}", synth(32, N, X, seed, eff=0.35), X, ut, gs, pp, di)


def wider_cases(ut, gs, pp, di):
    """cases P, Q (round 6): p = 40 and p = 48 - three 16-row tiles of the matrix-core Gram accumulation; the same design family
    as the other wide cases (a 2-level and a 4-level factor + continuous covariates)."""
    for pw, N, seed in ((40, 260, 41), (48, 300, 42)):
        rng = np.random.default_rng(100 + pw)
        a, b = np.arange(N) % 2, (np.arange(N) // 2) % 4
        cols = [np.ones(N), (a == 1)] + [(b == k) for k in (1, 2, 3)]
        while len(cols) < pw:
            cols.append(rng.normal(0, 0.6, N))
        X = np.column_stack([np.asarray(v, dtype=float) for v in cols])
        # expressed genes only: with 40+ coefficients on ~6 samples each, a gene of mean count 3 sends the reference's IRLS
        # to its 250-iteration limit and through the L-BFGS-B rescue, whose stopping point on that flat likelihood is the
        # chaos kat_hard.npz pins separately - here the widths themselves are what is tested
        c = synth(80, N, X, seed, eff=0.3)
        c = c[:, c.mean(0) >= 40][:, :24]
        assert c.shape[1] == 24
        kat_case(f"p{pw}", c, X, ut, gs, pp, di)


def narrow_cases(ut, gs, pp, di):
    """cases H..K (round 3): design widths 3, 5, 6, 7 - the widths around the kernels' routing thresholds (per-cell sums
    from 5 cells / P >= 3, sixteen-lane IRLS from P = 5, inlined cell evaluation from P = 7, split second sweep from 9):
    p = 3: one 3-level factor (3 cells = p: linear-model mu_hat); p = 5: 2 x 4 levels (8 cells, cell path);
    p = 6: 2 x 3 levels + a continuous covariate (no cells: register path); p = 7: 3 x 5 levels (15 cells)."""
    specs = {3: (48, 18), 5: (64, 19), 6: (72, 20), 7: (90, 21)}
    for pw, (N, seed) in specs.items():
        rng = np.random.default_rng(300 + pw)
        i = np.arange(N)
        if pw == 3:
            cols = [np.ones(N), (i % 3 == 1), (i % 3 == 2)]
        elif pw == 5:
            a, b = i % 2, (i // 2) % 4
            cols = [np.ones(N), a == 1] + [(b == k) for k in (1, 2, 3)]
        elif pw == 6:
            a, b = i % 2, (i // 2) % 3
            cols = [np.ones(N), a == 1] + [(b == k) for k in (1, 2)] + [rng.normal(0, 0.6, N), rng.normal(0, 0.6, N)]
        else:
            a, b = i % 3, (i // 3) % 5
            cols = [np.ones(N)] + [(a == k) for k in (1, 2)] + [(b == k) for k in (1, 2, 3, 4)]
        X = np.column_stack([np.asarray(v, dtype=float) for v in cols])
        assert X.shape[1] == pw
        kat_case(f"p{pw}", synth(40, N, X, seed, eff=0.5), X, ut, gs, pp, di)


def width_cases(ut, gs, pp, di):
    """cases L..O (round 4): the design widths that had no reference KAT - p = 1 (intercept only: what vst() fits), p = 9 and
    p = 11 (register path with the split second sweep) - and "p8m", the benchmark's MIXED design shape (BASELINE
    configs[4]: a 2-level and a 4-level factor + three continuous covariates, p = 8) for the kernels that split
    X^T W X into a per-cell block and a small continuous block."""
    specs = {"p1": (1, 24, 31), "p9": (9, 100, 32), "p11": (11, 110, 33), "p8m": (8, 160, 34)}
    for name, (pw, N, seed) in specs.items():
        rng = np.random.default_rng(400 + pw + (50 if name == "p8m" else 0))
        i = np.arange(N)
        if pw == 1:
            cols = [np.ones(N)]
        else:
            a, b = i % 2, (i // 2) % 4
            cols = [np.ones(N), a == 1] + [(b == k) for k in (1, 2, 3)]
            while len(cols) < pw:
                cols.append(rng.normal(0, 0.6 if name != "p8m" else 1.0, N))
        X = np.column_stack([np.asarray(v, dtype=float) for v in cols])
        assert X.shape[1] == pw
        kat_case(name, synth(40, N, X, seed, eff=0.4), X, ut, gs, pp, di)


def shrink_wide_cases(ut, cases=(("p16", 1), ("p24", 3)), out="kat_shrink_wide.npz", g_max=24):
    """utils.nbinomGLM (the unmodified reference, L-BFGS-B as ds.py:407 calls it) on the wide-design KAT inputs:
    p = 16 (one factor with 16 levels) and p = 24 -> kat_shrink_wide.npz; round 6: p = 40, 48 -> kat_shrink_wider.npz"""
    sh = {}
    for case, sidx in cases:
        k = np.load(os.path.join(HERE, f"kat_{case}.npz"))
        counts, X, sf = k["counts"], k["X"], k["sf"]
        size = 1.0 / np.clip(k["map_alpha"], 1e-8, max(10, counts.shape[0]))
        G = min(counts.shape[1], g_max)
        for tag, ps in (("a", 1.0), ("b", 0.3)):
            r = [ut.nbinomGLM(X, counts[:, i], size[i], np.log(sf), 15, ps, "L-BFGS-B", sidx) for i in range(G)]
            sh[f"{case}{tag}_beta"] = np.stack([x[0] for x in r])
            sh[f"{case}{tag}_invh"] = np.stack([x[1] for x in r])
            sh[f"{case}{tag}_conv"] = np.array([x[2] for x in r], dtype=bool)
            sh[f"{case}{tag}_scale"] = np.array(ps)
        sh[f"{case}_sidx"], sh[f"{case}_size"], sh[f"{case}_G"] = np.array(sidx), size[:G], np.array(G)
    np.savez(os.path.join(HERE, out), **sh)


def shrink_mid_cases(ut):
    """utils.nbinomGLM (the unmodified reference) on the KAT inputs of the design widths between the three of kat_shrink.npz:
    p = 5, 6, 7 (the wavefront-resident optimiser, dsq_lbfgsb_wave.h, with padding rows) and p = 9 ... 12 -> kat_shrink_mid.npz"""
    sh = {}
    for case, sidx in (("p5", 1), ("p6", 4), ("p7", 2), ("p9", 1), ("p10", 6), ("p11", 3), ("p12", 1)):
        k = np.load(os.path.join(HERE, f"kat_{case}.npz"))
        counts, X, sf = k["counts"], k["X"], k["sf"]
        size = 1.0 / np.clip(k["map_alpha"], 1e-8, max(10, counts.shape[0]))
        G = min(counts.shape[1], 24)
        for tag, ps in (("a", 1.0), ("b", 0.3)):
            r = [ut.nbinomGLM(X, counts[:, i], size[i], np.log(sf), 15, ps, "L-BFGS-B", sidx) for i in range(G)]
            sh[f"{case}{tag}_beta"] = np.stack([x[0] for x in r])
            sh[f"{case}{tag}_invh"] = np.stack([x[1] for x in r])
            sh[f"{case}{tag}_conv"] = np.array([x[2] for x in r], dtype=bool)
            sh[f"{case}{tag}_scale"] = np.array(ps)
        sh[f"{case}_sidx"], sh[f"{case}_size"], sh[f"{case}_G"] = np.array(sidx), size[:G], np.array(G)
    np.savez(os.path.join(HERE, "kat_shrink_mid.npz"), **sh)


def shrink_optimizer_cases(ut):
    """utils.nbinomGLM (the unmodified reference) with optimizer = "BFGS" and "Newton-CG" (utils.py:1028-1030, 1112-1121)
    on the KAT inputs of p = 2, 4, 8, 12 -> kat_shrink_opt.npz"""
    import warnings

    sh = {}
    for case, sidx in (("p2", 1), ("p4", 3), ("p8", 1), ("p12", 1)):
        k = np.load(os.path.join(HERE, f"kat_{case}.npz"))
        counts, X, sf = k["counts"], k["X"], k["sf"]
        size = 1.0 / np.clip(k["map_alpha"], 1e-8, max(10, counts.shape[0]))
        G = min(counts.shape[1], 24)
        for opt, tag in (("BFGS", "bfgs"), ("Newton-CG", "ncg")):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r = [ut.nbinomGLM(X, counts[:, i], size[i], np.log(sf), 15, 0.5, opt, sidx) for i in range(G)]
            sh[f"{case}_{tag}_beta"] = np.stack([x[0] for x in r])
            sh[f"{case}_{tag}_invh"] = np.stack([x[1] for x in r])
            sh[f"{case}_{tag}_conv"] = np.array([x[2] for x in r], dtype=bool)
        sh[f"{case}_sidx"], sh[f"{case}_size"], sh[f"{case}_G"], sh[f"{case}_scale"] = (np.array(sidx), size[:G], np.array(G),
                                                                                        np.array(0.5))
    np.savez(os.path.join(HERE, "kat_shrink_opt.npz"), **sh)


def rest_of_main(ut):

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
