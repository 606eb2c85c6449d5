"""Shared test helpers (fixture loading, treatment-coded designs, comparisons)."""
import os

import numpy as np
import pandas as pd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kat(name):
    return dict(np.load(os.path.join(GOLD, f"kat_{name}.npz"), allow_pickle=False))


def load_e2e(name):
    """kat_e2e_<name>.npz (tests/golden/make_golden.py e2e): the UNMODIFIED reference end to end at a benchmark shape.
    Returns (counts, X, ref) with the counts regenerated from the stored seed (checked against the stored digest) and
    ``ref`` an object carrying the reference's outputs under the oracle's / engine's result field names."""
    import hashlib
    from types import SimpleNamespace

    from pydeseq2_amd.synth import synth_counts

    k = load_kat(f"e2e_{name}")
    counts, X = synth_counts(int(k["G"]), int(k["N"]), str(k["design"]), int(k["seed"]))
    assert hashlib.sha256(np.ascontiguousarray(counts)).hexdigest() == str(k["counts_sha256"]), "generator drifted"
    assert np.array_equal(X, k["X"])
    ref = SimpleNamespace(**{f: (v if v.ndim else v.item()) for f, v in k.items()
                             if f not in ("G", "N", "seed", "design", "counts_sha256", "X")})
    return counts, X, ref


def flag_flips(res, ref):
    """Genes on which two runs disagree about an L-BFGS-B success flag (genewise, MAP) or about the refit."""
    with np.errstate(invalid="ignore"):
        gw = (res.genewise_converged != ref.genewise_converged) & ref.non_zero
        mp = (res.MAP_converged != ref.MAP_converged) & ref.non_zero
    return gw, mp, np.asarray(res.refitted) != np.asarray(ref.refitted)


def load_dataset(which):
    """Return (counts DataFrame samples x genes, metadata DataFrame)."""
    d = {"synthetic": "synthetic", "continuous": "r_continuous", "wide": "r_wide"}[which]
    counts = pd.read_csv(os.path.join(GOLD, d, "test_counts.csv"), index_col=0).T
    meta = pd.read_csv(os.path.join(GOLD, d, "test_metadata.csv"), index_col=0)
    return counts, meta


def r_csv(sub, name):
    return pd.read_csv(os.path.join(GOLD, f"r_{sub}", name), index_col=0)


def treatment_design(meta, factors, continuous=()):
    """Intercept + treatment-coded (first sorted level = reference) factor columns.

    Same column order formulaic gives for "~f1 + f2 + x": Intercept, f1[T.*], f2[T.*], x.
    Returns (X ndarray, column names).
    """
    cols, names = [np.ones(len(meta))], ["Intercept"]
    for f in factors:
        levels = sorted(meta[f].unique())
        for lv in levels[1:]:
            cols.append((meta[f] == lv).to_numpy().astype(float))
            names.append(f"{f}[T.{lv}]")
    for c in continuous:
        cols.append(meta[c].to_numpy().astype(float))
        names.append(c)
    return np.column_stack(cols), names


def max_rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    ok = ~np.isnan(b)
    assert (np.isnan(a) == np.isnan(b)).all(), "NaN pattern differs"
    return float(np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))) if ok.any() else 0.0


def assert_close(a, b, rtol, atol=0.0, what=""):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert (np.isnan(a) == np.isnan(b)).all(), f"{what}: NaN pattern differs"
    inf = np.isinf(b)
    assert (a[inf] == b[inf]).all() and (np.isinf(a) == inf).all(), f"{what}: inf pattern differs"
    ok = ~np.isnan(b) & ~inf
    err = np.abs(a[ok] - b[ok]) - (atol + rtol * np.abs(b[ok]))
    assert (err <= 0).all(), (
        f"{what}: max excess {err.max():.3e}; worst rel "
        f"{np.max(np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1e-300)):.3e}"
    )


def cr_loss(counts, X, mu, alpha):
    """Cox-Reid adjusted NB negative log-likelihood of one gene (utils.py:509-515), numpy/scipy arithmetic."""
    from scipy.special import gammaln

    a = 1.0 / alpha
    n = len(counts)
    logbinom = gammaln(counts + a) - gammaln(counts + 1) - gammaln(a)
    nll = n * a * np.log(alpha) + (-logbinom + (counts + a) * np.log(mu + a) - counts * np.log(mu)).sum()
    W = mu / (1 + mu * alpha)
    return nll + 0.5 * np.linalg.slogdet((X.T * W) @ X)[1]


def check_hard_dispersion_genes(k, a, c, grid_log_alpha):
    """Assertions shared by the host-instantiation and the GPU test on kat_hard.npz's dispersion genes
    (huge counts: scipy's L-BFGS-B reports success = False inside the rounding noise of the reference's
    loss and the reference returns the quantised grid value, utils.py:556-564)."""
    X = k["X"]
    # the grid search itself is deterministic: 1e-12 on every gene
    assert np.abs(grid_log_alpha - k["a_grid_log_alpha"]).max() < 1e-12
    ref_nc = ~k["a_conv"]
    both_nc = ref_nc & ~c
    assert_close(a[both_nc], k["a_alpha"][both_nc], 1e-12, 0, "alpha after the grid fallback on both sides")
    both_c = k["a_conv"] & c
    assert both_c.sum() >= 5
    assert_close(a[both_c], k["a_alpha"][both_c], 2e-5, 0, "alpha, both converged")
    # the engine folds the n/alpha log(alpha) term into the per-sample sum (DESIGN.md par. 7) and converges
    # where the reference's noisier loss does not: its value then lies within the grid's resolution of the
    # reference's and is the better optimum of the reference's own objective
    only_ref = ref_nc & c
    step = 2 * (np.log(40.0) - np.log(1e-8)) / 99 / 99
    assert (np.abs(np.log(a[only_ref]) - np.log(k["a_alpha"][only_ref])) <= step * (1 + 1e-9)).all()
    for g in np.nonzero(only_ref)[0]:
        y, m = k["a_counts"][:, g].astype(float), k["a_mu_hat"][:, g]
        assert cr_loss(y, X, m, a[g]) <= cr_loss(y, X, m, k["a_alpha"][g]) + 1e-9 * abs(cr_loss(y, X, m, a[g]))
    return int(both_nc.sum()), int(only_ref.sum())


def check_hard_lfc_genes(k, b, mu, H, conv, min_fallback=0.6):
    """kat_hard.npz's LFC genes: IRLS diverges, the bounded L-BFGS-B rescue terminates ABNORMALly in the
    reference -> grid_fit_beta (utils.py:374-413).  Where the engine's rescue fails too the result is the
    reference's grid value; where it reports success it stopped at the iterate scipy stopped at (the success
    flag of these line searches is decided at rounding-noise level)."""
    nc = ~conv
    assert not k["b_conv"].any()
    assert nc.mean() >= min_fallback, nc.mean()
    assert_close(b[nc], k["b_beta"][nc], 1e-10, 1e-12, "beta after the grid fallback")
    assert_close(b[nc], k["b_grid_beta"][nc], 1e-12, 1e-12, "= grid_fit_beta")
    assert_close(mu[:, nc], k["b_mu"][:, nc], 1e-9, 1e-12, "mu")
    assert_close(H[:, nc], k["b_H"][:, nc], 1e-8, 1e-12, "H")
    assert_close(b[~nc], k["b_rescue_x"][~nc], 1e-3, 1e-6, "rescue iterate")
    return float(nc.mean())


def check_bfgs_kats(load_kat, alpha_mle, irls, exact=False):
    """optimizer="BFGS" (utils.py:343, 389-399, 546-554) against kat_bfgs.npz (the unmodified reference).

    ``alpha_mle(counts, X, mu, alpha_hat, min_disp, max_disp, prior_var, cr_reg, prior_reg) -> (alpha, conv)`` and
    ``irls(counts, sf, X, disp) -> (beta, conv)`` with the BFGS optimiser selected.  exact: the oracle (scipy itself).
    For the device templates the usual rule applies: genes on which both sides agree about scipy's ``success`` must
    agree to 1e-6 (the stopping point under gtol = 1e-5 moves with the last bits of the loss); the flag itself is a
    coin flip on the genes whose line search ends in the rounding noise of the loss ("precision loss" at
    |g| ~ 1.2e-5, the huge-count genes of kat_hard), so only the number of such genes is bounded."""
    k = load_kat("bfgs")
    tol = 1e-12 if exact else 1e-6
    for case in ("p2", "p8"):
        kk = load_kat(case)
        N = kk["counts"].shape[0]
        a, c = alpha_mle(kk["counts"], kk["X"], kk["mu_hat"], kk["mom"], 1e-8, max(10, N), None, True, False)
        assert (c == k[f"{case}_gw_conv"]).all()
        assert_close(a, k[f"{case}_gw_alpha"], tol, 0, f"{case} genewise alpha (BFGS)")
        a, c = alpha_mle(kk["counts"], kk["X"], kk["mu_hat"], kk["fitted"], 1e-8, max(10, N),
                         float(kk["prior_var"]), True, True)
        same = c == k[f"{case}_map_conv"]
        assert (~same).sum() <= (0 if exact else 3)
        assert_close(a[same], k[f"{case}_map_alpha"][same], tol, 0, f"{case} MAP alpha (BFGS)")
    h = load_kat("hard")
    a, c = alpha_mle(h["a_counts"], h["X"], h["a_mu_hat"], h["a_mom"], 1e-8, 40.0, None, True, False)
    same = c == k["hard_conv"]
    assert same.mean() >= (1.0 if exact else 0.5)
    assert_close(a[same & c], k["hard_alpha"][same & c], 1e-12 if exact else 1e-5, 0, "huge-count genes, both converged")
    assert_close(a[same & ~c], k["hard_alpha"][same & ~c], 1e-12, 0, "huge-count genes, both on the grid")
    b, cv = irls(h["b_counts"], h["sf"], h["X"], h["b_disp"])
    same = cv == k["b2_conv"]
    assert (~same).sum() <= (0 if exact else 2)
    assert_close(b[same], k["b2_beta"][same], 1e-12 if exact else 1e-6, 1e-9, "p = 2 rescue (BFGS / grid)")
    b, cv = irls(k["b8_counts"], k["b8_sf"], k["b8_X"], k["b8_disp"])
    assert (cv == k["b8_conv"]).all()
    assert_close(b, k["b8_beta"], 1e-12 if exact else 1e-6, 1e-7, "p = 8 rescue (BFGS)")
