"""Host instantiation of the device templates (tests/hostsim: the SAME headers the GPU kernels compile)
against the oracle on inputs far from the benchmark's regime: counts up to 2^26, means of 0.1, strongly over- and
under-dispersed genes, wide and mixed designs, six samples, N not a multiple of anything."""
import numpy as np
import pytest
from scipy.stats import f as f_dist

from oracle import nbglm_oracle as orc
from tests import hostsim as hs


def make(G, N, mean_log2, disp, seed, design="2level"):
    r = np.random.default_rng(seed)
    X = orc.make_design(design, N, r)
    beta = np.zeros((X.shape[1], G))
    beta[0] = mean_log2
    beta[1] = r.normal(0, 1, G)
    sf = np.exp(r.normal(0, 0.3, N))
    mu = sf[:, None] * 2.0 ** (X @ beta)
    size = 1 / disp
    c = r.negative_binomial(size, size / (size + mu)).astype(np.int64)
    return c[:, c.sum(0) > 0], X, sf


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert (np.isfinite(a) == np.isfinite(b)).all()
    m = np.isfinite(b)
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300))) if m.any() else 0.0


@pytest.mark.parametrize("name,mean_log2,disp,design,N", [
    ("huge", 26.0, 0.01, "2level", 60), ("tiny", -3.0, 0.5, "2level", 60), ("wide", 6.0, 0.3, "3factor", 90),
    ("mixed", 7.0, 0.2, "mixed", 48), ("six_samples", 8.0, 0.1, "2level", 6), ("odd_n", 5.0, 0.4, "2level", 131)])
def test_irls_cooks_wald_far_from_the_benchmark_regime(name, mean_log2, disp, design, N):
    c, X, sf = make(30, N, mean_log2, disp, 5, design)
    P = X.shape[1]
    normed = c / sf[:, None]
    mom = orc.mom_dispersions(normed, X, sf, 1e-8, max(10, N))
    assert rel(hs.mom(c, sf, X, 1e-8, max(10, N))["mom"], mom) < 1e-12
    b_o, mu_o, H_o, cv_o = orc.irls(c, sf, X, mom)[:4]
    b_h, mu_h, H_h, cv_h, _, _ = hs.irls(c, sf, X, mom)
    assert (cv_h == cv_o).all()
    assert rel(b_h, b_o) < 1e-9 and rel(mu_h, mu_o) < 1e-9 and rel(H_h, H_o) < 1e-11
    cutoff = f_dist.ppf(0.99, P, N - P)
    ck_h, _, _ = hs.cooks(c, sf, X, mu_o, H_o, cutoff)
    assert rel(ck_h, orc.cooks_distance(c, normed, X, mu_o, H_o)) < 1e-11
    ridge, con = np.diag(np.repeat(1e-6, P)), np.eye(P)[1]
    p_o, s_o, se_o = orc.wald_test(X, mom, b_o, np.exp(X @ b_o.T) * sf[:, None], ridge, con)
    p_h, s_h, se_h = hs.wald(X, mom, b_o, sf, ridge, con, 0.0, None)
    assert rel(se_h, se_o) < 1e-12 and rel(s_h, s_o) < 1e-10 and rel(p_h, p_o) < 1e-9


@pytest.mark.parametrize("mean_log2,disp,tol", [(-3.0, 0.5, 1e-9), (6.0, 5.0, 1e-9), (12.0, 0.01, 1e-5)])
def test_dispersion_fit_across_count_magnitudes(mean_log2, disp, tol):
    """Genes on which both L-BFGS-B runs converge agree within `tol` (the relative stopping rule makes
    the fit of very large counts depend on gradient rounding noise; 2^12 is already beyond the benchmark's means)."""
    N = 60
    c, X, sf = make(40, N, mean_log2, disp, 11)
    mom = orc.mom_dispersions(c / sf[:, None], X, sf, 1e-8, max(10, N))
    mu_hat = orc.lin_reg_mu(c, sf, X, 0.5)
    assert rel(hs.lin_mu(c, sf, X, 0.5), mu_hat) < 1e-12
    a_o, conv_o = orc.alpha_mle(c, X, mu_hat, mom, 1e-8, max(10, N), n_jobs=1)
    a_h, conv_h = hs.alpha_mle(c, X, mu_hat, mom, 1e-8, max(10, N))[:2]
    both = np.asarray(conv_o, bool) & np.asarray(conv_h, bool)
    assert both.mean() > 0.9
    assert np.max(np.abs(a_h - a_o)[both] / a_o[both]) < tol


def test_trend_fit_failure_modes_and_odd_inputs():
    """The trend-fit template agrees with the oracle's restatement of dds.py:1199-1275 on coefficients, on the
    number of outer (re-filtering) iterations and on WHEN the fit is declared failed (flat / increasing / pure-noise
    dispersions -> mean trend), with outliers, genes at min_disp, a handful of genes, NaN means."""
    import warnings

    rng = np.random.default_rng(0)
    G = 400
    nm = np.exp(rng.normal(4, 2, G))
    true = 0.05 + 3.0 / nm
    noisy = lambda s=0.3: true * np.exp(rng.normal(0, s, G))  # noqa: E731
    outl = noisy()
    outl[::17] *= 1e4
    at_min = true.copy()
    at_min[:200] = 1e-8
    cases = {
        "clean": (noisy(), nm), "few": (noisy()[:12], nm[:12]), "outliers": (outl, nm),
        "flat": (np.full(G, 0.2) * np.exp(rng.normal(0, 0.05, G)), nm),
        "increasing": ((0.01 + nm * 1e-3) * np.exp(rng.normal(0, 0.2, G)), nm),
        "noise": (np.exp(rng.normal(-2, 3, G)), nm), "at_min": (at_min * np.exp(rng.normal(0, 0.3, G)), nm),
        "nan_means": (noisy(), np.where(np.arange(G) % 9 == 0, np.nan, nm)),
    }
    failed = set()
    for name, (gw, means) in cases.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            co, n_outer = orc.fit_parametric_trend(np.clip(gw, 1e-8, 100.0), means)
        ch, ok, no = hs.trend_fit(gw, means, 1e-8, 100.0)
        assert ok == (co is not None), name
        assert no == n_outer, name
        if ok:
            assert rel(ch, co) < 1e-9, name
        else:
            failed.add(name)
    assert failed == {"flat", "increasing", "noise"}


@pytest.mark.parametrize("name,mean_log2,disp,design,N,scale", [
    ("huge", 22.0, 0.01, "2level", 40, 1.0), ("tiny", -2.0, 0.5, "2level", 40, 0.3),
    ("wide", 6.0, 0.3, "3factor", 90, 0.5), ("mixed", 7.0, 0.2, "mixed", 48, 1.0),
    ("tight_prior", 6.0, 0.2, "2level", 30, 0.01), ("six_samples", 8.0, 0.1, "2level", 6, 1.0)])
def test_apeglm_templates_far_from_the_benchmark_regime(name, mean_log2, disp, design, N, scale):
    import warnings

    c, X, sf = make(20, N, mean_log2, disp, 5, design)
    G, P = c.shape[1], X.shape[1]
    size, off = np.full(G, 1 / disp), np.log(sf)
    bh, ih, cvh = hs.shrink(c, X, size, off, 15.0, scale, 1)
    for g in range(G):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            bo, io, cvo = orc.nbinom_glm_gene(X, c[:, g], size[g], off, 15.0, scale, 1)
        assert bool(cvh[g]) == bool(cvo)
        assert np.max(np.abs(bh[g] - bo) / np.maximum(np.abs(bo), 1e-6)) < 1e-8
        assert np.max(np.abs(ih[g] - io)) / np.abs(io).max() < 1e-9
