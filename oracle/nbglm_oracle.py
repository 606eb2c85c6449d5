"""CPU oracle for the PyDESeq2 ``deseq2()`` -> Wald hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy *restatement* of the reference algorithm
(owkin/PyDESeq2 v0.5.3).  It exists to check the HIP engine in
``pydeseq2_amd`` and to provide the ``cpu_baseline`` leg of ``bench.py``.
Nothing under ``pydeseq2_amd/`` may import it: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

Parity pinning
--------------
* every per-gene routine here is checked against the *unmodified* reference
  functions (imported from ``/root/reference`` through an on-disk shim) on
  seeded inputs by ``tests/golden/make_golden.py``; the resulting vectors are
  committed as ``tests/golden/*.npz`` and re-checked by
  ``tests/test_oracle_golden.py`` on every run;
* the end-to-end orchestration is checked against the reference's own
  R-DESeq2 fixtures (``tests/data/{single_factor,multi_factor,continuous,wide}``,
  copied to ``tests/golden/r_*``) at the reference's own tolerances;
* end to end at the benchmark shapes (8000 x 1000 p=2, 4000 x 500 p=8, 1000 x 5000 p=8 with continuous
  covariates) against the outputs of the unmodified ``DefaultInference`` (``tests/golden/kat_e2e_*.npz``): the
  per-gene routines below repeat the reference's operation sequence call for call (per-gene ``scipy.linalg.lstsq`` as
  sklearn's ``LinearRegression`` does, per-gene IRLS with ``scipy.linalg.solve(assume_a="pos")``, the gamma-GLM
  regressors built through pandas, ``(X.T * W) @ X`` products), so every output is BIT-IDENTICAL to the reference's on
  the machine that generated the files (``test_oracle_end_to_end_is_the_unmodified_reference_at_benchmark_shapes``).

Third-party arithmetic the reference delegates to (and so does this oracle):
scipy 1.15.3 ``optimize.minimize(method="L-BFGS-B")``, ``special.gammaln``,
``special.polygamma``, ``stats.norm/f``; numpy 2.2.6 ``linalg``.

All ``file:line`` citations are relative to the reference tree
(``pydeseq2/...``).  Conventions follow the reference: ``counts`` is
samples x genes, natural-log LFCs, dispersions ``alpha`` with
``var = mu + alpha mu^2``.
"""

from __future__ import annotations

import math
import os
import warnings
from dataclasses import dataclass, field

import numpy as np
from scipy.linalg import solve as sp_solve
from scipy.optimize import minimize
from scipy.special import gammaln, polygamma
from scipy.stats import f as f_dist
from scipy.stats import norm, trim_mean

# --------------------------------------------------------------------------
# a1/a2  size factors: median of ratios          preprocessing.py:31-102
# --------------------------------------------------------------------------


def logmeans_and_filter(counts: np.ndarray):
    """Gene-wise mean of log counts and the finite-mean mask (preprocessing.py:31-56)."""
    with np.errstate(divide="ignore"):
        lc = np.log(counts)
    lm = lc.mean(0)
    return lm, ~np.isinf(lm)


def size_factors_ratio(counts: np.ndarray):
    """Median-of-ratios size factors (preprocessing.py:59-102, dds.py:692-708).

    Returns (size_factors[N], normed_counts[N,G], logmeans[G], filtered[G]).
    """
    lm, keep = logmeans_and_filter(counts)
    with np.errstate(divide="ignore"):
        lc = np.log(counts)
    ratios = lc[:, keep] - lm[keep]
    sf = np.exp(np.median(ratios, axis=1))
    return sf, counts / sf[:, None], lm, keep


def nb_nll_genes(counts, mu, alpha):
    """Per-gene NLL, array-alpha branch of utils.nb_nll (utils.py:216-226): counts, mu N x G, alpha [G]."""
    a1 = 1 / alpha
    logbinom = gammaln(counts + a1) - gammaln(counts + 1) - gammaln(a1)
    return (a1 * np.log(alpha) - logbinom + (counts + a1) * np.log(mu + a1) - counts * np.log(mu)).sum(0)


def size_factors_iterative(counts, niter=10, quant=0.95, min_mu=0.5, min_disp=1e-8, max_disp=10.0, beta_tol=1e-8,
                           n_jobs=1):
    """``DeseqDataSet._fit_iterate_size_factors`` (dds.py:1460-1548): alternate dispersion fits with an
    intercept-only design (mean trend) and a Powell search over the log size factors on the NLL summed
    over the genes below its 0.95 quantile.  As in the reference, the method-of-moments start values are
    computed on the RAW counts throughout (``layers["normed_counts"]`` is only refreshed at the end)."""
    counts = np.asarray(counts)
    N, G = counts.shape
    max_disp = max(max_disp, N)
    X1 = np.ones((N, 1))
    sf = np.ones(N)
    nz = ~(counts == 0).all(axis=0)
    nzi = np.nonzero(nz)[0]
    c_nz = counts[:, nzi]
    for i in range(niter):
        _, mu_hat, gw, _ = _fit_genewise(c_nz, c_nz.astype(float), sf, X1, min_mu, min_disp, max_disp, beta_tol, n_jobs)
        use = gw > 10 * min_disp
        if not use.any():
            break
        mean_disp = trim_mean(gw[use], proportiontocut=0.001)
        fitted = np.full(len(nzi), mean_disp)
        sq, prior_var = dispersion_prior(gw, fitted, N, 1, min_disp)
        mp, _ = alpha_mle(c_nz, X1, mu_hat, fitted, min_disp, max_disp, prior_disp_var=float(prior_var), cr_reg=True,
                          prior_reg=True, n_jobs=n_jobs)
        disp = np.clip(mp, min_disp, max_disp)
        out = np.log(gw) > np.log(fitted) + 2 * np.sqrt(sq)
        disp[out] = gw[out]
        old_sf = sf.copy()
        base = mu_hat / old_sf[:, None]

        def objective(p):
            s = np.exp(p - np.mean(p))
            nll = nb_nll_genes(c_nz, base * s[:, None], disp)
            return np.sum(nll[nll < np.quantile(nll, quant)])

        res = minimize(objective, np.log(old_sf), method="Powell")
        sf = np.exp(res.x - np.mean(res.x))
        if not res.success:
            break
        if (i > 1) and np.sum((np.log(old_sf) - np.log(sf)) ** 2) < 1e-4:
            break
    return sf


def size_factors_control(counts, control_mask):
    """Median-of-ratios size factors restricted to control genes (dds.py:640-650, 692-703)."""
    lm, keep = logmeans_and_filter(counts)
    keep = keep & np.asarray(control_mask, dtype=bool)
    with np.errstate(divide="ignore"):
        lc = np.log(counts)
    return np.exp(np.median(lc[:, keep] - lm[keep], axis=1))


def size_factors_poscounts(counts, control_mask=None):
    """``fit_size_factors(fit_type="poscounts")`` (dds.py:655-680): geometric means over the positive
    counts only, per-sample median over the positive entries of the usable genes, normalised to a
    geometric mean of 1."""
    counts = np.asarray(counts)
    log_counts = np.zeros_like(counts, dtype=float)
    np.log(counts, out=log_counts, where=counts != 0)
    logmeans = log_counts.mean(0)
    mask = (~np.isinf(logmeans)) & (logmeans > 0)
    if control_mask is not None:
        mask = mask & np.asarray(control_mask, dtype=bool)

    def one(x):
        m = np.logical_and(mask, x > 0)
        return np.exp(np.median(np.log(x[m]) - logmeans[m]))

    sf = np.apply_along_axis(one, 1, counts)
    return sf / np.exp(np.mean(np.log(sf)))


# --------------------------------------------------------------------------
# a3/a4  method-of-moments initial dispersions    utils.py:814-885, dds.py:1140-1162
# --------------------------------------------------------------------------


def _ols_fit_predict(X: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """Least-squares fitted values X (X^+ Y) (sklearn LinearRegression(fit_intercept=False))."""
    coef, *_ = np.linalg.lstsq(X, Y, rcond=None)
    return X @ coef


def rough_dispersions(normed: np.ndarray, X: np.ndarray) -> np.ndarray:
    """Rough dispersion from OLS residuals (utils.py:814-853)."""
    n, p = X.shape
    if n == p:
        raise ValueError(
            "The number of samples and the number of design variables are "
            "equal, i.e., there are no replicates to estimate the "
            "dispersion. Please use a design with fewer variables."
        )
    yhat = np.maximum(_ols_fit_predict(X, normed), 1)
    a = (((normed - yhat) ** 2 - yhat) / ((n - p) * yhat**2)).sum(0)
    return np.maximum(a, 0)


def moments_dispersions(normed: np.ndarray, sf: np.ndarray) -> np.ndarray:
    """Moments dispersion (utils.py:856-885)."""
    normed = normed[:, ~(normed == 0).all(axis=0)]
    s_mean_inv = (1 / np.asarray(sf)).mean()
    mu = normed.mean(0)
    sigma = normed.var(0, ddof=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.nan_to_num((sigma - s_mean_inv * mu) / mu**2)


def mom_dispersions(normed, X, sf, min_disp, max_disp):
    """clip(min(rough, moments)) (dds.py:1149-1162)."""
    return np.clip(
        np.minimum(rough_dispersions(normed, X), moments_dispersions(normed, sf)),
        min_disp,
        max_disp,
    )


# --------------------------------------------------------------------------
# a5  linear-model mu_hat                         utils.py:682-715
# --------------------------------------------------------------------------


def lin_reg_mu(counts, sf, X, min_mu):
    """mu_hat = max(sf * OLS-fit(counts/sf), min_mu), gene by gene (utils.py:682-715).

    The reference fits ``sklearn.linear_model.LinearRegression(fit_intercept=False)`` per gene; sklearn's dense path is
    ``scipy.linalg.lstsq(X, y, cond=max(X.shape) * eps)`` on the gene's vector and ``X @ coef`` for the prediction.
    The same two calls per gene give the bit-identical mu_hat (a batched lstsq over all genes differs in the last
    bit on a quarter of the entries, which is enough to move the stopping point of ~0.06 % of the L-BFGS-B runs that
    consume it - measured, DESIGN.md par. 7)."""
    from scipy.linalg import lstsq

    counts = np.asarray(counts)
    cond = max(X.shape) * np.finfo(np.float64).eps
    out = np.empty(counts.shape, dtype=np.float64)
    for g in range(counts.shape[1]):
        coef = lstsq(X, counts[:, g] / sf, cond=cond)[0]
        out[:, g] = np.maximum(sf * (X @ coef.T), min_mu)
    return out


# --------------------------------------------------------------------------
# NB negative log-likelihood                      utils.py:163-270
# --------------------------------------------------------------------------


def nb_nll(counts, mu, alpha: float) -> float:
    """Scalar-alpha branch of utils.nb_nll (utils.py:216-234)."""
    n = len(counts)
    a1 = 1 / alpha
    logbinom = gammaln(counts + a1) - gammaln(counts + 1) - gammaln(a1)
    return n * a1 * np.log(alpha) + (
        -logbinom + (counts + a1) * np.log(a1 + mu) - counts * np.log(mu)
    ).sum()


def dnb_nll(counts, mu, alpha: float) -> float:
    """d nll / d alpha (utils.py:237-270)."""
    a1 = 1 / alpha
    return -(
        a1**2
        * (
            polygamma(0, a1)
            - polygamma(0, counts + a1)
            + np.log(1 + mu * alpha)
            + (counts - mu) / (mu + a1)
        ).sum()
    )


def vec_nb_nll(counts, mu, alpha):
    """NLL over a grid of alpha (1-D mu) or of mu (2-D mu) (grid_search.py:7-51)."""
    n = len(counts)
    a1 = 1 / alpha
    logbinom = gammaln(counts[:, None] + a1) - gammaln(counts + 1)[:, None] - gammaln(a1)
    if mu.ndim == 1:
        return n * a1 * np.log(alpha) + (
            -logbinom
            + (counts[:, None] + a1) * np.log(mu[:, None] + a1)
            - (counts * np.log(mu))[:, None]
        ).sum(0)
    return n * a1 * np.log(alpha) + (
        -logbinom + (counts[:, None] + a1) * np.log(mu + a1) - (counts[:, None] * np.log(mu))
    ).sum(0)


# --------------------------------------------------------------------------
# a6/a7  dispersion MLE / MAP                     utils.py:441-564, grid_search.py:54-142
# --------------------------------------------------------------------------


def grid_fit_alpha(counts, X, mu, alpha_hat, min_disp, max_disp, prior_disp_var=None,
                   cr_reg=True, prior_reg=False, grid_length=100) -> float:
    """Two-level 1-D grid search; returns log(alpha) (grid_search.py:54-142)."""
    lo, hi = np.log(min_disp), np.log(max_disp)
    grid = np.linspace(lo, hi, grid_length)

    def loss(la):
        alpha = np.exp(la)
        W = mu[:, None] / (1 + mu[:, None] * alpha)
        reg = 0
        if cr_reg:
            reg = reg + 0.5 * np.linalg.slogdet(
                (X.T[:, :, None] * W).transpose(2, 0, 1) @ X
            )[1]
        if prior_reg:
            reg = reg + (np.log(alpha) - np.log(alpha_hat)) ** 2 / (2 * prior_disp_var)
        return vec_nb_nll(counts, mu, alpha) + reg

    ll = loss(grid)
    k = np.argmin(ll)
    delta = grid[1] - grid[0]
    fine = np.linspace(grid[k] - delta, grid[k] + delta, grid_length)
    ll = loss(fine)
    return fine[np.argmin(ll)]


def alpha_mle_gene(counts, X, mu, alpha_hat, min_disp, max_disp, prior_disp_var=None,
                   cr_reg=True, prior_reg=False, return_info=False, optimizer="L-BFGS-B"):
    """One gene's dispersion fit by L-BFGS-B (or, optimizer="BFGS", unbounded BFGS) in log(alpha)
    (utils.py:441-564).

    On ``success == False`` the grid search is called with the reference's six
    positional arguments only, i.e. *without* the prior (utils.py:556-564).
    """
    la_hat = np.log(alpha_hat)

    def loss(la):
        alpha = np.exp(la)
        reg = 0
        if cr_reg:
            W = mu / (1 + mu * alpha)
            reg += 0.5 * np.linalg.slogdet((X.T * W) @ X)[1]
        if prior_reg:
            reg += (la - la_hat) ** 2 / (2 * prior_disp_var)
        return nb_nll(counts, mu, alpha) + reg

    def dloss(la):
        alpha = np.exp(la)
        rg = 0
        if cr_reg:
            W = mu / (1 + mu * alpha)
            dW = -(W**2)
            rg += (0.5 * (np.linalg.inv((X.T * W) @ X) * ((X.T * dW) @ X)).sum()) * alpha
        if prior_reg:
            rg += (la - la_hat) / prior_disp_var
        return alpha * dnb_nll(counts, mu, alpha) + rg

    res = minimize(
        lambda x: loss(x[0]),
        x0=np.asarray([la_hat]),
        jac=lambda x: np.asarray([dloss(x[0])]),
        method=optimizer,
        bounds=[(np.log(min_disp), np.log(max_disp))] if optimizer == "L-BFGS-B" else None,
    )
    if res.success:
        out = np.exp(res.x[0])
    else:
        out = np.exp(grid_fit_alpha(counts, X, mu, alpha_hat, min_disp, max_disp))
    if return_info:
        return out, bool(res.success), res
    return out, bool(res.success)


def _alpha_chunk(args):
    counts, X, mu, ah, min_disp, max_disp, pv, cr, pr = args[:9]
    opt = args[9] if len(args) > 9 else "L-BFGS-B"
    import warnings

    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = [
            alpha_mle_gene(counts[:, j], X, mu[:, j], ah[j], min_disp, max_disp, pv, cr, pr, optimizer=opt)
            for j in range(counts.shape[1])
        ]
    return np.array([o[0] for o in out]), np.array([o[1] for o in out], dtype=bool)


def _run_chunks(fn, chunks, n_jobs):
    if n_jobs is None or n_jobs <= 1 or len(chunks) <= 1:
        return [fn(c) for c in chunks]
    from joblib import Parallel, delayed

    return Parallel(n_jobs=n_jobs, backend="loky")(delayed(fn)(c) for c in chunks)


def alpha_mle(counts, X, mu, alpha_hat, min_disp, max_disp, prior_disp_var=None,
              cr_reg=True, prior_reg=False, n_jobs=1, chunk=256, optimizer="L-BFGS-B"):
    """All genes (default_inference.py:126-161).  Returns (alpha[G], converged[G])."""
    G = counts.shape[1]
    chunks = [
        (counts[:, s:s + chunk], X, mu[:, s:s + chunk], alpha_hat[s:s + chunk], min_disp,
         max_disp, prior_disp_var, cr_reg, prior_reg, optimizer)
        for s in range(0, G, chunk)
    ]
    res = _run_chunks(_alpha_chunk, chunks, n_jobs)
    if not res:
        return np.zeros(0), np.zeros(0, dtype=bool)
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


# --------------------------------------------------------------------------
# a10/a11  IRLS NB-GLM fit                        utils.py:273-438, grid_search.py:145-221
# --------------------------------------------------------------------------


def grid_fit_beta(counts, sf, X, disp, min_mu=0.5, grid_length=60, min_beta=-30, max_beta=30):
    """2-D two-level grid search on beta (p == 2 only) (grid_search.py:145-221)."""
    xg = np.linspace(min_beta, max_beta, grid_length)
    yg = np.linspace(min_beta, max_beta, grid_length)
    ll = np.zeros((grid_length, grid_length))

    def loss(beta):
        mu = np.maximum(sf[:, None] * np.exp(X @ beta.T), min_mu)
        return vec_nb_nll(counts, mu, disp) + 0.5 * (1e-6 * beta**2).sum(1)

    for i, x in enumerate(xg):
        ll[i, :] = loss(np.array([[x, y] for y in yg]))
    k = np.unravel_index(np.argmin(ll, axis=None), ll.shape)
    delta = xg[1] - xg[0]
    fx = np.linspace(xg[k[0]] - delta, xg[k[0]] + delta, grid_length)
    fy = np.linspace(yg[k[1]] - delta, yg[k[1]] + delta, grid_length)
    for i, x in enumerate(fx):
        ll[i, :] = loss(np.array([[x, y] for y in fy]))
    k = np.unravel_index(np.argmin(ll, axis=None), ll.shape)
    return np.array([fx[k[0]], fy[k[1]]])


def _irls_fallback(counts, sf, X, disp, beta_init, min_mu, min_beta, max_beta, optimizer="L-BFGS-B"):
    """L-BFGS-B (or unbounded BFGS) (+ grid for p<=2) rescue when IRLS diverges (utils.py:374-413)."""
    p = X.shape[1]
    ridge = np.diag(np.repeat(1e-6, p))

    def f(beta):
        mu_ = np.maximum(sf * np.exp(X @ beta), min_mu)
        return nb_nll(counts, mu_, disp) + 0.5 * (ridge @ beta**2).sum()

    def df(beta):
        mu_ = np.maximum(sf * np.exp(X @ beta), min_mu)
        return -X.T @ counts + ((1 / disp + counts) * mu_ / (1 / disp + mu_)) @ X + ridge @ beta

    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize(f, beta_init, jac=df, method=optimizer,
                       bounds=[(min_beta, max_beta)] * p if optimizer == "L-BFGS-B" else None)
    beta = res.x
    if not res.success and p <= 2:
        beta = grid_fit_beta(counts, sf, X, disp)
    return beta, bool(res.success)


def irls_gene(y, sf, X, disp, start, min_mu=0.5, beta_tol=1e-8, min_beta=-30, max_beta=30, maxiter=250,
              optimizer="L-BFGS-B"):
    """One gene's NB log-link GLM by IRLS, in the reference's own operation sequence (utils.py:340-438) so that the
    result is bit-identical to ``utils.irls_solver``: ``start`` = (Q, R) of the design when it has full rank, else None
    (utils.py:349-357; rank and QR depend on the design only and are hoisted out of the gene loop).
    Returns (beta[p], mu[N] UNclamped, hat diagonal[N], converged, sweeps)."""
    p = X.shape[1]
    if start is not None:  # utils.py:349-353
        Q, R = start
        with np.errstate(divide="ignore"):
            beta_init = sp_solve(R, Q.T @ np.log(y / sf + 0.1))
    else:  # utils.py:354-357
        beta_init = np.zeros(p)
        with np.errstate(divide="ignore"):
            beta_init[0] = np.log(y / sf).mean()
    beta = beta_init
    ridge = np.diag(np.repeat(1e-6, p))
    mu = np.maximum(sf * np.exp(X @ beta), min_mu)
    dev, ratio, sweeps, converged = 1000.0, 1.0, 0, True
    while ratio > beta_tol:
        W = mu / (1.0 + mu * disp)
        z = np.log(mu / sf) + (y - mu) / mu
        bh = sp_solve((X.T * W) @ X + ridge, X.T @ (W * z), assume_a="pos")
        sweeps += 1
        if sum(np.abs(bh) > max_beta) > 0 or sweeps >= maxiter:  # utils.py:374-413
            beta, converged = _irls_fallback(y, sf, X, disp, beta_init, min_mu, min_beta, max_beta, optimizer)
            mu = np.maximum(sf * np.exp(X @ beta), min_mu)
            break
        beta = bh
        mu = np.maximum(sf * np.exp(X @ beta), min_mu)
        old = dev
        dev = -2 * nb_nll(y, mu, disp)
        ratio = np.abs(dev - old) / (np.abs(dev) + 0.1)
    # hat diagonal with the CLAMPED mu (utils.py:427-433); returned mu is UNclamped (utils.py:435-437)
    W = mu / (1.0 + mu * disp)
    h = np.einsum("ij,jk,ki->i", X, np.linalg.inv((X.T * W[None, :]) @ X + ridge), X.T)
    sq = np.sqrt(W)
    return beta, sf * np.exp(X @ beta), sq * h * sq, converged, sweeps


def _irls_chunk(args):
    counts, sf, X, disp, start, min_mu, beta_tol, min_beta, max_beta, maxiter, optimizer = args
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = [irls_gene(counts[:, g], sf, X, disp[g], start, min_mu, beta_tol, min_beta, max_beta, maxiter, optimizer)
               for g in range(counts.shape[1])]
    return (np.array([o[0] for o in out]).reshape(len(out), X.shape[1]),
            np.array([o[1] for o in out]).reshape(len(out), X.shape[0]),
            np.array([o[2] for o in out]).reshape(len(out), X.shape[0]),
            np.array([o[3] for o in out], dtype=bool), np.array([o[4] for o in out], dtype=int))


def irls(counts, sf, X, disp, min_mu=0.5, beta_tol=1e-8, min_beta=-30, max_beta=30, maxiter=250,
         return_iters=False, optimizer="L-BFGS-B", n_jobs=1, chunk=256):
    """NB log-link GLM by IRLS for all genes (utils.py:273-438 per gene, default_inference.py:83-124 over genes).
    Returns (beta[G,p], mu[N,G] (UNclamped), H[N,G], converged[G])."""
    counts = np.asarray(counts)
    N, G = counts.shape
    p = X.shape[1]
    disp = np.broadcast_to(np.asarray(disp, dtype=float), (G,))
    start = np.linalg.qr(X) if np.linalg.matrix_rank(X) == p else None  # utils.py:349-350, hoisted
    chunks = [(counts[:, s:s + chunk], sf, X, disp[s:s + chunk], start, min_mu, beta_tol, min_beta, max_beta, maxiter,
               optimizer) for s in range(0, G, chunk)]
    res = _run_chunks(_irls_chunk, chunks, n_jobs)
    if not res:
        z = np.zeros((N, 0))
        return (np.zeros((0, p)), z, z.copy(), np.zeros(0, dtype=bool)) + ((np.zeros(0, dtype=int),) if return_iters else ())
    beta = np.concatenate([r[0] for r in res])
    mu = np.concatenate([r[1] for r in res]).T
    Hd = np.concatenate([r[2] for r in res]).T
    conv = np.concatenate([r[3] for r in res])
    if return_iters:
        return beta, mu, Hd, conv, np.concatenate([r[4] for r in res])
    return beta, mu, Hd, conv


# --------------------------------------------------------------------------
# a8  dispersion trend                            default_inference.py:200-230, dds.py:1199-1299
# --------------------------------------------------------------------------


def trend_gamma_glm(cov: np.ndarray, targets: np.ndarray):
    """2-coefficient gamma GLM disp ~ a0 + a1*cov (default_inference.py:200-230)."""
    import pandas as pd

    # the reference builds the regressors through pandas (Series.to_frame, insert of an integer intercept column,
    # .values): the resulting array's memory layout decides which BLAS kernel ``A @ c`` runs, i.e. the last bit of mu -
    # built the same way here so that the fit is bit-identical (default_inference.py:203-206)
    frame = pd.Series(np.asarray(cov)).to_frame()
    frame.insert(0, "intercept", 1)
    A = frame.values
    targets = np.asarray(targets)

    def loss(c):
        mu = A @ c
        return np.nanmean(targets / mu + np.log(mu), axis=0)

    def grad(c):
        mu = A @ c
        return -np.nanmean(((targets / mu - 1)[:, None] * A) / mu[:, None], axis=0)

    with np.errstate(all="ignore"):
        res = minimize(loss, x0=np.array([1.0, 1.0]), jac=grad, method="L-BFGS-B",
                       bounds=[(1e-12, np.inf)])
    return res.x, A @ res.x, bool(res.success)


def mean_trend(genewise: np.ndarray, min_disp: float) -> float:
    """Trimmed-mean trend (dds.py:1277-1299). ``genewise`` may contain NaN (zero genes)."""
    with np.errstate(invalid="ignore"):
        sel = genewise[genewise > 10 * min_disp]
    return float(trim_mean(sel, proportiontocut=0.001))


def fit_parametric_trend(genewise_nz: np.ndarray, normed_means_nz: np.ndarray, glm=None):
    """Iterated gamma-GLM trend over non-zero genes (dds.py:1199-1275).

    Returns (coeffs[2] or None on failure, n_outer_iterations).
    """
    with np.errstate(divide="ignore"):
        cov_all = 1 / normed_means_nz
    ok = ~(np.isinf(cov_all) | np.isnan(cov_all))  # dds.py:1225-1231
    sel = np.nonzero(ok)[0]
    old = np.array([0.1, 0.1])
    coeffs = np.array([1.0, 1.0])
    n_it = 0
    while (coeffs > 1e-10).all() and (np.log(np.abs(coeffs / old)) ** 2).sum() >= 1e-6:
        old = coeffs
        if glm is not None:  # a plugged-in Inference gets pandas Series, as dds.py:1212-1245 passes them
            import pandas as pd

            coeffs, pred, conv = glm(pd.Series(cov_all[sel]), pd.Series(genewise_nz[sel]))
            pred = np.asarray(pred)
        else:
            coeffs, pred, conv = trend_gamma_glm(cov_all[sel], genewise_nz[sel])
        coeffs = np.asarray(coeffs)
        n_it += 1
        if not conv or (coeffs <= 1e-10).any():
            return None, n_it
        r = genewise_nz[sel] / pred
        sel = sel[~((r < 1e-4) | (r >= 15))]
    return coeffs, n_it


# --------------------------------------------------------------------------
# a9  dispersion prior                            dds.py:840-884, utils.py:1210-1227
# --------------------------------------------------------------------------


def mean_absolute_deviation(x):
    c = np.median(x)
    return np.median(np.abs(x - c)) / norm.ppf(0.75)


def dispersion_prior(genewise_nz, fitted_nz, N, p, min_disp):
    """(squared_logres, prior_disp_var) (dds.py:866-884)."""
    res = np.log(genewise_nz) - np.log(fitted_nz)
    above = genewise_nz >= 100 * min_disp
    sq = mean_absolute_deviation(res[above]) ** 2
    return sq, np.maximum(sq - polygamma(1, (N - p) / 2), 0.25)


# --------------------------------------------------------------------------
# a12  Cook's distances                           utils.py:567-679, 888-960, dds.py:986-1040
# --------------------------------------------------------------------------


def design_cells(X: np.ndarray):
    """Group samples by identical design rows: (cell_id[N], cell_size[n_cells])."""
    _, inv, cnt = np.unique(X, axis=0, return_inverse=True, return_counts=True)
    return np.asarray(inv).reshape(-1), cnt


def trimmed_mean(x, trim=0.1, axis=0):
    s = np.sort(x, axis=axis)
    n = x.shape[axis]
    nt = math.floor(n * trim)
    return np.take(s, np.arange(nt, n - nt), axis).mean(axis)


def _trimfn(n):
    return 2 if n >= 23.5 else 1 if n >= 3.5 else 0


def trimmed_cell_variance(normed, cell_id):
    """max over cells of scaled trimmed variance (utils.py:602-650)."""
    ratios = (1 / 3, 1 / 4, 1 / 8)
    scales = (2.04, 1.86, 1.51)
    levels = np.unique(cell_id)
    sq = np.zeros_like(normed)
    for lv in levels:
        m = cell_id == lv
        t = ratios[_trimfn(m.sum())]
        sq[m, :] = normed[m, :] - trimmed_mean(normed[m, :], trim=t, axis=0)[None, :]
    sq **= 2
    var = np.zeros((len(levels), normed.shape[1]))
    for i, lv in enumerate(levels):
        m = cell_id == lv
        k = _trimfn(m.sum())
        var[i, :] = scales[k] * trimmed_mean(sq[m, :], trim=ratios[k], axis=0)
    return var.max(axis=0)


def robust_mom_disp(normed, X):
    """Trimmed method-of-moments dispersion, floor 0.04 (utils.py:914-960)."""
    cid, cnt = design_cells(X)
    three = cnt[cid] >= 3
    if three.any():
        v = trimmed_cell_variance(normed[three, :], cid[three])
    else:
        rm = trimmed_mean(normed, trim=0.125, axis=0)
        v = 1.51 * trimmed_mean((normed - rm) ** 2, trim=0.125, axis=0)
    m = normed.mean(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = (v - m) / m**2
    return np.maximum(a, 0.04)


def cooks_distance(counts, normed, X, mu, H):
    """Cook's distances for non-zero genes (dds.py:1001-1028)."""
    p = X.shape[1]
    a = robust_mom_disp(normed, X)
    V = mu + a[None, :] * mu**2
    with np.errstate(divide="ignore", invalid="ignore"):
        return (counts - mu) ** 2 / V / p * (H / (1 - H) ** 2)


# --------------------------------------------------------------------------
# a14  Wald test                                  utils.py:718-811
# --------------------------------------------------------------------------


def wald_test(X, disp, lfc, mu, ridge, contrast, lfc_null=0.0, alt_hypothesis=None):
    """Wald statistics for all genes (utils.py:718-811, default_inference.py:163-198).

    ``lfc`` is G x p, ``mu`` N x G.  Returns (pvalue[G], stat[G], se[G]).
    """
    G = lfc.shape[0]
    pv = np.full(G, np.nan)
    st = np.full(G, np.nan)
    se = np.full(G, np.nan)
    for g in range(G):
        Wg = mu[:, g] / (1 + mu[:, g] * disp[g])
        Mg = (X.T * Wg[None, :]) @ X  # the reference's own product (utils.py:771-772): bit-identical M
        if not np.isfinite(Mg).all():
            continue
        Hc = np.linalg.inv(Mg + ridge) @ contrast
        s = np.sqrt(Hc.T @ Mg @ Hc)
        b = lfc[g]
        with np.errstate(divide="ignore", invalid="ignore"):
            if alt_hypothesis is None:
                t = float(contrast @ (b - lfc_null) / s)
                q = 2 * norm.sf(np.abs(t))
            elif alt_hypothesis == "greater":
                t = contrast @ np.fmax((b - lfc_null) / s, 0)
                q = norm.sf(t)
            elif alt_hypothesis == "less":
                t = contrast @ np.fmin((b - lfc_null) / s, 0)
                q = norm.sf(np.abs(t))
            elif alt_hypothesis == "greaterAbs":
                t = contrast @ (np.sign(b) * np.fmax((np.abs(b) - lfc_null) / s, 0))
                q = 2 * norm.sf(np.abs(t))
            elif alt_hypothesis == "lessAbs":
                ta = contrast @ np.fmax((b + abs(lfc_null)) / s, 0)
                pa = norm.sf(ta)
                tb = contrast @ np.fmin((b - abs(lfc_null)) / s, 0)
                pb = norm.sf(np.abs(tb))
                t = min(ta, tb, key=abs)
                q = max(pa, pb)
            else:
                raise KeyError(alt_hypothesis)
        pv[g], st[g], se[g] = q, t, s
    return pv, st, se


# --------------------------------------------------------------------------
# orchestration                                   dds.py:516-562, 1042-1110, 1301-1458; ds.py:303-360
# --------------------------------------------------------------------------


@dataclass
class DeseqResult:
    """Field names follow the reference's AnnData schema."""

    size_factors: np.ndarray = None
    normed_means: np.ndarray = None
    non_zero: np.ndarray = None
    mom_dispersions: np.ndarray = None
    mu_hat: np.ndarray = None
    genewise_dispersions: np.ndarray = None
    genewise_converged: np.ndarray = None
    trend_coeffs: np.ndarray = None
    disp_function_type: str = "parametric"
    mean_disp: float = None
    fitted_dispersions: np.ndarray = None
    squared_logres: float = None
    prior_disp_var: float = None
    MAP_dispersions: np.ndarray = None
    MAP_converged: np.ndarray = None
    outlier_genes: np.ndarray = None
    dispersions: np.ndarray = None
    LFC: np.ndarray = None
    LFC_converged: np.ndarray = None
    mu_LFC: np.ndarray = None
    hat_diagonals: np.ndarray = None
    cooks: np.ndarray = None
    replaced: np.ndarray = None
    refitted: np.ndarray = None
    new_all_zeroes: np.ndarray = None
    replace_cooks: np.ndarray = None
    cooks_outlier: np.ndarray = None
    pvalue: np.ndarray = None
    stat: np.ndarray = None
    lfcSE: np.ndarray = None
    timings: dict = field(default_factory=dict)


class _OracleInference:
    """The oracle's own kernels behind the reference's ``Inference`` method names and keyword arguments
    (inference.py:9-362), so that ``deseq2(..., inference=obj)`` can drive any other implementation of that
    interface in ``DeseqDataSet``'s call order (dds.py:713-984, ds.py:303-360) — tests use it to run the
    engine's ``HipInference`` plug-in under the reference's orchestration."""

    def __init__(self, n_jobs=1, irls_maxiter=250):
        self.n_jobs = n_jobs
        self.irls_maxiter = irls_maxiter  # `maxiter` of Inference.irls (inference.py:46-119): tests lower it to push genes
                                          # through the rescue of utils.py:374-413

    def fit_rough_dispersions(self, normed_counts, design_matrix):
        return rough_dispersions(normed_counts, design_matrix)

    def fit_moments_dispersions(self, normed_counts, size_factors):
        return moments_dispersions(normed_counts, size_factors)

    def lin_reg_mu(self, counts, size_factors, design_matrix, min_mu):
        return lin_reg_mu(counts, size_factors, design_matrix, min_mu)

    def irls(self, counts, size_factors, design_matrix, disp, min_mu, beta_tol, **kw):
        return irls(counts, size_factors, design_matrix, disp, min_mu, beta_tol, maxiter=self.irls_maxiter,
                    n_jobs=self.n_jobs)

    def alpha_mle(self, counts, design_matrix, mu, alpha_hat, min_disp, max_disp, prior_disp_var=None,
                  cr_reg=True, prior_reg=False, **kw):
        return alpha_mle(counts, design_matrix, mu, alpha_hat, min_disp, max_disp, prior_disp_var=prior_disp_var,
                         cr_reg=cr_reg, prior_reg=prior_reg, n_jobs=self.n_jobs)

    def dispersion_trend_gamma_glm(self, covariates, targets):
        return trend_gamma_glm(np.asarray(covariates), np.asarray(targets))

    def wald_test(self, design_matrix, disp, lfc, mu, ridge_factor, contrast, lfc_null, alt_hypothesis=None):
        return wald_test(design_matrix, disp, lfc, mu, ridge_factor, contrast, lfc_null, alt_hypothesis)


def _fit_genewise(counts_nz, normed_nz, sf, X, min_mu, min_disp, max_disp, beta_tol, n_jobs, inf=None):
    """MoM -> mu_hat -> genewise alpha on non-zero genes (dds.py:713-797)."""
    inf = inf if inf is not None else _OracleInference(n_jobs)
    # dds.py:1149-1162: rough and moments estimates through the plug-in, min / clip on the host
    rde = inf.fit_rough_dispersions(normed_nz, X)
    mde = inf.fit_moments_dispersions(normed_nz, sf)
    mom = np.clip(np.minimum(rde, mde), min_disp, max_disp)
    n_cells = len(np.unique(X, axis=0))
    if n_cells == X.shape[1]:  # dds.py:747-756
        mu_hat = inf.lin_reg_mu(counts=counts_nz, size_factors=sf, design_matrix=X, min_mu=min_mu)
    else:  # dds.py:757-765
        _, mu_hat, _, _ = inf.irls(counts=counts_nz, size_factors=sf, design_matrix=X, disp=mom, min_mu=min_mu,
                                   beta_tol=beta_tol)
    gw, conv = inf.alpha_mle(counts=counts_nz, design_matrix=X, mu=mu_hat, alpha_hat=mom, min_disp=min_disp,
                             max_disp=max_disp)
    return mom, mu_hat, np.clip(gw, min_disp, max_disp), conv


def deseq2(counts, X, contrast=None, *, min_mu=0.5, min_disp=1e-8, max_disp=10.0,
           refit_cooks=True, min_replicates=7, beta_tol=1e-8, fit_type="parametric",
           lfc_null=0.0, alt_hypothesis=None, n_jobs=1, keep_layers=True, inference=None):
    """End-to-end restatement of ``DeseqDataSet.deseq2()`` + ``DeseqStats.run_wald_test()``.

    Follows dds.py:516-562 step by step, then ds.py:303-360.  ``counts`` is
    N x G non-negative integers, ``X`` the N x p design matrix (intercept first).
    ``inference``: an object with the reference's ``Inference`` interface (inference.py:9-362) that
    replaces the oracle's own per-gene kernels, as ``DeseqDataSet(inference=...)`` does (dds.py:323-336).
    """
    import time

    inf = inference if inference is not None else _OracleInference(n_jobs)

    counts = np.asarray(counts)
    X = np.asarray(X, dtype=float)
    N, G = counts.shape
    p = X.shape[1]
    max_disp = max(max_disp, N)  # dds.py:312
    if contrast is None:
        contrast = np.zeros(p)
        contrast[-1] = 1.0
    r = DeseqResult()
    T = r.timings
    t0 = time.perf_counter()

    # -- size factors (dds.py:692-708); iterative mode when every gene contains a zero (dds.py:682-690)
    if (counts == 0).any(0).all():
        warnings.warn("Every gene contains at least one zero, cannot compute log geometric means. "
                      "Switching to iterative mode.", UserWarning, stacklevel=2)
        sf = size_factors_iterative(counts, min_mu=min_mu, min_disp=min_disp, max_disp=max_disp, beta_tol=beta_tol,
                                    n_jobs=n_jobs)
        normed = counts / sf[:, None]
    else:
        sf, normed, _, _ = size_factors_ratio(counts)
    r.size_factors = sf
    r.normed_means = normed.mean(0)
    T["size_factors"] = time.perf_counter() - t0

    # -- genewise dispersions (dds.py:713-797)
    t = time.perf_counter()
    nz = ~(counts == 0).all(axis=0)
    nzi = np.nonzero(nz)[0]
    r.non_zero = nz
    c_nz = counts[:, nzi]
    mom, mu_hat, gw, gconv = _fit_genewise(c_nz, normed[:, nzi], sf, X, min_mu, min_disp,
                                           max_disp, beta_tol, n_jobs, inf)
    r.mom_dispersions = _scatter(G, nzi, mom)
    r.genewise_dispersions = _scatter(G, nzi, gw)
    r.genewise_converged = _scatter(G, nzi, gconv.astype(float))
    if keep_layers:
        r.mu_hat = _scatter2(N, G, nzi, mu_hat)
    T["genewise"] = time.perf_counter() - t

    # -- trend (dds.py:799-838)
    t = time.perf_counter()
    coeffs = None
    if fit_type == "parametric":
        coeffs, _ = fit_parametric_trend(gw, r.normed_means[nzi], glm=inf.dispersion_trend_gamma_glm)
        if coeffs is None:
            warnings.warn("The dispersion trend curve fitting did not converge. "
                          "Switching to a mean-based dispersion trend.", UserWarning, stacklevel=2)
    if coeffs is not None:
        r.trend_coeffs = coeffs
        r.disp_function_type = "parametric"
        fitted = np.full(G, np.nan)
        fitted[nzi] = coeffs[0] + coeffs[1] / r.normed_means[nzi]
    else:
        r.disp_function_type = "mean"
        r.mean_disp = mean_trend(r.genewise_dispersions, min_disp)
        fitted = np.full(G, r.mean_disp)
    r.fitted_dispersions = fitted
    T["trend"] = time.perf_counter() - t

    # -- prior (dds.py:840-884)
    r.squared_logres, r.prior_disp_var = dispersion_prior(gw, fitted[nzi], N, p, min_disp)
    r.prior_disp_var = float(r.prior_disp_var)

    # -- MAP (dds.py:886-935)
    t = time.perf_counter()
    mp, mconv = inf.alpha_mle(counts=c_nz, design_matrix=X, mu=mu_hat, alpha_hat=fitted[nzi], min_disp=min_disp,
                              max_disp=max_disp, prior_disp_var=r.prior_disp_var, cr_reg=True, prior_reg=True)
    r.MAP_dispersions = _scatter(G, nzi, np.clip(mp, min_disp, max_disp))
    r.MAP_converged = _scatter(G, nzi, mconv.astype(float))
    disp = r.MAP_dispersions.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        out_g = np.log(r.genewise_dispersions) > np.log(fitted) + 2 * np.sqrt(r.squared_logres)
    disp[out_g] = r.genewise_dispersions[out_g]
    r.outlier_genes = out_g
    r.dispersions = disp
    T["MAP"] = time.perf_counter() - t

    # -- LFC (dds.py:937-984)
    t = time.perf_counter()
    beta, mu, Hd, lconv = inf.irls(counts=c_nz, size_factors=sf, design_matrix=X, disp=disp[nzi], min_mu=min_mu,
                                   beta_tol=beta_tol)
    r.LFC = _rows(G, p, nzi, beta)
    r.LFC_converged = _scatter(G, nzi, lconv.astype(float))
    T["LFC"] = time.perf_counter() - t

    # -- Cook's (dds.py:986-1040)
    t = time.perf_counter()
    ck_nz = cooks_distance(c_nz, normed[:, nzi], X, mu, Hd)
    cooks = _scatter2(N, G, nzi, ck_nz)
    if keep_layers:
        r.mu_LFC, r.hat_diagonals, r.cooks = mu, Hd, cooks
    T["cooks"] = time.perf_counter() - t

    # -- refit (dds.py:1042-1064, 1301-1458)
    t = time.perf_counter()
    cutoff = f_dist.ppf(0.99, p, N - p)
    cid, cnt = design_cells(X)
    r.replaced = np.zeros(G, dtype=bool)
    r.refitted = np.zeros(G, dtype=bool)
    r.new_all_zeroes = np.zeros(G, dtype=bool)
    replace_cooks = None
    if refit_cooks:
        replaceable = cnt[cid] >= min_replicates
        if replaceable.sum() > 0:
            with np.errstate(invalid="ignore"):
                idx = cooks > cutoff
            r.replaced = idx.any(axis=0)
            if r.replaced.sum() > 0:
                rp = np.nonzero(r.replaced)[0]
                sub = counts[:, rp].copy()
                tbm = trimmed_mean(sub / sf[:, None], trim=0.2, axis=0)
                repl = (tbm[:, None] * sf[None, :]).astype(int).T  # truncation, dds.py:1344-1352
                m = replaceable[:, None] & idx[:, rp]
                sub[m] = repl[m]
                naz = (sub == 0).all(axis=0)
                r.new_all_zeroes[rp[naz]] = True
                r.refitted[rp[~naz]] = True
                if naz.sum() > 0:  # dds.py:1380-1383
                    r.normed_means[rp[naz]] = 0
                    r.LFC[rp[naz], :] = 0
                if r.refitted.sum() > 0:
                    rf = rp[~naz]
                    s_c = sub[:, ~naz]
                    s_n = s_c / sf[:, None]
                    inf1 = inference if inference is not None else _OracleInference(1)
                    _, s_mu, s_gw, _ = _fit_genewise(s_c, s_n, sf, X, min_mu, min_disp,
                                                     max_disp, beta_tol, 1, inf1)
                    s_means = s_n.mean(0)
                    if r.disp_function_type == "parametric":
                        s_fit = r.trend_coeffs[0] + r.trend_coeffs[1] / s_means
                    else:
                        s_fit = np.full(len(rf), r.mean_disp)
                    s_map, _ = inf1.alpha_mle(counts=s_c, design_matrix=X, mu=s_mu, alpha_hat=s_fit,
                                              min_disp=min_disp, max_disp=max_disp,
                                              prior_disp_var=r.prior_disp_var, cr_reg=True, prior_reg=True)
                    s_disp = np.clip(s_map, min_disp, max_disp)
                    s_out = np.log(s_gw) > np.log(s_fit) + 2 * np.sqrt(r.squared_logres)
                    s_disp[s_out] = s_gw[s_out]
                    s_beta, _, _, _ = inf1.irls(counts=s_c, size_factors=sf, design_matrix=X, disp=s_disp,
                                                min_mu=min_mu, beta_tol=beta_tol)
                    r.normed_means[rf] = s_means
                    r.LFC[rf, :] = s_beta
                    r.genewise_dispersions[rf] = s_gw
                    r.fitted_dispersions[rf] = s_fit
                    r.dispersions[rf] = s_disp
                    replace_cooks = cooks.copy()
                    for col in rf:
                        replace_cooks[replaceable, col] = 0.0
    r.replace_cooks = replace_cooks if keep_layers else None
    T["refit"] = time.perf_counter() - t

    # -- cooks_outlier (dds.py:1066-1110)
    use_for_max = cnt[cid] >= 3
    src = replace_cooks if (refit_cooks and r.refitted.sum() > 0 and replace_cooks is not None) else cooks
    with np.errstate(invalid="ignore"):
        co = (src[use_for_max, :] > cutoff).any(axis=0)
    if co.any():
        pos = cooks[:, co].argmax(0)
        cc = counts[:, co]
        co[co] = (cc > cc[pos, np.arange(len(pos))]).sum(0) < 3
    r.cooks_outlier = co

    # -- Wald (ds.py:303-360)
    t = time.perf_counter()
    with np.errstate(invalid="ignore", over="ignore"):
        mu_w = np.exp(X @ r.LFC.T) * sf[:, None]
    ridge = np.diag(np.repeat(1e-6, p))
    pv, st, se = inf.wald_test(design_matrix=X, disp=r.dispersions, lfc=r.LFC, mu=mu_w, ridge_factor=ridge,
                               contrast=np.asarray(contrast, float), lfc_null=np.log(2) * lfc_null,
                               alt_hypothesis=alt_hypothesis)
    if refit_cooks and r.replaced.sum() > 0:
        z = r.new_all_zeroes
        se[z], st[z], pv[z] = 0.0, 0.0, 1.0
    r.pvalue, r.stat, r.lfcSE = pv, st, se
    T["wald"] = time.perf_counter() - t
    T["total"] = time.perf_counter() - t0
    return r


def _scatter(G, idx, v):
    out = np.full(G, np.nan)
    out[idx] = v
    return out


def _scatter2(N, G, idx, v):
    out = np.full((N, G), np.nan)
    out[:, idx] = v
    return out


def _rows(G, p, idx, v):
    out = np.full((G, p), np.nan)
    out[idx, :] = v
    return out


# --------------------------------------------------------------------------
# DeseqStats.summary() tail                       ds.py:266-286, 486-550
# --------------------------------------------------------------------------


def bh_adjust(p):
    """Benjamini-Hochberg (scipy.stats.false_discovery_control(method='bh'))."""
    p = np.asarray(p, dtype=float)
    m = len(p)
    order = np.argsort(p)
    # scipy's operation order, `ps *= m / i` (a quotient first): (p * m) / i rounds differently and moves an adjusted value
    # across alpha when p-values tie on the boundary (found by a tie-heavy test vector: 208 vs 206 rejections)
    ps = p[order] * (m / np.arange(1, m + 1))
    ps = np.minimum.accumulate(ps[::-1])[::-1]
    out = np.empty(m)
    out[order] = np.clip(ps, 0, 1)
    return out


def p_value_adjustment(pvalue):
    """BH over non-NaN p-values (ds.py:529-542)."""
    padj = np.full(len(pvalue), np.nan)
    ok = ~np.isnan(pvalue)
    if ok.any():
        padj[ok] = bh_adjust(pvalue[ok])
    return padj


# --------------------------------------------------------------------------
# variance stabilising transformation        SURVEY.md §8(f)-4
# --------------------------------------------------------------------------


def vst(counts, X, use_design=False, fit_type="parametric", min_mu=0.5, min_disp=1e-8, max_disp=10.0,
        beta_tol=1e-8, n_jobs=1):
    """DeseqDataSet.vst (dds.py:349-514): size factors, genewise dispersions and trend fitted with an
    intercept-only design (or the full one), then the closed-form transform of the normalised counts.
    Returns (vst_counts N x G, info)."""
    counts = np.asarray(counts)
    N, G = counts.shape
    max_disp = max(max_disp, N)
    Xd = np.asarray(X, dtype=float) if use_design else np.ones((N, 1))
    sf, normed, _, _ = size_factors_ratio(counts)
    nz = ~(counts == 0).all(axis=0)
    nzi = np.nonzero(nz)[0]
    _, _, gw, _ = _fit_genewise(counts[:, nzi], normed[:, nzi], sf, Xd, min_mu, min_disp, max_disp, beta_tol, n_jobs)
    gw_all = _scatter(G, nzi, gw)
    info = dict(size_factors=sf, genewise_dispersions=gw_all)
    coeffs = None
    if fit_type == "parametric":
        coeffs, _ = fit_parametric_trend(gw, normed.mean(0)[nzi])
    if coeffs is not None:
        a0, a1 = coeffs
        info["trend_coeffs"] = coeffs
        out = np.log2((1 + a1 + 2 * a0 * normed + 2 * np.sqrt(a0 * normed * (1 + a1 + a0 * normed))) / (4 * a0))
    else:
        use = gw_all > 10 * min_disp
        mean_disp = trim_mean(gw_all[use], proportiontocut=0.001)
        info["mean_disp"] = mean_disp
        out = (2 * np.arcsinh(np.sqrt(mean_disp * normed)) - np.log(mean_disp) - np.log(4)) / np.log(2)
    return out, info


def vst_transform_new(counts_new, counts_train, info):
    """``vst_transform(counts)`` for new samples (dds.py:471-514): size factors from the training
    logmeans / usable genes (preprocessing.py:59-102), then the fitted closed form."""
    lm, keep = logmeans_and_filter(np.asarray(counts_train))
    counts_new = np.asarray(counts_new)
    with np.errstate(divide="ignore", invalid="ignore"):
        sf = np.exp(np.median((np.log(counts_new) - lm)[:, keep], axis=1))
    normed = counts_new / sf[:, None]
    if "trend_coeffs" in info:
        a0, a1 = info["trend_coeffs"]
        return np.log2((1 + a1 + 2 * a0 * normed + 2 * np.sqrt(a0 * normed * (1 + a1 + a0 * normed))) / (4 * a0))
    md = info["mean_disp"]
    return (2 * np.arcsinh(np.sqrt(md * normed)) - np.log(md) - np.log(4)) / np.log(2)


# --------------------------------------------------------------------------
# apeGLM LFC shrinkage                      SURVEY.md §8(f)-2
# --------------------------------------------------------------------------


def nbinom_fn(beta, X, counts, size, offset, prior_no_shrink_scale, prior_scale, shrink_index=1):
    """NB negative log-likelihood + apeGLM prior (utils.py:1145-1207)."""
    p = X.shape[-1]
    shrink_mask = np.zeros(p)
    shrink_mask[shrink_index] = 1
    no_shrink_mask = np.ones(p) - shrink_mask
    xbeta = X @ beta
    prior = ((beta * no_shrink_mask) ** 2 / (2 * prior_no_shrink_scale**2)).sum() + np.log1p(
        (beta[shrink_index] / prior_scale) ** 2)
    nll = (counts * xbeta - (counts + size) * np.logaddexp(xbeta + offset, np.log(size))).sum(0)
    return prior - nll


def grid_fit_shrink_beta(counts, offset, X, size, prior_no_shrink_scale, prior_scale, scale_cnst,
                         grid_length=60, min_beta=-30, max_beta=30):
    """Two-level 2-D grid search (grid_search.py:224-320); its loss always shrinks coefficient 1."""
    def loss(b):
        return nbinom_fn(b, X, counts, size, offset, prior_no_shrink_scale, prior_scale) / scale_cnst

    xg = np.linspace(min_beta, max_beta, grid_length)
    yg = np.linspace(min_beta, max_beta, grid_length)
    ll = np.array([[loss(np.array([x, y])) for y in yg] for x in xg])
    i, j = np.unravel_index(np.argmin(ll, axis=None), ll.shape)
    delta = xg[1] - xg[0]
    fx = np.linspace(xg[i] - delta, xg[i] + delta, grid_length)
    fy = np.linspace(yg[j] - delta, yg[j] + delta, grid_length)
    ll = np.array([[loss(np.array([x, y])) for y in fy] for x in fx])
    i, j = np.unravel_index(np.argmin(ll, axis=None), ll.shape)
    return np.array([fx[i], fy[j]])


def nbinom_glm_gene(X, counts, size, offset, prior_no_shrink_scale, prior_scale, shrink_index=1,
                    optimizer="L-BFGS-B"):
    """utils.nbinomGLM (utils.py:990-1142): (beta, inv_hessian, converged).  ``optimizer``: 'L-BFGS-B' (what ds.py:407
    passes), 'BFGS' or 'Newton-CG' - handed to scipy.optimize.minimize with the Hessian only for 'Newton-CG'
    (utils.py:1112-1121; the ftol / gtol options are unknown to some of the methods: scipy warns and ignores them).

    Kept as in the reference: the Hessian adds ``np.diag(h)`` where ``h`` is already a diagonal matrix,
    i.e. the VECTOR of prior curvatures is broadcast onto every row (utils.py:1099-1110)."""
    p = X.shape[-1]
    shrink_mask = np.zeros(p)
    shrink_mask[shrink_index] = 1
    no_shrink_mask = np.ones(p) - shrink_mask
    beta_init = np.ones(p) * 0.1 * (-1) ** (np.arange(p))
    cnst = np.maximum(nbinom_fn(np.zeros(p), X, counts, size, offset, prior_no_shrink_scale, prior_scale,
                                shrink_index), 1)

    def f(beta):
        return nbinom_fn(beta, X, counts, size, offset, prior_no_shrink_scale, prior_scale, shrink_index) / cnst

    def df(beta):
        xbeta = X @ beta
        d_neg_prior = (beta * no_shrink_mask / prior_no_shrink_scale**2
                       + 2 * beta * shrink_mask / (prior_scale**2 + beta[shrink_index] ** 2))
        d_nll = (counts - (counts + size) / (1 + size * np.exp(-xbeta - offset))) @ X
        return (d_neg_prior - d_nll) / cnst

    def ddf(beta, c=1):
        xbeta = X @ beta
        e = np.exp(xbeta + offset)
        frac = (counts + size) * size * e / (size + e) ** 2
        h11 = 1 / prior_no_shrink_scale**2
        h22 = 2 * (prior_scale**2 - beta[shrink_index] ** 2) / (prior_scale**2 + beta[shrink_index] ** 2) ** 2
        h = np.diag(no_shrink_mask * h11 + shrink_mask * h22)
        return 1 / c * ((X.T * frac) @ X + np.diag(h))

    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize(f, beta_init, jac=df, hess=(lambda b: ddf(b, cnst)) if optimizer == "Newton-CG" else None,
                       method=optimizer, options={"ftol": 1e-8, "gtol": 1e-8})
    beta, converged = res.x, res.success
    if not converged and p == 2:
        beta = grid_fit_shrink_beta(counts, offset, X, size, prior_no_shrink_scale, prior_scale, cnst)
    return beta, np.linalg.inv(ddf(beta, 1)), converged


def fit_prior_var(lfc, se, min_var=1e-6, max_var=400.0):
    """DeseqStats._fit_prior_var (ds.py:551-590): zero of the moment-matching equation."""
    from scipy.optimize import root_scalar

    keep = ~np.isnan(lfc)
    S, D = lfc[keep] ** 2, se[keep] ** 2

    def objective(a):
        coeff = 1 / (2 * (a + D) ** 2)
        return ((S - D) * coeff).sum() / coeff.sum() - a

    if objective(min_var) < 0:
        return min_var
    return root_scalar(objective, bracket=(min_var, max_var)).root


def lfc_shrink(counts, X, res, coeff_idx, adapt=True):
    """DeseqStats.lfc_shrink (ds.py:363-447) on a DeseqResult: shrunken (LFC column, lfcSE, converged).

    ``res.lfcSE`` must come from the Wald test of the same coefficient (contrast = unit vector)."""
    counts = np.asarray(counts)
    nz = np.asarray(res.non_zero, dtype=bool)
    size = 1.0 / res.dispersions
    offset = np.log(res.size_factors)
    prior_scale = 1
    if adapt:
        prior_scale = np.minimum(np.sqrt(fit_prior_var(res.LFC[:, coeff_idx], res.lfcSE)), 1)
    lfc, se = res.LFC[:, coeff_idx].copy(), np.array(res.lfcSE, dtype=float)
    conv = np.full(len(nz), np.nan)
    for g in np.nonzero(nz)[0]:
        b, ih, cv = nbinom_glm_gene(X, counts[:, g], size[g], offset, 15, prior_scale, coeff_idx)
        lfc[g], se[g], conv[g] = b[coeff_idx], np.sqrt(np.abs(ih[coeff_idx, coeff_idx])), cv
    return lfc, se, conv, float(prior_scale)


def lowess(features, targets, frac=2.0 / 3.0, iters=3):
    """Robust locally weighted regression, restating utils.lowess (utils.py:1379-1442) line by line."""
    features = np.asarray(features, dtype=float)
    targets = np.asarray(targets, dtype=float)
    n = len(features)
    r = int(np.ceil(frac * n))
    h = np.maximum(np.array([np.sort(np.abs(features - features[i]))[r] for i in range(n)]), 1e-12)
    with np.errstate(invalid="ignore", divide="ignore"):
        w = np.clip(np.abs(np.nan_to_num((features[:, None] - features[None, :]) / h)), 0.0, 1.0)
    w = (1 - w**3) ** 3
    yest = np.zeros(n)
    delta = np.ones(n)
    for _ in range(iters):
        for i in range(n):
            weights = delta * w[:, i]
            b = np.array([np.sum(weights * targets), np.sum(weights * targets * features)])
            A = np.array([[np.sum(weights), np.sum(weights * features)],
                          [np.sum(weights * features), np.sum(weights * features * features)]])
            beta = np.linalg.lstsq(A, b, rcond=None)[0]
            yest[i] = beta[0] + beta[1] * features[i]
        residuals = targets - yest
        s = np.median(np.abs(residuals))
        if s == 0:
            delta = (np.abs(residuals) > 0).astype(float)
        else:
            delta = np.clip(residuals / (6.0 * s), -1, 1)
        delta = (1 - delta**2) ** 2
    return yest


def independent_filtering(base_mean, pvalue, alpha=0.05):
    """DeseqStats._independent_filtering (ds.py:486-527) -> (padj, info).

    50 candidate baseMean cut-offs (quantiles theta of base_mean), BH over the genes above each one,
    lowess-smoothed number of rejections picks the cut-off; padj is the BH column of that cut-off.
    """
    base_mean = np.asarray(base_mean, dtype=float)
    pvalue = np.asarray(pvalue, dtype=float)
    G = len(base_mean)
    lower_quantile = np.mean(base_mean == 0)
    upper_quantile = 0.95 if lower_quantile < 0.95 else 1
    theta = np.linspace(lower_quantile, upper_quantile, 50)
    cutoffs = np.quantile(base_mean, theta)
    result = np.full((G, len(theta)), np.nan)
    for i, cutoff in enumerate(cutoffs):
        use = (base_mean >= cutoff) & (~np.isnan(pvalue))
        if use.any():
            result[use, i] = bh_adjust(pvalue[use])
    with np.errstate(invalid="ignore"):
        num_rej = (result < alpha).sum(0).astype(int)
    lowess_res = lowess(theta, num_rej, frac=1 / 5)
    if num_rej.max() <= 10:
        j = 0
    else:
        residual = num_rej[num_rej > 0] - lowess_res[num_rej > 0]
        thresh = lowess_res.max() - np.sqrt(np.mean(residual**2))
        j = int(np.where(num_rej > thresh)[0][0]) if np.any(num_rej > thresh) else 0
    return result[:, j], dict(theta=theta, cutoffs=cutoffs, num_rej=num_rej, lowess=lowess_res, j=j)


def summary(res, contrast, alpha=0.05, cooks_filter=True, independent_filter=True):
    """DeseqStats.summary() after the Wald test (ds.py:266-286): Cook's filtering of the p-values
    (ds.py:543-549), adjusted p-values, and the result columns (log2 scale for the LFCs)."""
    contrast = np.asarray(contrast, dtype=float)
    pvalue = np.array(res.pvalue, dtype=float)
    if cooks_filter:
        pvalue[np.asarray(res.cooks_outlier, dtype=bool)] = np.nan
    base_mean = np.asarray(res.normed_means, dtype=float)
    if independent_filter:
        padj, info = independent_filtering(base_mean, pvalue, alpha)
    else:
        padj, info = p_value_adjustment(pvalue), {}
    return dict(baseMean=base_mean, log2FoldChange=np.asarray(res.LFC) @ contrast / np.log(2),
                lfcSE=np.asarray(res.lfcSE) / np.log(2), stat=np.asarray(res.stat), pvalue=pvalue, padj=padj,
                info=info)


# --------------------------------------------------------------------------
# synthetic inputs                                SURVEY.md §8(d) / BASELINE.md §3
# --------------------------------------------------------------------------


# the synthetic-data generator lives outside the oracle (it defines workloads, not reference behaviour)
from pydeseq2_amd.synth import make_design, synth_counts  # noqa: E402,F401


def default_n_jobs() -> int:
    return os.cpu_count() or 1
