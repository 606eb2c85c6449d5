"""Dispersion trend and prior: the two cross-gene steps of deseq2() (host side, O(G)).

Replaces DefaultInference.dispersion_trend_gamma_glm (default_inference.py:200-230),
DeseqDataSet._fit_parametric_dispersion_trend / _fit_mean_dispersion_trend
(dds.py:1199-1299) and fit_dispersion_prior (dds.py:840-884).  They need every gene at
once (an all-gather in the multi-GPU layout), touch 2-3 doubles per gene and are
latency-, not bandwidth-bound, so they run on the host between the per-gene kernels.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import minimize
from scipy.special import polygamma
from scipy.stats import norm, trim_mean


def gamma_glm_fit(cov: np.ndarray, targets: np.ndarray):
    """(coeffs[2], predictions, converged) of disp ~ a0 + a1*cov (gamma GLM, L-BFGS-B)."""
    A = np.column_stack([np.ones_like(cov), cov])

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


def fit_parametric_trend(genewise_nz: np.ndarray, normed_means_nz: np.ndarray, glm=gamma_glm_fit):
    """Iterated trend fit over the non-zero genes (dds.py:1216-1264). None on failure."""
    with np.errstate(divide="ignore"):
        cov_all = 1.0 / normed_means_nz
    sel = np.nonzero(~(np.isinf(cov_all) | np.isnan(cov_all)))[0]
    old = np.array([0.1, 0.1])
    coeffs = np.array([1.0, 1.0])
    while (coeffs > 1e-10).all() and (np.log(np.abs(coeffs / old)) ** 2).sum() >= 1e-6:
        old = coeffs
        coeffs, pred, conv = glm(cov_all[sel], genewise_nz[sel])
        coeffs = np.asarray(coeffs)
        if not conv or (coeffs <= 1e-10).any():
            return None
        r = genewise_nz[sel] / pred
        sel = sel[~((r < 1e-4) | (r >= 15))]
    return coeffs


def mean_trend(genewise_all: np.ndarray, min_disp: float) -> float:
    """Trimmed-mean trend (dds.py:1288-1293); NaN entries (all-zero genes) are ignored."""
    with np.errstate(invalid="ignore"):
        sel = genewise_all[genewise_all > 10 * min_disp]
    return float(trim_mean(sel, proportiontocut=0.001))


def dispersion_prior(genewise_nz, fitted_nz, n_obs: int, n_vars: int, min_disp: float):
    """(squared_logres, prior_disp_var) (dds.py:866-884, utils.py:1210-1227)."""
    res = np.log(genewise_nz) - np.log(fitted_nz)
    x = res[genewise_nz >= 100 * min_disp]
    mad = np.median(np.abs(x - np.median(x))) / norm.ppf(0.75)
    sq = mad**2
    return float(sq), float(np.maximum(sq - polygamma(1, (n_obs - n_vars) / 2), 0.25))
