"""``DeseqStats.summary()`` after the Wald test (pydeseq2/ds.py:266-286, 486-549) on the device.

Cook's filtering of the p-values, Benjamini-Hochberg adjustment with (or without) independent
filtering, and the result columns.  The 50 BH passes of the reference's independent filtering
collapse into one device sort plus prefix counts (csrc/dsq_k_summary.hip); only the lowess choice of
the cut-off — a 50-point smoother — runs on the host.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, DeviceArray

_vp = C.c_void_p


def lowess(x, y, frac=2.0 / 3.0, iters=3):
    """Robust locally weighted linear regression with tricube weights (what utils.lowess computes,
    utils.py:1379-1442), vectorised over the evaluation points.

    Bandwidth h_i = distance to the ceil(frac*n)-th nearest x (index r of the sorted distances, which
    include the point itself), weights (1 - |d/h|^3)^3, ``iters`` passes with bisquare robustness
    weights from the residuals / (6 * median |residual|).  Each local fit is the minimum-norm solution
    of its 2x2 weighted normal equations (``numpy.linalg.lstsq`` in the reference, a pseudo-inverse here).
    """
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    n = len(x)
    r = int(np.ceil(frac * n))
    dist = np.abs(x[:, None] - x[None, :])
    h = np.maximum(np.sort(dist, axis=0)[r], 1e-12)  # per evaluation point (column)
    w = np.clip(np.nan_to_num(dist / h[None, :]), 0.0, 1.0)  # w[j, i]: point j seen from evaluation point i
    w = (1 - w**3) ** 3
    est = np.zeros(n)
    delta = np.ones(n)
    for _ in range(iters):
        wt = delta[:, None] * w  # [point j, evaluation i]
        s0, s1, s2 = wt.sum(0), (wt * x[:, None]).sum(0), (wt * (x * x)[:, None]).sum(0)
        t0, t1 = (wt * y[:, None]).sum(0), (wt * (y * x)[:, None]).sum(0)
        # 2x2 normal equations: Cramer's rule where they are well conditioned, minimum-norm solution
        # (what numpy.linalg.lstsq returns in the reference) for the degenerate ones
        det = s0 * s2 - s1 * s1
        good = np.abs(det) > 1e-10 * np.maximum(np.abs(s0 * s2), 1e-300)
        safe = np.where(good, det, 1.0)
        b0 = np.where(good, (t0 * s2 - t1 * s1) / safe, 0.0)
        b1 = np.where(good, (s0 * t1 - s1 * t0) / safe, 0.0)
        if not good.all():
            bad = np.nonzero(~good)[0]
            A = np.stack([np.stack([s0[bad], s1[bad]], -1), np.stack([s1[bad], s2[bad]], -1)], -2)
            sol = np.einsum("ijk,ik->ij", np.linalg.pinv(A, rcond=2 * np.finfo(float).eps),
                            np.stack([t0[bad], t1[bad]], -1))
            b0[bad], b1[bad] = sol[:, 0], sol[:, 1]
        est = b0 + b1 * x
        res = y - est
        s = np.median(np.abs(res))
        delta = (np.abs(res) > 0).astype(float) if s == 0 else np.clip(res / (6.0 * s), -1, 1)
        delta = (1 - delta**2) ** 2
    return est


def choose_cutoff(theta, num_rej):
    """Index of the baseMean cut-off to use (ds.py:512-525)."""
    num_rej = np.asarray(num_rej, dtype=int)
    fit = lowess(theta, num_rej, frac=1 / 5)
    if num_rej.max() <= 10:
        return 0, fit
    pos = num_rej > 0
    residual = num_rej[pos] - fit[pos]
    thresh = fit.max() - np.sqrt(np.mean(residual**2))
    above = np.nonzero(num_rej > thresh)[0]
    return (int(above[0]) if len(above) else 0), fit


def adjusted_pvalues(ctx: Context, base_mean, pvalue, alpha=0.05, independent_filter=True):
    """padj for all genes (NaN where the gene has no p-value or falls below the chosen cut-off)."""
    base_mean = np.ascontiguousarray(base_mean, dtype=np.float64)
    pvalue = np.ascontiguousarray(pvalue, dtype=np.float64)
    G = len(base_mean)
    # device buffers are kept per context and gene count (hipMalloc / hipFree cost more than the kernels)
    cache = ctx.__dict__.setdefault("_padj_buffers", {})
    if G not in cache:
        cache.clear()
        cache[G] = (DeviceArray(ctx, (G,), np.float64), DeviceArray(ctx, (G,), np.float64),
                    DeviceArray(ctx, (G,), np.uint64), DeviceArray(ctx, (G,), np.int32),
                    DeviceArray(ctx, (G,), np.uint8), DeviceArray(ctx, (G,), np.float64))
    d_bm, d_p, d_sp, d_si, d_bins, d_padj = cache[G]
    ctx.h2d(d_bm.ptr, base_mean)
    ctx.h2d(d_p.ptr, pvalue)
    out = (C.c_double * 200)()
    n_valid = C.c_int(0)
    ctx.call("dsq_dev_padj_prepare", _vp(d_bm.ptr), _vp(d_p.ptr), G, C.c_double(alpha), _vp(d_sp.ptr),
             _vp(d_si.ptr), _vp(d_bins.ptr), out, C.byref(n_valid))
    o = np.array(out[:]).reshape(4, 50)
    info = dict(theta=o[0], cutoffs=o[1], num_rej=o[2].astype(int), m=o[3].astype(int), n_valid=n_valid.value)
    if independent_filter:
        j, fit = choose_cutoff(info["theta"], info["num_rej"])
        info.update(j=j, lowess=fit)
    else:
        j = -1
    ctx.call("dsq_dev_padj_finish", _vp(d_sp.ptr), _vp(d_si.ptr), _vp(d_bins.ptr), G, n_valid.value, int(j),
             _vp(d_padj.ptr))
    return d_padj.to_host(), info


def summary(res, contrast, alpha=0.05, cooks_filter=True, independent_filter=True, ctx: Context | None = None,
            device: int = 0):
    """Result columns of ``DeseqStats.summary()`` from a :class:`DeseqResult` (log2 scale for the LFCs).

    Returns a dict with baseMean, log2FoldChange, lfcSE, stat, pvalue, padj (+ ``info``: thetas, cut-offs,
    rejections per cut-off, the chosen index).
    """
    ctx = ctx if ctx is not None else Context(device)
    contrast = np.asarray(contrast, dtype=float)
    pvalue = np.array(res.pvalue, dtype=float)
    if cooks_filter:  # ds.py:543-549
        pvalue[np.asarray(res.cooks_outlier, dtype=bool)] = np.nan
    base_mean = np.asarray(res.normed_means, dtype=float)
    padj, info = adjusted_pvalues(ctx, base_mean, pvalue, alpha, independent_filter)
    return dict(baseMean=base_mean, log2FoldChange=np.asarray(res.LFC) @ contrast / np.log(2),
                lfcSE=np.asarray(res.lfcSE) / np.log(2), stat=np.asarray(res.stat), pvalue=pvalue, padj=padj,
                info=info)


# ------------------------------------------------------------------ apeGLM LFC shrinkage
def fit_prior_var(lfc, se, min_var=1e-6, max_var=400.0):
    """Prior variance of the apeGLM model from the MLE LFCs and their standard errors: the zero of the
    moment-matching equation of ``DeseqStats._fit_prior_var`` (ds.py:551-590)."""
    from scipy.optimize import root_scalar

    keep = ~np.isnan(lfc)
    s2, d2 = np.asarray(lfc, dtype=float)[keep] ** 2, np.asarray(se, dtype=float)[keep] ** 2

    def moment_gap(a):
        wgt = 1 / (2 * (a + d2) ** 2)
        return ((s2 - d2) * wgt).sum() / wgt.sum() - a

    if moment_gap(min_var) < 0:
        return min_var
    return root_scalar(moment_gap, bracket=(min_var, max_var)).root


def lfc_shrink(pipe, res, coeff_idx, adapt=True, prior_no_shrink_scale=15.0):
    """``DeseqStats.lfc_shrink`` (ds.py:363-447) on the pipeline's device-resident counts.

    ``res`` is the DeseqResult of ``pipe.deseq2(contrast=unit vector of coeff_idx)`` (its lfcSE feeds the
    adaptive prior).  Returns (shrunken LFC of the coefficient [G], its SE [G], converged [G] with NaN
    for all-zero genes, prior_scale)."""
    ctx, N, G, P = pipe.ctx, pipe.N, pipe.G, pipe.P
    prior_scale = 1.0
    if adapt:
        prior_scale = float(min(np.sqrt(fit_prior_var(res.LFC[:, coeff_idx], res.lfcSE)), 1.0))
    nz = np.asarray(res.non_zero, dtype=bool)
    disp = np.where(nz, res.dispersions, 1.0)  # all-zero genes are fitted too (cheap) and masked below
    d_size = DeviceArray.from_host(ctx, 1.0 / disp)
    d_off = DeviceArray.from_host(ctx, np.log(res.size_factors))
    # only inv_hessian[coeff][coeff] is used (the shrunken lfcSE, ds.py:424-433): 8 bytes per gene come back, not 8 p^2
    entry_only = P <= 12
    d_beta = DeviceArray(ctx, (G, P), np.float64)
    d_ih = DeviceArray(ctx, (G,) if entry_only else (G, P * P), np.float64)
    d_conv = DeviceArray(ctx, (G,), np.uint8)
    ctx.call("dsq_dev_lfc_shrink2", _vp(pipe.d_y.ptr), pipe.ldn, _vp(d_off.ptr), _vp(pipe.d_Xt.ptr), pipe.design.ldx,
             N, G, P, _vp(d_size.ptr), C.c_double(prior_no_shrink_scale), C.c_double(prior_scale), int(coeff_idx),
             _vp(d_beta.ptr), None if entry_only else _vp(d_ih.ptr), _vp(d_conv.ptr), _vp(d_ih.ptr) if entry_only else None)
    beta = d_beta.to_host()
    ihd = d_ih.to_host() if entry_only else d_ih.to_host().reshape(G, P, P)[:, coeff_idx, coeff_idx]
    lfc = np.where(nz, beta[:, coeff_idx], res.LFC[:, coeff_idx])
    se = np.where(nz, np.sqrt(np.abs(ihd)), res.lfcSE)
    conv = np.where(nz, d_conv.to_host().astype(float), np.nan)
    return lfc, se, conv, prior_scale
