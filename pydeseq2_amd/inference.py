"""HipInference — MI355X drop-in for PyDESeq2's ``Inference`` plug-in interface.

Mirrors ``pydeseq2.inference.Inference`` (pydeseq2/inference.py:9-362) method for method
(same names, argument meaning, return shapes/orders and error behaviour as
``DefaultInference``, pydeseq2/default_inference.py:14-264), so it can be passed as
``DeseqDataSet(..., inference=HipInference())`` / ``DeseqStats(..., inference=...)``
(dds.py:226, 323-336; ds.py:143, 194-207).  Every method is a thin marshalling layer over
one ``dsq_inf_*`` entry point of libdeseq_hip.so; there is no CPU implementation behind it.
"""
from __future__ import annotations

import ctypes as C
from typing import Literal

import numpy as np

from ._lib import ALT, GENE_MAJOR, I32, I64, SAMPLE_MAJOR, Context, _PinnedPool

_vp, c_int, c_double = C.c_void_p, C.c_int, C.c_double
_OPTIMIZER = {"L-BFGS-B": 0, "BFGS": 1}  # the `optimizer` argument of dsq_inf_irls / dsq_inf_alpha_mle
_SHRINK_OPTIMIZER = {"L-BFGS-B": 0, "BFGS": 1, "Newton-CG": 2}  # ... of dsq_inf_lfc_shrink_nbinom_glm (utils.py:1028-1030)


def _counts_arg(counts):
    """(array kept alive, count_type, layout) for an N x G count matrix."""
    a = np.asarray(counts)
    if a.dtype.kind == "f":
        if (a % 1 != 0).any():
            raise ValueError("The count matrix should only contain integers.")
        a = a.astype(np.int64)
    elif a.dtype.kind == "b":
        a = a.astype(np.int32)
    elif a.dtype.kind not in "iu":
        raise ValueError("The count matrix should only contain numbers.")
    if a.dtype not in (np.int32, np.int64):
        a = a.astype(np.int64)
    ctype = I32 if a.dtype == np.int32 else I64
    if a.flags.c_contiguous:
        return a, ctype, SAMPLE_MAJOR
    if a.flags.f_contiguous:
        return a, ctype, GENE_MAJOR
    return np.ascontiguousarray(a), ctype, SAMPLE_MAJOR


def _matrix_arg(m):
    a = np.asarray(m, dtype=np.float64)
    if a.flags.c_contiguous:
        return a, SAMPLE_MAJOR
    if a.flags.f_contiguous:
        return a, GENE_MAJOR
    return np.ascontiguousarray(a), SAMPLE_MAJOR


def _vec(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64))


class HipInference:
    """GPU implementation of the 8 inference routines of the DESeq2 pipeline.

    Parameters
    ----------
    device : int
        HIP device ordinal (one process per GPU).
    n_cpus : int, optional
        Accepted for interface compatibility (``DeseqDataSet`` sets it when present,
        dds.py:324-333); ignored.
    """

    def __init__(self, device: int = 0, n_cpus: int | None = None, ctx: Context | None = None,
                 pinned_outputs: bool = True):
        self.ctx = ctx if ctx is not None else Context(device)
        self._n_cpus = n_cpus
        # The N x G layers a method returns (mu_hat, mu, hat diagonals: 0.5 GB each at 60k x 1k) are written into
        # page-locked buffers that go back to a free list when the caller drops the array (the reference copies them into
        # its own layers and does): a fresh np.empty costs 40 ms of page faults per layer, five times its PCIe transfer.
        self._pinned = _PinnedPool(self.ctx) if pinned_outputs else None
        self._pageable_first = 6  # layers of one size handed out pageable before the pool starts page-locking (two fits: mu_hat, mu, hat each)

    def _layer(self, G, N):
        """Host buffer of a G x N output layer (returned to the caller as its N x G transpose view).

        The first layers of a size are pageable, later ones come from the page-locked pool: hipHostMalloc of 480 MB takes
        82 ms on the GPU box (tools/probes/pin_probe.py), the DMA into a page-locked buffer 8.5 ms, a copy into fresh
        pageable memory 37 ms through the runtime and ~17 ms through the engine's own staged, multi-threaded copy-out (round 6,
        csrc pc_download_rows) - so the first two deseq2() through the plug-in (three such layers each) do not pay 250 ms for
        buffers they may never reuse (first fit at c3: 160-225 -> ~105 ms), and a caller that keeps coming back finds them
        page-locked from its third fit on.  (Page-locking them on a helper thread meanwhile was measured and dropped: it
        serialises with the main thread's page faults and digests in the kernel - first fit 249 -> 290 ms.)"""
        nbytes = G * N * 8
        if self._pinned is None or nbytes < (1 << 20):
            return np.empty((G, N))
        slab = self._pinned.take_free(nbytes)
        if slab is None:
            seen = self.__dict__.setdefault("_layer_sizes_seen", {})
            seen[nbytes] = seen.get(nbytes, 0) + 1
            if seen[nbytes] <= self._pageable_first:
                return np.empty((G, N))
            slab = self._pinned.take(nbytes)
        return slab.view(0, G * N, np.float64).reshape(G, N)

    # ---- the device cache behind the entry points (include/deseq_hip.h, csrc/dsq_plugin_cache.h)
    def cache_stats(self) -> dict:
        """Counters of the content-addressed device cache of this context."""
        v = (c_double * 13)()
        self.ctx.call("dsq_plugin_cache_stats", v, 13)
        keys = ("hits", "misses", "adopted_outputs", "evictions", "h2d_bytes", "d2h_bytes", "hash_ms", "resident_bytes",
                "pooled_free_bytes", "resident_matrices", "device_mallocs", "budget_bytes", "verified_hits")
        return {k: (round(float(x), 3) if k == "hash_ms" else int(x)) for k, x in zip(keys, v)}

    def cache_config(self, enabled: bool | None = None, budget_bytes: int | None = None):
        """Switch the cache on / off or set its byte budget (None: unchanged)."""
        self.ctx.call("dsq_plugin_cache_config", -1 if enabled is None else int(bool(enabled)),
                      C.c_longlong(-1 if budget_bytes is None else int(budget_bytes)))

    def cache_clear(self):
        """Free every resident matrix and pooled device buffer of the plug-in entry points."""
        self.ctx.call("dsq_plugin_cache_clear")

    def __del__(self):
        try:
            if self._pinned is not None:
                self._pinned.close()
        except Exception:
            pass

    @property
    def n_cpus(self):  # noqa: D102
        return self._n_cpus

    @n_cpus.setter
    def n_cpus(self, n_cpus):
        self._n_cpus = n_cpus

    # ------------------------------------------------------------------ lin_reg_mu
    def lin_reg_mu(self, counts, size_factors, design_matrix, min_mu):
        """See ``Inference.lin_reg_mu`` (inference.py:13-44). Returns mu_hat (N x G)."""
        y, ct, lay = _counts_arg(counts)
        N, G = y.shape
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        sf = _vec(size_factors)
        out = self._layer(G, N)
        self.ctx.call("dsq_inf_lin_reg_mu", _vp(y.ctypes.data), ct, lay, _vp(sf.ctypes.data),
                      _vp(X.ctypes.data), N, G, X.shape[1], c_double(min_mu), _vp(out.ctypes.data))
        return out.T

    # ------------------------------------------------------------------ irls
    def irls(self, counts, size_factors, design_matrix, disp, min_mu, beta_tol, min_beta=-30,
             max_beta=30, optimizer: Literal["BFGS", "L-BFGS-B"] = "L-BFGS-B", maxiter=250):
        """See ``Inference.irls`` (inference.py:46-119).

        Returns (beta G x p, mu N x G, hat diagonals N x G, converged G).
        """
        assert optimizer in ["BFGS", "L-BFGS-B"]
        y, ct, lay = _counts_arg(counts)
        N, G = y.shape
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        P = X.shape[1]
        sf, d = _vec(size_factors), _vec(disp)
        beta, mu, H = np.empty((G, P)), self._layer(G, N), self._layer(G, N)
        conv = np.empty(G, dtype=np.uint8)
        # last argument: the rescue of diverged genes - bounded L-BFGS-B (0) or scipy's BFGS restated (1)
        self.ctx.call("dsq_inf_irls2", _vp(y.ctypes.data), ct, lay, _vp(sf.ctypes.data), _vp(X.ctypes.data),
                      _vp(d.ctypes.data), N, G, P, c_double(min_mu), c_double(beta_tol), c_double(min_beta),
                      c_double(max_beta), int(maxiter), _vp(beta.ctypes.data), _vp(mu.ctypes.data),
                      _vp(H.ctypes.data), _vp(conv.ctypes.data), _OPTIMIZER[optimizer])
        return beta, mu.T, H.T, conv.astype(bool)

    # ------------------------------------------------------------------ alpha_mle
    def alpha_mle(self, counts, design_matrix, mu, alpha_hat, min_disp, max_disp, prior_disp_var=None,
                  cr_reg=True, prior_reg=False, optimizer: Literal["BFGS", "L-BFGS-B"] = "L-BFGS-B"):
        """See ``Inference.alpha_mle`` (inference.py:121-178). Returns (alpha G, converged G)."""
        assert optimizer in ["BFGS", "L-BFGS-B"]
        if prior_reg and prior_disp_var is None:
            raise ValueError("Sigma_prior is required for prior regularization")
        y, ct, lay = _counts_arg(counts)
        N, G = y.shape
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        m, mlay = _matrix_arg(mu)
        ah = _vec(alpha_hat)
        out, conv = np.empty(G), np.empty(G, dtype=np.uint8)
        self.ctx.call("dsq_inf_alpha_mle2", _vp(y.ctypes.data), ct, lay, _vp(X.ctypes.data), _vp(m.ctypes.data),
                      mlay, _vp(ah.ctypes.data), N, G, X.shape[1], c_double(min_disp), c_double(max_disp),
                      c_double(prior_disp_var if prior_disp_var is not None else 1.0), int(bool(cr_reg)),
                      int(bool(prior_reg)), _vp(out.ctypes.data), _vp(conv.ctypes.data), _OPTIMIZER[optimizer])
        return out, conv.astype(bool)

    # ------------------------------------------------------------------ wald_test
    def wald_test(self, design_matrix, disp, lfc, mu, ridge_factor, contrast, lfc_null, alt_hypothesis=None):
        """See ``Inference.wald_test`` (inference.py:180-235). Returns (pvals, stats, se)."""
        if alt_hypothesis not in ALT:
            raise KeyError(alt_hypothesis)
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        N, P = X.shape
        m, mlay = _matrix_arg(mu)
        G = m.shape[1]
        d, b = _vec(disp), np.ascontiguousarray(np.asarray(lfc, dtype=np.float64))
        r = np.ascontiguousarray(np.asarray(ridge_factor, dtype=np.float64))
        c = _vec(contrast)
        p, s, se = np.empty(G), np.empty(G), np.empty(G)
        self.ctx.call("dsq_inf_wald_test", _vp(X.ctypes.data), _vp(d.ctypes.data), _vp(b.ctypes.data),
                      _vp(m.ctypes.data), mlay, _vp(r.ctypes.data), _vp(c.ctypes.data), c_double(float(lfc_null)),
                      ALT[alt_hypothesis], N, G, P, _vp(p.ctypes.data), _vp(s.ctypes.data), _vp(se.ctypes.data))
        return p, s, se

    # ------------------------------------------------------------------ MoM dispersions
    def fit_rough_dispersions(self, normed_counts, design_matrix):
        """See ``Inference.fit_rough_dispersions`` (inference.py:237-259)."""
        v, lay = _matrix_arg(normed_counts)
        N, G = v.shape
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        out = np.empty(G)
        self.ctx.call("dsq_inf_fit_rough_dispersions", _vp(v.ctypes.data), lay, _vp(X.ctypes.data), N, G,
                      X.shape[1], _vp(out.ctypes.data))
        return out

    def fit_moments_dispersions(self, normed_counts, size_factors):
        """See ``Inference.fit_moments_dispersions`` (inference.py:261-282)."""
        v, lay = _matrix_arg(normed_counts)
        N, G = v.shape
        sf = _vec(size_factors)
        out, zero = np.empty(G), np.empty(G, dtype=np.uint8)
        self.ctx.call("dsq_inf_fit_moments_dispersions2", _vp(v.ctypes.data), lay, _vp(sf.ctypes.data), N, G,
                      _vp(out.ctypes.data), _vp(zero.ctypes.data))
        # utils.py:878 drops the all-zero genes before taking the moments (the flags come from the device: no host pass
        # over the matrix); dds.py:1149 never passes one
        return out[zero == 0] if zero.any() else out

    # ------------------------------------------------------------------ trend
    def dispersion_trend_gamma_glm(self, covariates, targets):
        """See ``Inference.dispersion_trend_gamma_glm`` (inference.py:284-308).

        Same objective, start point, bounds and optimiser as default_inference.py:200-230, run by the
        trend kernel of the device pipeline in its one-fit mode (``dsq_inf_dispersion_trend_gamma_glm``).
        Returns (coeffs[2] intercept first, predictions, converged).
        """
        cov = _vec(getattr(covariates, "values", covariates))
        tgt = _vec(getattr(targets, "values", targets))
        if cov.shape != tgt.shape or cov.ndim != 1:
            raise ValueError("covariates and targets must be vectors of the same length")
        coeffs, pred, ok = np.empty(2), np.empty(len(cov)), c_int(0)
        self.ctx.call("dsq_inf_dispersion_trend_gamma_glm", _vp(cov.ctypes.data), _vp(tgt.ctypes.data), len(cov),
                      _vp(coeffs.ctypes.data), _vp(pred.ctypes.data), C.byref(ok))
        return coeffs, pred, bool(ok.value)

    # ------------------------------------------------------------------ grid searches (grid_search.py)
    def grid_fit_alpha(self, counts, design_matrix, mu, min_disp, max_disp):
        """``grid_search.grid_fit_alpha`` (grid_search.py:54-142) for every gene: log(alpha) G."""
        y, ct, lay = _counts_arg(counts)
        N, G = y.shape
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        m, mlay = _matrix_arg(mu)
        out = np.empty(G)
        self.ctx.call("dsq_inf_grid_fit_alpha", _vp(y.ctypes.data), ct, lay, _vp(X.ctypes.data), _vp(m.ctypes.data),
                      mlay, N, G, X.shape[1], c_double(min_disp), c_double(max_disp), _vp(out.ctypes.data))
        return out

    def grid_fit_beta(self, counts, size_factors, design_matrix, disp, min_mu=0.5, grid_length=60, min_beta=-30,
                      max_beta=30):
        """``grid_search.grid_fit_beta`` (grid_search.py:145-221) for every gene: beta G x 2."""
        y, ct, lay = _counts_arg(counts)
        N, G = y.shape
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        if X.shape[1] != 2:
            raise ValueError("grid_fit_beta is defined for two design columns")
        sf, d = _vec(size_factors), _vec(disp)
        out = np.empty((G, 2))
        self.ctx.call("dsq_inf_grid_fit_beta", _vp(y.ctypes.data), ct, lay, _vp(sf.ctypes.data), _vp(X.ctypes.data),
                      _vp(d.ctypes.data), N, G, c_double(min_mu), int(grid_length), c_double(min_beta),
                      c_double(max_beta), _vp(out.ctypes.data))
        return out

    # ------------------------------------------------------------------ apeGLM shrinkage
    def lfc_shrink_nbinom_glm(self, design_matrix, counts, size, offset, prior_no_shrink_scale, prior_scale,
                              optimizer, shrink_index):
        """See ``Inference.lfc_shrink_nbinom_glm`` (inference.py:306-362, default_inference.py:232-264).
        Returns (beta G x p, inv_hessian G x p x p, converged G)."""
        if optimizer not in _SHRINK_OPTIMIZER:
            raise ValueError(f"optimizer: one of {sorted(_SHRINK_OPTIMIZER)} (utils.py:1028-1030)")
        y, ct, lay = _counts_arg(counts)
        N, G = y.shape
        X = np.ascontiguousarray(np.asarray(design_matrix, dtype=np.float64))
        P = X.shape[1]
        sz, off = _vec(size), _vec(offset)
        beta, invh, conv = np.empty((G, P)), np.empty((G, P, P)), np.empty(G, dtype=np.uint8)
        self.ctx.call("dsq_inf_lfc_shrink_nbinom_glm2", _vp(y.ctypes.data), ct, lay, _vp(X.ctypes.data),
                      _vp(sz.ctypes.data), _vp(off.ctypes.data), N, G, P, c_double(prior_no_shrink_scale),
                      c_double(prior_scale), int(shrink_index), _vp(beta.ctypes.data), _vp(invh.ctypes.data),
                      _vp(conv.ctypes.data), _SHRINK_OPTIMIZER[optimizer])
        return beta, invh, conv.astype(bool)
