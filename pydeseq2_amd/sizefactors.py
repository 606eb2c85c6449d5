"""``iterative`` size factors (pydeseq2/dds.py:1460-1548): the mode ``deseq2()`` falls back to when every
gene contains a zero (dds.py:682-690), and ``fit_size_factors(fit_type="iterative")``.

Outer loop (at most ``niter`` times): dispersions for the current size factors with an intercept-only
design — genewise fit, mean "trend", prior, MAP, dispersion outliers — all on the device kernels of the
main path; then a Powell search (scipy, exactly the reference's optimiser) over the log size factors
whose objective — the per-gene NLL under the rescaled means, summed over the genes below its 0.95
quantile — is evaluated on the device (``dsq_dev_nll_scaled``); only the G-vector of NLLs returns to the
host for the quantile cut.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np
from scipy.optimize import minimize
from scipy.stats import trim_mean

from ._lib import DeviceArray

_vp, c_double = C.c_void_p, C.c_double


def iterative_size_factors(pipe, niter: int = 10, quant: float = 0.95) -> np.ndarray:
    """Size factors [N] for the counts resident in ``pipe`` (any design: an intercept-only twin is used)."""
    from .pipeline import DeseqPipeline, DeseqResult

    ctx, N, G = pipe.ctx, pipe.N, pipe.G
    # intercept-only twin sharing the device-resident counts; of the pipeline's own class: on a gene shard the
    # cross-gene steps (trimmed mean of the dispersions, prior, the objective's quantile and sum) then run over the
    # genes of ALL ranks (`_all_genes_host`, `_prior`), and every rank walks the same Powell search
    p1 = type(pipe).__new__(type(pipe))
    p1.__dict__.update(pipe.__dict__)
    from ._design import DesignPack

    p1.design = DesignPack(np.ones((N, 1)), pipe.min_replicates)
    p1.P = 1
    p1._cells = None
    p1.d_Xt = DeviceArray.from_host(ctx, p1.design.Xt)
    p1.d_pinv = DeviceArray.from_host(ctx, p1.design.pinvXt)
    p1._pool_free, p1._pool_used = [], []
    try:
        nz = pipe._down_nonzero()
        nzi = np.nonzero(nz)[0]
        Gn = len(nzi)
        sf = np.ones(N)
        d_ones = DeviceArray.from_host(ctx, np.ones(N))
        for i in range(niter):
            p1._pool_reset()
            d_y = p1.d_y
            if Gn < G:
                d_idx = p1._up(nzi.astype(np.int32), np.int32)
                d_y = p1._dmat(Gn, np.int32)
                ctx.call("dsq_dev_gather_rows_i32", _vp(p1.d_y.ptr), p1.ldn, _vp(d_idx.ptr), Gn, N, _vp(d_y.ptr))
            S = p1._dev_slab(Gn)
            d_sf = p1._up(sf)
            # genewise dispersions; MoM start values on the RAW counts as in the reference
            ctx.call("dsq_dev_mom_raw", _vp(d_y.ptr), p1.ldn, _vp(d_ones.ptr), _vp(d_sf.ptr), _vp(p1.d_Xt.ptr),
                     _vp(p1.d_pinv.ptr), p1.design.ldx, N, Gn, 1, c_double(p1.min_disp), c_double(p1.max_disp),
                     _vp(S["nm"].ptr), _vp(S["mom"].ptr))
            d_mu = p1._dmat(Gn)
            ctx.call("dsq_dev_lin_mu", _vp(d_y.ptr), p1.ldn, _vp(d_sf.ptr), _vp(p1.d_Xt.ptr), _vp(p1.d_pinv.ptr),
                     p1.design.ldx, N, Gn, 1, c_double(p1.min_mu), _vp(d_mu.ptr))
            mh = type("MuHat", (), {})()
            mh.d_mu, mh.d_coef, mh.nll_const = d_mu, None, p1._dvec(Gn)
            ctx.call("dsq_dev_alpha_mle", _vp(d_y.ptr), _vp(d_mu.ptr), p1.ldn, _vp(p1.d_Xt.ptr), p1.design.ldx, N, Gn, 1,
                     _vp(S["mom"].ptr), c_double(p1.min_disp), c_double(p1.max_disp), c_double(1.0), 1, 0,
                     _vp(S["gw"].ptr), _vp(S["gconv"].ptr), None, _vp(mh.nll_const.ptr), 1)
            gw = np.clip(p1._all_genes_host(S["gw"], Gn), p1.min_disp, p1.max_disp)
            use = gw > 10 * p1.min_disp
            if not use.any():
                print("No genes have a dispersion above 10 * min_disp in iterative size factors.", file=sys.stderr)
                break
            mean_disp = float(trim_mean(gw[use], proportiontocut=0.001))
            ctx.call("dsq_dev_trend_eval", _vp(S["nm"].ptr), Gn, c_double(mean_disp), c_double(0.0), _vp(S["fit"].ptr))
            p1._last_gw_dev = (S["gw"], S["nm"])
            r = DeseqResult()
            r.disp_function_type, r.mean_disp = "mean", mean_disp
            sq, prior_var = p1._prior(Gn, S["fit"], r)
            p1._stage_map(d_y, mh, Gn, d_sf, prior_var, sq, S)
            # objective of the Powell search
            d_cst, d_nll, d_scale = p1._dvec(Gn), p1._dvec(Gn), p1._dvec(N)
            ctx.call("dsq_dev_nll_const", _vp(d_y.ptr), p1.ldn, N, Gn, _vp(S["disp"].ptr), _vp(d_cst.ptr))
            old_sf = sf.copy()

            def objective(p):
                s = np.exp(p - np.mean(p))
                ctx.h2d(d_scale.ptr, np.ascontiguousarray(s / old_sf))
                ctx.call("dsq_dev_nll_scaled", _vp(d_y.ptr), _vp(d_mu.ptr), p1.ldn, N, Gn, _vp(S["disp"].ptr),
                         _vp(d_scale.ptr), _vp(d_cst.ptr), _vp(d_nll.ptr))
                nll = p1._all_genes_host(d_nll, Gn)
                return np.sum(nll[nll < np.quantile(nll, quant)])

            res = minimize(objective, np.log(old_sf), method="Powell")
            sf = np.exp(res.x - np.mean(res.x))
            if not res.success:
                print("A size factor fitting iteration failed.", file=sys.stderr)
                break
            if (i > 1) and np.sum((np.log(old_sf) - np.log(sf)) ** 2) < 1e-4:
                break
            elif i == niter - 1:
                print("Iterative size factor fitting did not converge.", file=sys.stderr)
        return sf
    finally:
        for _cap, ptr in p1._pool_free + p1._pool_used:
            ctx.free(ptr)
        p1._pool_free, p1._pool_used = [], []
        p1.__dict__.clear()  # the twin owns nothing else: keep its __del__ from freeing the shared buffers
