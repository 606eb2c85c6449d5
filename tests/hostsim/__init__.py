"""ctypes front-end of the TEST-ONLY host build of the per-gene templates (see hostsim.cpp)."""
import ctypes as C

import numpy as np

from .build import build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def gene_major(counts):
    """N x G counts -> contiguous int32 [G][N]."""
    return np.ascontiguousarray(np.asarray(counts).T.astype(np.int32))


def design_pack(X):
    X = np.asarray(X, dtype=np.float64)
    Xt = np.ascontiguousarray(X.T)
    pinv = np.ascontiguousarray(np.linalg.pinv(X))
    full_rank = int(np.linalg.matrix_rank(X) == X.shape[1])
    return Xt, pinv, full_rank


def set_workspace_fill(byte):
    """What the routines' wave-private workspaces hold when they start (device: stale LDS): 0x00, 0xFF (NaNs), ..."""
    lib().hs_set_workspace_fill(C.c_int(int(byte)))


def lgamma_digamma(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    lg, dg = np.empty_like(x), np.empty_like(x)
    lib().hs_lgamma_digamma(_p(x, C.c_double), C.c_int(x.size), _p(lg, C.c_double), _p(dg, C.c_double))
    return lg, dg


def norm_sf(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    o = np.empty_like(x)
    lib().hs_norm_sf(_p(x, C.c_double), C.c_int(x.size), _p(o, C.c_double))
    return o


FG_CB = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double))


def lbfgsb1d(fg, x0, l, u):
    def cb(x, pf, pg):
        f, g = fg(x)
        pf[0], pg[0] = f, g

    x, f = C.c_double(), C.c_double()
    ok, nfev, nit, st = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib().hs_lbfgsb1d(FG_CB(cb), C.c_double(x0), C.c_double(l), C.c_double(u), C.byref(x), C.byref(f),
                      C.byref(ok), C.byref(nfev), C.byref(nit), C.byref(st))
    return x.value, f.value, bool(ok.value), nfev.value, nit.value, st.value


def alpha_mle(counts, X, mu, alpha_hat, min_disp, max_disp, prior_var=None, cr_reg=True, prior_reg=False,
              optimizer="L-BFGS-B"):
    y = gene_major(counts)
    G, N = y.shape
    m = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).T)
    Xt, _, _ = design_pack(X)
    ah = np.ascontiguousarray(alpha_hat, dtype=np.float64)
    out, conv, nfev = np.empty(G), np.empty(G, np.uint8), np.empty(G, np.int32)
    fn = lib().hs_alpha_mle if optimizer == "L-BFGS-B" else lib().hs_alpha_mle_bfgs
    rc = fn(_p(y, C.c_int32), _p(m, C.c_double), C.c_int(N), _p(Xt, C.c_double), C.c_int(N),
                            C.c_int(N), C.c_int(G), C.c_int(Xt.shape[0]), _p(ah, C.c_double),
                            C.c_double(min_disp), C.c_double(max_disp),
                            C.c_double(prior_var if prior_var is not None else 1.0), C.c_int(cr_reg),
                            C.c_int(prior_reg), _p(out, C.c_double), _p(conv, C.c_uint8), _p(nfev, C.c_int32))
    assert rc == 0
    return out, conv.astype(bool), nfev


def grid_alpha(counts, X, mu, min_disp, max_disp):
    y = gene_major(counts)
    G, N = y.shape
    m = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).T)
    Xt, _, _ = design_pack(X)
    out = np.empty(G)
    rc = lib().hs_grid_alpha(_p(y, C.c_int32), _p(m, C.c_double), C.c_int(N), _p(Xt, C.c_double), C.c_int(N),
                             C.c_int(N), C.c_int(G), C.c_int(Xt.shape[0]), C.c_double(min_disp),
                             C.c_double(max_disp), _p(out, C.c_double))
    assert rc == 0
    return out


def irls(counts, sf, X, disp, min_mu=0.5, beta_tol=1e-8, min_beta=-30.0, max_beta=30.0, maxiter=250,
         optimizer="L-BFGS-B"):
    y = gene_major(counts)
    G, N = y.shape
    Xt, pinv, fr = design_pack(X)
    P = Xt.shape[0]
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    d = np.ascontiguousarray(disp, dtype=np.float64)
    beta, mu, H = np.empty((G, P)), np.empty((G, N)), np.empty((G, N))
    conv, it, fb = np.empty(G, np.uint8), np.empty(G, np.int32), np.empty(G, np.uint8)
    fn = lib().hs_irls if optimizer == "L-BFGS-B" else lib().hs_irls_bfgs
    rc = fn(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(Xt, C.c_double),
                       _p(pinv, C.c_double), C.c_int(N), C.c_int(N), C.c_int(G), C.c_int(P), _p(d, C.c_double),
                       C.c_double(min_mu), C.c_double(beta_tol), C.c_double(min_beta), C.c_double(max_beta),
                       C.c_int(maxiter), C.c_int(fr), _p(beta, C.c_double), _p(mu, C.c_double),
                       _p(H, C.c_double), _p(conv, C.c_uint8), _p(it, C.c_int32), _p(fb, C.c_uint8))
    assert rc == 0
    return beta, mu.T, H.T, conv.astype(bool), it, fb.astype(bool)


def logmeans(counts):
    y = gene_major(counts)
    G, N = y.shape
    lm, nz = np.empty(G), np.empty(G, np.uint8)
    lib().hs_logmeans(_p(y, C.c_int32), C.c_int(N), C.c_int(N), C.c_int(G), _p(lm, C.c_double), _p(nz, C.c_uint8))
    return lm, nz.astype(bool)


def mom(counts, sf, X, min_disp, max_disp):
    y = gene_major(counts)
    G, N = y.shape
    Xt, pinv, _ = design_pack(X)
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    o = [np.empty(G) for _ in range(4)]
    rc = lib().hs_mom(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(Xt, C.c_double), _p(pinv, C.c_double),
                      C.c_int(N), C.c_int(N), C.c_int(G), C.c_int(Xt.shape[0]), C.c_double(min_disp),
                      C.c_double(max_disp), *[_p(a, C.c_double) for a in o])
    assert rc == 0
    return dict(normed_mean=o[0], rough=o[1], moments=o[2], mom=o[3])


def mom_lin_mu(counts, sf, X, min_disp, max_disp, min_mu):
    y = gene_major(counts)
    G, N = y.shape
    Xt, pinv, _ = design_pack(X)
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    nm, mom_, mu = np.empty(G), np.empty(G), np.empty((G, N))
    rc = lib().hs_mom_lin_mu(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(Xt, C.c_double),
                             _p(pinv, C.c_double), C.c_int(N), C.c_int(N), C.c_int(G), C.c_int(Xt.shape[0]),
                             C.c_double(min_disp), C.c_double(max_disp), C.c_double(min_mu), _p(nm, C.c_double),
                             _p(mom_, C.c_double), _p(mu, C.c_double))
    assert rc == 0
    return dict(normed_mean=nm, mom=mom_, mu=mu.T)


def lin_mu(counts, sf, X, min_mu):
    y = gene_major(counts)
    G, N = y.shape
    Xt, pinv, _ = design_pack(X)
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    mu = np.empty((G, N))
    rc = lib().hs_lin_mu(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(Xt, C.c_double), _p(pinv, C.c_double),
                         C.c_int(N), C.c_int(N), C.c_int(G), C.c_int(Xt.shape[0]), C.c_double(min_mu),
                         _p(mu, C.c_double))
    assert rc == 0
    return mu.T


ALT = {None: 0, "greaterAbs": 1, "lessAbs": 2, "greater": 3, "less": 4}


def wald(X, disp, beta, sf, ridge, contrast, lfc_null, alt, mu=None):
    Xt, _, _ = design_pack(X)
    P, N = Xt.shape
    G = len(disp)
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    m = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).T) if mu is not None else None
    d = np.ascontiguousarray(disp, dtype=np.float64)
    b = np.ascontiguousarray(beta, dtype=np.float64)
    r = np.ascontiguousarray(ridge, dtype=np.float64)
    c = np.ascontiguousarray(contrast, dtype=np.float64)
    p, s, se = np.empty(G), np.empty(G), np.empty(G)
    rc = lib().hs_wald(_p(m, C.c_double), C.c_int(N), _p(sf, C.c_double), _p(Xt, C.c_double), C.c_int(N),
                       C.c_int(N), C.c_int(G), C.c_int(P), _p(d, C.c_double), _p(b, C.c_double), _p(r, C.c_double),
                       _p(c, C.c_double), C.c_double(lfc_null), C.c_int(ALT[alt]), _p(p, C.c_double),
                       _p(s, C.c_double), _p(se, C.c_double))
    assert rc == 0
    return p, s, se


def cell_plan(X, min_replicates=7):
    """Group samples by identical design rows (cells with >= 3 replicates are listed)."""
    X = np.asarray(X, dtype=np.float64)
    _, inv, cnt = np.unique(X, axis=0, return_inverse=True, return_counts=True)
    inv = np.asarray(inv).reshape(-1)
    size = cnt[inv]
    flags = ((size >= 3).astype(np.uint8)) | ((size >= min_replicates).astype(np.uint8) << 1)
    cells = [np.nonzero(inv == c)[0] for c in range(len(cnt)) if cnt[c] >= 3]
    whole = int(len(cells) == 0)
    if whole:
        offsets = np.array([0, len(inv)], np.int32)
        index = np.arange(len(inv), dtype=np.int32)
        ncell = 0
    else:
        offsets = np.concatenate([[0], np.cumsum([len(c) for c in cells])]).astype(np.int32)
        index = np.concatenate(cells).astype(np.int32)
        ncell = len(cells)
    return offsets, index, ncell, whole, np.ascontiguousarray(flags)


def cooks(counts, sf, X, mu, H, cutoff, min_replicates=7):
    y = gene_major(counts)
    G, N = y.shape
    P = np.asarray(X).shape[1]
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    m = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).T)
    h = np.ascontiguousarray(np.asarray(H, dtype=np.float64).T)
    off, idx, ncell, whole, flags = cell_plan(X, min_replicates)
    ck, rd = np.empty((G, N)), np.empty(G)
    fl = [np.empty(G, np.uint8) for _ in range(4)]
    rc = lib().hs_cooks(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(m, C.c_double), _p(h, C.c_double),
                        _p(off, C.c_int32), _p(idx, C.c_int32), C.c_int(ncell), C.c_int(whole), _p(flags, C.c_uint8),
                        C.c_int(N), C.c_int(G), C.c_int(P), C.c_double(cutoff), _p(ck, C.c_double),
                        _p(rd, C.c_double), *[_p(a, C.c_uint8) for a in fl])
    assert rc == 0
    return ck.T, rd, [a.astype(bool) for a in fl]


def robust_disp_seg(counts, sf, X, seg_len):
    """robust_disp_gene with the batched small-cell path (seg_len = power of two >= the largest cell, 0: one cell at a time)"""
    y = gene_major(counts)
    G, N = y.shape
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    off, idx, ncell, whole, flags = cell_plan(X, 7)
    out = np.empty(G)
    rc = lib().hs_robust_disp_seg(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(off, C.c_int32),
                                  _p(idx, C.c_int32), C.c_int(ncell), C.c_int(whole), C.c_int(N), C.c_int(G),
                                  C.c_int(seg_len), _p(out, C.c_double))
    assert rc == 0
    return out


def robust_disp_lean(counts, sf, X):
    """robust_disp_gene_lean (no buffer of a cell's values: bucket pass or radix selection over an accessor, any cell size):
    (robust dispersions [G], failed [G])"""
    y = gene_major(counts)
    G, N = y.shape
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    off, idx, ncell, whole, flags = cell_plan(X, 7)
    out, failed = np.empty(G), np.empty(G, np.uint8)
    rc = lib().hs_robust_disp_lean(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(off, C.c_int32),
                                   _p(idx, C.c_int32), C.c_int(ncell), C.c_int(whole), C.c_int(N), C.c_int(G),
                                   _p(out, C.c_double), _p(failed, C.c_uint8))
    assert rc == 0
    return out, failed.astype(bool)


def trimmed_base_mean_lean(counts, sf, trim=0.2):
    y = gene_major(counts)
    G, N = y.shape
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    out = np.empty(G)
    lib().hs_trimmed_base_mean_lean(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), C.c_int(N), C.c_int(G),
                                    C.c_double(trim), _p(out, C.c_double))
    return out


def select_rank_sum(buf, j_lo, j_hi):
    """sum of the ranks j_lo .. j_hi (inclusive) among the entries >= 0 of buf, by the accessor-based radix selection"""
    b = np.ascontiguousarray(buf, dtype=np.float64)
    f = lib().hs_select_rank_sum
    f.restype = C.c_double
    return float(f(_p(b, C.c_double), C.c_int(len(b)), C.c_int(int(j_lo)), C.c_int(int(j_hi))))


def trimmed_base_mean(counts, sf, trim=0.2):
    y = gene_major(counts)
    G, N = y.shape
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    out = np.empty(G)
    lib().hs_trimmed_base_mean(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), C.c_int(N), C.c_int(G),
                               C.c_double(trim), _p(out, C.c_double))
    return out


def alpha_eval(y, mu, X, la, la_hat=0.0, prior_var=1.0, cr_reg=True, prior_reg=False):
    yv = np.ascontiguousarray(y, dtype=np.int32)
    m = np.ascontiguousarray(mu, dtype=np.float64)
    Xt, _, _ = design_pack(X)
    f, g = C.c_double(), C.c_double()
    rc = lib().hs_alpha_eval(_p(yv, C.c_int32), _p(m, C.c_double), _p(Xt, C.c_double), C.c_int(len(yv)),
                             C.c_int(len(yv)), C.c_int(Xt.shape[0]), C.c_double(la), C.c_double(la_hat),
                             C.c_double(prior_var), C.c_int(cr_reg), C.c_int(prior_reg), C.byref(f), C.byref(g))
    assert rc == 0
    return f.value, g.value


FGN_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


def lbfgsb_dense(fg, x0, bounds):
    return lbfgsb_nd(fg, x0, bounds, entry="hs_lbfgsb_dense")


def lbfgsb_nd(fg, x0, bounds, entry="hs_lbfgsb_nd"):
    """bounds: list of (lo, hi) with None/inf for unbounded (scipy convention)."""
    n = len(x0)
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    l, u, nbd = np.zeros(n), np.zeros(n), np.zeros(n, np.int32)
    for i, (lo, hi) in enumerate(bounds):
        has_l = lo is not None and np.isfinite(lo)
        has_u = hi is not None and np.isfinite(hi)
        if has_l:
            l[i] = lo
        if has_u:
            u[i] = hi
        nbd[i] = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}[(has_l, has_u)]

    def cb(px, pf, pg):
        xx = np.array([px[i] for i in range(n)])
        f, g = fg(xx)
        pf[0] = f
        for i in range(n):
            pg[i] = g[i]

    f = C.c_double()
    ok, nfev, nit, st = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = getattr(lib(), entry)(FGN_CB(cb), C.c_int(n), _p(x, C.c_double), _p(l, C.c_double), _p(u, C.c_double),
                            _p(nbd, C.c_int32), C.byref(f), C.byref(ok), C.byref(nfev), C.byref(nit), C.byref(st))
    assert rc == 0
    return x, f.value, bool(ok.value), nfev.value, nit.value, st.value


def bfgs(fg, x0):
    """scipy.optimize.minimize(method="BFGS") restatement (dsq_bfgs.h) on a Python objective fg(x) -> (f, g)."""
    n = len(x0)
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()

    def cb(px, pf, pg):
        xx = np.array([px[i] for i in range(n)])
        f, g = fg(xx)
        pf[0] = f
        for i in range(n):
            pg[i] = g[i]

    ok, nfev, nit, st = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().hs_bfgs(FGN_CB(cb), C.c_int(n), _p(x, C.c_double), C.byref(ok), C.byref(nfev), C.byref(nit),
                       C.byref(st))
    assert rc == 0
    return x, bool(ok.value), nfev.value, nit.value, st.value


def trend_fit(disp, means, min_disp, max_disp):
    d = np.ascontiguousarray(disp, dtype=np.float64)
    m = np.ascontiguousarray(means, dtype=np.float64)
    c = np.empty(2)
    ok, no = C.c_int(), C.c_int()
    lib().hs_trend_fit(_p(d, C.c_double), _p(m, C.c_double), C.c_int(len(d)), C.c_double(min_disp),
                       C.c_double(max_disp), _p(c, C.c_double), C.byref(ok), C.byref(no))
    return c, bool(ok.value), no.value


def flog(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    a, b, c = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    lib().hs_flog(_p(x, C.c_double), C.c_int(x.size), _p(a, C.c_double), _p(b, C.c_double), _p(c, C.c_double))
    return a, b, c


def fexp(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().hs_fexp(_p(x, C.c_double), C.c_int(x.size), _p(out, C.c_double))
    return out


def trimmed_sum(buf, nt):
    """sum(sorted(buf)[nt : len - nt]) through the device's selection routine (host instantiation)."""
    b = np.ascontiguousarray(buf, dtype=np.float64)
    f = lib().hs_trimmed_sum
    f.restype = C.c_double
    return float(f(_p(b, C.c_double), C.c_int(len(b)), C.c_int(int(nt))))


def shrink(counts, X, size, offset, prior_no_shrink_scale, prior_scale, shrink_index, optimizer="L-BFGS-B"):
    """apeGLM MAP fit (nbinomGLM) through the device templates: (beta G x p, inv_hessian G x p x p, converged)."""
    y = gene_major(counts)
    G, N = y.shape
    Xt, _, _ = design_pack(X)
    P = Xt.shape[0]
    sz = np.ascontiguousarray(size, dtype=np.float64)
    off = np.ascontiguousarray(offset, dtype=np.float64)
    beta, invh, conv = np.empty((G, P)), np.empty((G, P, P)), np.empty(G, np.uint8)
    rc = lib().hs_shrink(_p(y, C.c_int32), C.c_int(N), _p(off, C.c_double), _p(Xt, C.c_double), C.c_int(N),
                         C.c_int(N), C.c_int(G), C.c_int(P), _p(sz, C.c_double), C.c_double(prior_no_shrink_scale),
                         C.c_double(prior_scale), C.c_int(shrink_index), _p(beta, C.c_double),
                         _p(invh, C.c_double), _p(conv, C.c_uint8),
                         C.c_int({"L-BFGS-B": 0, "BFGS": 1, "Newton-CG": 2}[optimizer]))
    assert rc == 0
    return beta, invh, conv.astype(bool)


def cell_design(X):
    """(cell_of int32 [N], Xc [C][P], XX [C][T], C): the design's distinct rows (pydeseq2_amd/_design.py)."""
    X = np.asarray(X, dtype=np.float64)
    rows, inv = np.unique(X, axis=0, return_inverse=True)
    ii, jj = np.tril_indices(X.shape[1])
    Xc = np.ascontiguousarray(rows)
    return (np.ascontiguousarray(np.asarray(inv).reshape(-1).astype(np.int32)), Xc,
            np.ascontiguousarray(Xc[:, ii] * Xc[:, jj]), len(rows))


def alpha_mle_cell(counts, X, mu, alpha_hat, min_disp, max_disp, prior_var=None, cr_reg=True, prior_reg=False,
                   entry="hs_alpha_mle_cell", cells=True):
    y = gene_major(counts)
    G, N = y.shape
    m = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).T)
    Xt, _, _ = design_pack(X)
    cof, Xc, XX, Cn = cell_design(X)
    ah = np.ascontiguousarray(alpha_hat, dtype=np.float64)
    out, conv = np.empty(G), np.empty(G, np.uint8)
    rc = getattr(lib(), entry)(_p(y, C.c_int32), _p(m, C.c_double), C.c_int(N), _p(Xt, C.c_double), C.c_int(N),
                               C.c_int(N), C.c_int(G), C.c_int(Xt.shape[0]), _p(ah, C.c_double),
                               C.c_double(min_disp), C.c_double(max_disp),
                               C.c_double(prior_var if prior_var is not None else 1.0), C.c_int(cr_reg),
                               C.c_int(prior_reg), _p(cof, C.c_int32), _p(Xc, C.c_double), _p(XX, C.c_double),
                               C.c_int(Cn if cells else 0), _p(out, C.c_double), _p(conv, C.c_uint8))
    assert rc == 0
    return out, conv.astype(bool)


def alpha_mle_wide(counts, X, mu, alpha_hat, min_disp, max_disp, cells=False, **kw):
    """Run-time-P path (dsq_wide.h): Gram matrices by the (host stand-in of the) matrix-core accumulation."""
    return alpha_mle_cell(counts, X, mu, alpha_hat, min_disp, max_disp, entry="hs_alpha_mle_wide", cells=cells, **kw)


def mom_wide(counts, sf, X, min_disp, max_disp, min_mu=0.5):
    y = gene_major(counts)
    G, N = y.shape
    Xt, pinv, _ = design_pack(X)
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    out = [np.empty(G) for _ in range(4)]
    mu = np.empty((G, N))
    rc = lib().hs_mom_wide(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(Xt, C.c_double), _p(pinv, C.c_double),
                           C.c_int(N), C.c_int(N), C.c_int(G), C.c_int(Xt.shape[0]), C.c_double(min_disp),
                           C.c_double(max_disp), C.c_double(min_mu), *[_p(a, C.c_double) for a in out],
                           _p(mu, C.c_double))
    assert rc == 0
    return dict(normed_mean=out[0], rough=out[1], moments=out[2], mom=out[3], lin_mu=mu.T)


def lfc_fit(counts, sf, X, disp, cells=False, robust_disp=None, cutoff=0.0, contrast=None, lfc_null=0.0, alt=0,
            min_replicates=7, want_layers=True, entry="hs_lfc_fit"):
    """IRLS + fused epilogue (Cook's bookkeeping if robust_disp is given, Wald if contrast is given)."""
    y = gene_major(counts)
    G, N = y.shape
    Xt, pinv, fr = design_pack(X)
    P = Xt.shape[0]
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    d = np.ascontiguousarray(disp, dtype=np.float64)
    cof, Xc, XX, Cn = cell_design(X)
    if not cells:
        Cn = 0
    _, _, _, _, flags = cell_plan(X, min_replicates)
    beta, conv = np.empty((G, P)), np.empty(G, np.uint8)
    mu = np.empty((G, N)) if want_layers else None
    H = np.empty((G, N)) if want_layers else None
    ck = np.empty((G, N)) if robust_disp is not None else None
    fl = [np.empty(G, np.uint8) for _ in range(4)]
    pv, st, se = np.empty(G), np.empty(G), np.empty(G)
    rd = np.ascontiguousarray(robust_disp, dtype=np.float64) if robust_disp is not None else None
    ridge = np.ascontiguousarray(np.diag(np.repeat(1e-6, P))) if contrast is not None else None
    cvec = np.ascontiguousarray(contrast, dtype=np.float64) if contrast is not None else None
    rc = getattr(lib(), entry)(_p(y, C.c_int32), C.c_int(N), _p(sf, C.c_double), _p(Xt, C.c_double), _p(pinv, C.c_double),
                          C.c_int(N), C.c_int(N), C.c_int(G), C.c_int(P), _p(d, C.c_double), C.c_double(0.5),
                          C.c_double(1e-8), C.c_int(fr), _p(cof, C.c_int32), _p(Xc, C.c_double), _p(XX, C.c_double),
                          C.c_int(Cn), _p(rd, C.c_double) if rd is not None else None,
                          _p(flags, C.c_uint8) if rd is not None else None, C.c_double(cutoff),
                          _p(ck, C.c_double) if ck is not None else None, *[_p(a, C.c_uint8) for a in fl],
                          _p(ridge, C.c_double) if ridge is not None else None,
                          _p(cvec, C.c_double) if cvec is not None else None, C.c_double(lfc_null), C.c_int(alt),
                          _p(beta, C.c_double), _p(mu, C.c_double) if mu is not None else None,
                          _p(H, C.c_double) if H is not None else None, _p(conv, C.c_uint8), _p(pv, C.c_double),
                          _p(st, C.c_double), _p(se, C.c_double))
    assert rc == 0
    out = dict(beta=beta, conv=conv.astype(bool), p=pv, stat=st, se=se, flags=[a.astype(bool) for a in fl])
    if want_layers:
        out["mu"], out["H"] = mu.T, H.T
    if ck is not None:
        out["cooks"] = ck.T
    return out
