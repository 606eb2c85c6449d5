// dsq_bfgs.h — scipy.optimize.minimize(method="BFGS") restated (scipy 1.15.3: _optimize.py _minimize_bfgs,
// _line_search_wolfe12; _linesearch.py line_search_wolfe1 / scalar_search_wolfe1 (MINPACK-2 dcsrch through
// _dcsrch.py), line_search_wolfe2 / scalar_search_wolfe2 / _zoom / _cubicmin / _quadmin).
//
// The reference offers optimizer="BFGS" for the dispersion fit (utils.py:546-554, unbounded in log alpha) and for the
// rescue of a diverged IRLS fit (utils.py:389-399, unbounded coefficients).  As with L-BFGS-B, what it returns is
// whatever iterate scipy stops at under its default tolerances (gtol = 1e-5 on the max-norm of the gradient, Wolfe
// line search with c1 = 1e-4, c2 = 0.9), so the algorithm is restated step by step rather than replaced by "a"
// quasi-Newton method.  Written for wave-uniform execution: every lane runs the same scalar code on the same values,
// the objective `fg(x, f, g)` is the only wave-parallel part.  Not on any hot path (no caller in dds.py / ds.py selects
// this optimiser).
#pragma once
#include "dsq_lbfgsb1d.h"

namespace dsq {

struct BfgsResult {
    bool success;
    int nit, nfev, status;  // status: scipy's warnflag (0 ok, 1 maxiter, 2 precision loss, 3 NaN)
};

// MINPACK-2 dcsrch as scipy's DCSRCH class drives it for line_search_wolfe1: ftol = c1, gtol = c2, xtol = 1e-14,
// stpmin = 1e-100, stpmax = 1e100 (the L-BFGS-B instance above has stpmin = 0 and its own tolerances)
struct DcsrchWolfe : Dcsrch {
    double ftol, gtol, xtol, stpmin;
    DSQ_HD Task start_w(double f, double g, double stp, double c1, double c2, double xtol_, double amin, double amax) {
        ftol = c1; gtol = c2; xtol = xtol_; stpmin = amin; stpmax = amax;
        if (stp < stpmin || stp > stpmax || g >= 0.0) return ERR;
        brackt = false;
        stage = 1;
        finit = f;
        ginit = g;
        gtest = ftol * ginit;
        width = stpmax - stpmin;
        width1 = width / 0.5;
        stx = 0.0; fx = finit; gx = ginit;
        sty = 0.0; fy = finit; gy = ginit;
        stmin = 0.0;
        stmax = stp + 4.0 * stp;
        return FG;
    }
    DSQ_HD Task step_w(double f, double g, double& stp) {
        const double ftest = finit + stp * gtest;
        if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
        Task task = FG;
        if (brackt && (stp <= stmin || stp >= stmax)) task = WARN;
        if (brackt && stmax - stmin <= xtol * stmax) task = WARN;
        if (stp == stpmax && f <= ftest && g <= gtest) task = WARN;
        if (stp == stpmin && (f > ftest || g >= gtest)) task = WARN;
        if (f <= ftest && fabs(g) <= gtol * (-ginit)) task = CONV;
        if (task != FG) return task;
        if (stage == 1 && f <= fx && f > ftest) {
            const double fm = f - stp * gtest;
            double fxm = fx - stx * gtest, fym = fy - sty * gtest;
            const double gm = g - gtest;
            double gxm = gx - gtest, gym = gy - gtest;
            dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, stmin, stmax);
            fx = fxm + stx * gtest;
            fy = fym + sty * gtest;
            gx = gxm + gtest;
            gy = gym + gtest;
        } else {
            dcstep(stx, fx, gx, sty, fy, gy, stp, f, g, stmin, stmax);
        }
        if (brackt) {
            if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
            width1 = width;
            width = fabs(sty - stx);
        }
        if (brackt) {
            stmin = dmin(stx, sty);
            stmax = dmax(stx, sty);
        } else {
            stmin = stp + 1.1 * (stp - stx);
            stmax = stp + 4.0 * (stp - stx);
        }
        stp = dmin(dmax(stp, stpmin), stpmax);  // np.clip
        if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax)) stp = stx;
        return FG;
    }
};

namespace detail {
DSQ_HD bool finite_d(double v) { return v == v && fabs(v) <= 1.79769313486231570815e308; }

// _cubicmin / _quadmin (None <-> ok = false): numpy raises on division by zero / overflow / invalid, which the
// reference code turns into None; the finiteness test of the result covers the same cases
DSQ_HD bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double& xmin) {
    const double C = fpa;
    const double db = b - a, dc = c - a;
    const double denom = (db * dc) * (db * dc) * (db - dc);
    if (denom == 0.0 || !finite_d(denom)) return false;
    const double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
    double A = (dc * dc) * v0 + (-(db * db)) * v1;
    double B = (-(dc * dc * dc)) * v0 + (db * db * db) * v1;
    if (!finite_d(A) || !finite_d(B)) return false;
    A /= denom;
    B /= denom;
    const double radical = B * B - 3.0 * A * C;
    if (!(radical >= 0.0) || !finite_d(radical) || A == 0.0) return false;  // sqrt of a negative number / x / 0 raise
    xmin = a + (-B + sqrt(radical)) / (3.0 * A);
    return finite_d(xmin);
}
DSQ_HD bool quadmin(double a, double fa, double fpa, double b, double fb, double& xmin) {
    const double D = fa, C = fpa;
    const double db = b - a * 1.0;
    if (db * db == 0.0) return false;
    const double B = (fb - D - C * db) / (db * db);
    if (B == 0.0 || !finite_d(B)) return false;
    xmin = a - C / (2.0 * B);
    return finite_d(xmin);
}
}  // namespace detail

// Work arrays of an n-dimensional run (n <= NMAX): LDS or stack, the caller decides
template <int NMAX>
struct BfgsWork {
    double xk[NMAX], gfk[NMAX], pk[NMAX], xt[NMAX], gt[NMAX], gnew[NMAX], sk[NMAX], yk[NMAX];
    double H[NMAX * NMAX], T[NMAX * NMAX];
};

// _line_search_wolfe12 (scipy/optimize/_optimize.py): line_search_wolfe1 (MINPACK-2 dcsrch, at most 100 evaluations) and, if
// that finds no step, line_search_wolfe2 (bracketing + zoom, 10 iterations each).  eval_at(s, phi, derphi) evaluates the
// objective and its directional derivative at xk + s pk and leaves the gradient in W.gt; the gradient belonging to the
// accepted step - when the search evaluated one there - is left in W.gnew (have_grad).  have_old = false is scipy's
// old_old_fval = None (the first Newton-CG iteration); amin1 / amax1 are line_search_wolfe1's step bounds (BFGS passes
// 1e-100 / 1e100, Newton-CG leaves the defaults 1e-8 / 50), amax2 > 0 line_search_wolfe2's cap (0: None).
// Returns false for scipy's _LineSearchError.
template <int NMAX, class Eval>
DSQ_HD bool line_search_wolfe12(Eval&& eval_at, int n, BfgsWork<NMAX>& W, double phi0, double derphi0, bool have_old,
                                double old_old_fval, double amin1, double amax1, double amax2, double& alpha_k,
                                double& fval_new, bool& have_grad) {
    constexpr double c1 = 1e-4, c2 = 0.9;
    bool have_step = false;
    alpha_k = 0.0; fval_new = 0.0; have_grad = false;
    // ---- line_search_wolfe1
    {
        double alpha1 = 1.0;
        if (have_old && derphi0 != 0.0) {
            const double t = 1.01 * 2.0 * (phi0 - old_old_fval) / derphi0;
            alpha1 = (t < 1.0) ? t : 1.0;  // Python's min(1.0, t): 1.0 for a NaN
            if (alpha1 < 0.0) alpha1 = 1.0;
        }
        DcsrchWolfe ls;
        double stp = alpha1, phi1 = phi0, derphi1 = derphi0;
        Dcsrch::Task task = ls.start_w(phi1, derphi1, stp, c1, c2, 1e-14, amin1, amax1);
        bool ok = false;
        if (task == Dcsrch::FG) {
            bool exhausted = true;
            for (int it = 0; it < 100; ++it) {  // scipy: the START call is iteration 0
                if (it > 0) {
                    task = ls.step_w(phi1, derphi1, stp);
                }
                if (!detail::finite_d(stp)) { task = Dcsrch::WARN; exhausted = false; break; }
                if (task == Dcsrch::FG) {
                    eval_at(stp, phi1, derphi1);
                } else {
                    exhausted = false;
                    break;
                }
            }
            ok = !exhausted && task == Dcsrch::CONV;
        }
        if (ok) {
            alpha_k = stp; fval_new = phi1; have_step = true; have_grad = true;  // gt = gradient at the last derphi
            for (int i = 0; i < n; ++i) W.gnew[i] = W.gt[i];
        }
    }
    // ---- line_search_wolfe2 (c1, c2, amax = 1e100, maxiter = 10) when the first search found nothing
    if (!have_step) {
        const bool has_amax = amax2 > 0.0;  // (line_search_wolfe2's amax = None: no cap)
        const double amax = amax2;
        double alpha0 = 0.0, alpha1 = 1.0;
        if (have_old && derphi0 != 0.0) {
            const double t = 1.01 * 2.0 * (phi0 - old_old_fval) / derphi0;
            alpha1 = (t < 1.0) ? t : 1.0;  // min(1.0, t) with Python's NaN behaviour (returns 1.0)
        }
        if (alpha1 < 0.0) alpha1 = 1.0;
        if (has_amax) alpha1 = dmin(alpha1, amax);
        double phi_a1, dtmp;
        eval_at(alpha1, phi_a1, dtmp);
        double derphi_at_a1 = dtmp;  // scipy evaluates derphi(alpha1) lazily: same value
        for (int i = 0; i < n; ++i) W.T[i] = W.gt[i];  // gradient belonging to alpha1 (kept in T[0..n))
        double phi_a0 = phi0, derphi_a0 = derphi0;
        bool found = false, failed = false, star_has_grad = false;
        double a_star = 0.0, phi_star = 0.0;
        // zoom as a local routine
        auto zoom = [&](double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo) {
            int i = 0;
            const double delta1 = 0.2, delta2 = 0.1;
            double phi_rec = phi0, a_rec = 0.0, a_j = 0.0;
            while (true) {
                const double dalpha = a_hi - a_lo;
                double a, b;
                if (dalpha < 0) { a = a_hi; b = a_lo; } else { a = a_lo; b = a_hi; }
                bool have = false;
                double cchk = 0.0;
                if (i > 0) {
                    cchk = delta1 * dalpha;
                    have = detail::cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec, a_j);
                }
                if (i == 0 || !have || a_j > b - cchk || a_j < a + cchk) {
                    const double qchk = delta2 * dalpha;
                    have = detail::quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_j);
                    if (!have || a_j > b - qchk || a_j < a + qchk) a_j = a_lo + 0.5 * dalpha;
                }
                double phi_aj, derphi_aj;
                eval_at(a_j, phi_aj, derphi_aj);
                if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= phi_lo) {
                    phi_rec = phi_hi; a_rec = a_hi; a_hi = a_j; phi_hi = phi_aj;
                } else {
                    if (fabs(derphi_aj) <= -c2 * derphi0) {
                        a_star = a_j; phi_star = phi_aj; found = true; star_has_grad = true;
                        for (int q = 0; q < n; ++q) W.gnew[q] = W.gt[q];
                        return;
                    }
                    if (derphi_aj * (a_hi - a_lo) >= 0) {
                        phi_rec = phi_hi; a_rec = a_hi; a_hi = a_lo; phi_hi = phi_lo;
                    } else {
                        phi_rec = phi_lo; a_rec = a_lo;
                    }
                    a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
                }
                i += 1;
                if (i > 10) { failed = true; return; }
            }
        };
        bool finished = false;
        for (int i = 0; i < 10 && !finished; ++i) {
            if (alpha1 == 0.0 || (has_amax && alpha0 > amax)) { failed = true; finished = true; break; }
            if (phi_a1 > phi0 + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && i > 0)) {
                zoom(alpha0, alpha1, phi_a0, phi_a1, derphi_a0);
                finished = true;
                break;
            }
            const double derphi_a1 = derphi_at_a1;
            if (fabs(derphi_a1) <= -c2 * derphi0) {
                a_star = alpha1; phi_star = phi_a1; found = true; star_has_grad = true; finished = true;
                for (int q = 0; q < n; ++q) W.gnew[q] = W.T[q];
                break;
            }
            if (derphi_a1 >= 0) {
                zoom(alpha1, alpha0, phi_a1, phi_a0, derphi_a1);
                finished = true;
                break;
            }
            const double alpha2 = has_amax ? dmin(2.0 * alpha1, amax) : 2.0 * alpha1;
            alpha0 = alpha1;
            alpha1 = alpha2;
            phi_a0 = phi_a1;
            derphi_a0 = derphi_a1;
            eval_at(alpha1, phi_a1, dtmp);
            derphi_at_a1 = dtmp;
            for (int q = 0; q < n; ++q) W.T[q] = W.gt[q];
        }
        if (!finished) {  // maxiter reached: the last alpha1 is returned without a gradient
            a_star = alpha1; phi_star = phi_a1; found = true; star_has_grad = false;
        }
        if (found && !failed) {
            alpha_k = a_star; fval_new = phi_star; have_step = true; have_grad = star_has_grad;
        }
    }
    return have_step;
}

// minimize(fun, x0, jac=True-like, method="BFGS") with scipy's defaults; x: start in, solution out.
// fg(const double* x, double& f, double* g) evaluates the objective and its gradient (wave-parallel inside).
template <int NMAX, class FG>
DSQ_HD BfgsResult bfgs_min(FG&& fg, int n, double* x, BfgsWork<NMAX>& W, double gtol = 1e-5) {
    BfgsResult res;
    res.success = false; res.nit = 0; res.nfev = 0; res.status = 0;
    const int maxiter = n * 200;
    for (int i = 0; i < n; ++i) W.xk[i] = x[i];
    double old_fval;
    fg(W.xk, old_fval, W.gfk);
    res.nfev += 1;
    for (int i = 0; i < n * n; ++i) W.H[i] = 0.0;
    for (int i = 0; i < n; ++i) W.H[i * n + i] = 1.0;
    auto norm2 = [&](const double* v) { double s = 0.0; for (int i = 0; i < n; ++i) s += v[i] * v[i]; return sqrt(s); };
    auto norminf = [&](const double* v) {  // numpy.amax(abs(v)): NaN propagates
        double m = 0.0;
        bool nan = false;
        for (int i = 0; i < n; ++i) { const double a = fabs(v[i]); if (a != a) nan = true; if (a > m) m = a; }
        return nan ? NAN : m;
    };
    double old_old_fval = old_fval + norm2(W.gfk) / 2.0;
    int k = 0, warnflag = 0;
    double gnorm = norminf(W.gfk);
    // phi(s), derphi(s) along pk from xk: one evaluation serves both (scipy calls f and fprime separately)
    double phi_f = 0.0;
    auto eval_at = [&](double s, double& ph, double& dph) {
        for (int i = 0; i < n; ++i) W.xt[i] = W.xk[i] + s * W.pk[i];
        fg(W.xt, ph, W.gt);
        res.nfev += 1;
        double d = 0.0;
        for (int i = 0; i < n; ++i) d += W.gt[i] * W.pk[i];
        dph = d;
        phi_f = ph;
    };
    while (gnorm > gtol && k < maxiter) {
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += W.H[i * n + j] * W.gfk[j];
            W.pk[i] = -s;
        }
        double derphi0 = 0.0;
        for (int i = 0; i < n; ++i) derphi0 += W.gfk[i] * W.pk[i];
        const double phi0 = old_fval;
        double alpha_k = 0.0, fval_new = 0.0;
        bool have_grad = false;
        const bool have_step = line_search_wolfe12<NMAX>(eval_at, n, W, phi0, derphi0, true, old_old_fval, 1e-100, 1e100,
                                                         1e100, alpha_k, fval_new, have_grad);
        if (!have_step) { warnflag = 2; break; }
        old_old_fval = phi0;
        old_fval = fval_new;
        for (int i = 0; i < n; ++i) { W.sk[i] = alpha_k * W.pk[i]; W.xk[i] = W.xk[i] + W.sk[i]; }
        if (!have_grad) {
            double ftmp;
            fg(W.xk, ftmp, W.gnew);
            res.nfev += 1;
        }
        for (int i = 0; i < n; ++i) { W.yk[i] = W.gnew[i] - W.gfk[i]; W.gfk[i] = W.gnew[i]; }
        k += 1;
        gnorm = norminf(W.gfk);
        if (gnorm <= gtol) break;
        if (alpha_k * norm2(W.pk) <= 0.0) break;  // xrtol = 0
        if (!detail::finite_d(old_fval)) { warnflag = 2; break; }
        double rhok_inv = 0.0;
        for (int i = 0; i < n; ++i) rhok_inv += W.yk[i] * W.sk[i];
        const double rhok = (rhok_inv == 0.0) ? 1000.0 : 1.0 / rhok_inv;
        // Hk = A1 Hk A2 + rhok sk sk^T,  A1 = I - rhok sk yk^T,  A2 = I - rhok yk sk^T   (two dense products, as numpy)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {  // T = Hk A2
                double s = 0.0;
                for (int q = 0; q < n; ++q) s += W.H[i * n + q] * ((q == j ? 1.0 : 0.0) - W.yk[q] * W.sk[j] * rhok);
                W.T[i * n + j] = s;
            }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {  // H = A1 T + rhok sk sk^T
                double s = 0.0;
                for (int q = 0; q < n; ++q) s += ((i == q ? 1.0 : 0.0) - W.sk[i] * W.yk[q] * rhok) * W.T[q * n + j];
                W.H[i * n + j] = s + rhok * W.sk[i] * W.sk[j];
            }
    }
    bool xnan = false;
    for (int i = 0; i < n; ++i) xnan = xnan || (W.xk[i] != W.xk[i]);
    if (warnflag == 2) {
    } else if (k >= maxiter) {
        warnflag = 1;
    } else if (gnorm != gnorm || old_fval != old_fval || xnan) {
        warnflag = 3;
    }
    for (int i = 0; i < n; ++i) x[i] = W.xk[i];
    res.success = warnflag == 0;
    res.nit = k;
    res.status = warnflag;
    (void)phi_f;
    return res;
}

// ------------------------------------------------------------------------------------------------------------------
// scipy.optimize.minimize(method="Newton-CG", jac=..., hess=callable) restated (scipy 1.15.3 _optimize.py
// _minimize_newtoncg): the Newton direction from conjugate gradients on the DENSE matrix hess(xk) p = -grad f(xk) started at
// 0 (at most 20 n steps, stopped at |r|_1 <= min(0.5, sqrt|g|_1) |g|_1, at non-positive curvature, or - "curvature keeps
// increasing" - given up with status 3), then _line_search_wolfe12 with line_search_wolfe1's default step bounds
// (amin = 1e-8, amax = 50) and old_old_fval = None in the first iteration; stops when the step's 1-norm is <= n xtol
// (xtol = 1e-5: the ftol / gtol options the reference passes are unknown to this method and ignored with a warning,
// utils.py:1112-1121).  The matrix need not be symmetric - the reference's is not (its prior curvature is broadcast over
// the rows, dsq_shrink.h) - it only ever enters through A p.  hess(const double* x, double* A): row-major n x n.
template <int NMAX>
struct NewtonCgWork : BfgsWork<NMAX> {
    double A[NMAX * NMAX], b[NMAX], ri[NMAX], ps[NMAX], Ap[NMAX];
};

template <int NMAX, class FG, class HS>
DSQ_HD BfgsResult newton_cg_min(FG&& fg, HS&& hess, int n, double* x, NewtonCgWork<NMAX>& W, double avextol = 1e-5) {
    BfgsResult res;
    res.success = false; res.nit = 0; res.nfev = 0; res.status = 0;
    const int maxiter = n * 200, cg_maxiter = 20 * n;
    const double xtol = (double)n * avextol;
    double update_l1norm = 1.79769313486231570815e308;
    for (int i = 0; i < n; ++i) W.xk[i] = x[i];
    double old_fval, old_old_fval = 0.0;
    bool have_old = false, have_g = true;
    fg(W.xk, old_fval, W.gfk);
    res.nfev += 1;
    int k = 0, warnflag = 0;
    double phi_f = 0.0;
    auto eval_at = [&](double s, double& ph, double& dph) {
        for (int i = 0; i < n; ++i) W.xt[i] = W.xk[i] + s * W.pk[i];
        fg(W.xt, ph, W.gt);
        res.nfev += 1;
        double d = 0.0;
        for (int i = 0; i < n; ++i) d += W.gt[i] * W.pk[i];
        dph = d;
        phi_f = ph;
    };
    while (update_l1norm > xtol) {
        if (k >= maxiter) { warnflag = 1; break; }
        if (!have_g) {  // (the line search returned a step without its gradient)
            double ftmp;
            fg(W.xk, ftmp, W.gfk);
            res.nfev += 1;
        }
        double maggrad = 0.0, dri0 = 0.0;
        for (int i = 0; i < n; ++i) {
            W.b[i] = -W.gfk[i];
            maggrad += fabs(W.b[i]);
            W.pk[i] = 0.0;        // xsupi
            W.ri[i] = W.gfk[i];   // -b
            W.ps[i] = W.b[i];     // -ri
            dri0 += W.ri[i] * W.ri[i];
        }
        const double sq = sqrt(maggrad);
        const double eta = (sq < 0.5) ? sq : 0.5;  // Python's min(0.5, s): 0.5 for a NaN
        const double termcond = eta * maggrad;
        hess(W.xk, W.A);
        bool cg_done = false;
        int ic = 0;
        for (int k2 = 0; k2 < cg_maxiter; ++k2) {
            double r1 = 0.0;
            for (int i = 0; i < n; ++i) r1 += fabs(W.ri[i]);
            if (r1 <= termcond) { cg_done = true; break; }
            double curv = 0.0;
            for (int i = 0; i < n; ++i) {
                double v = 0.0;
                for (int j = 0; j < n; ++j) v += W.A[i * n + j] * W.ps[j];
                W.Ap[i] = v;
            }
            for (int i = 0; i < n; ++i) curv += W.ps[i] * W.Ap[i];
            if (0.0 <= curv && curv <= 3.0 * 2.220446049250313e-16) { cg_done = true; break; }
            if (curv < 0.0) {
                if (ic == 0) {  // steepest descent
                    const double c = dri0 / (-curv);
                    for (int i = 0; i < n; ++i) W.pk[i] = c * W.b[i];
                }
                cg_done = true;
                break;
            }
            const double alphai = dri0 / curv;
            double dri1 = 0.0;
            for (int i = 0; i < n; ++i) {
                W.pk[i] += alphai * W.ps[i];
                W.ri[i] += alphai * W.Ap[i];
                dri1 += W.ri[i] * W.ri[i];
            }
            const double betai = dri1 / dri0;
            for (int i = 0; i < n; ++i) W.ps[i] = -W.ri[i] + betai * W.ps[i];
            ic += 1;
            dri0 = dri1;
        }
        if (!cg_done) { warnflag = 3; break; }  // "CG iterations didn't converge. The Hessian is not positive definite."
        double derphi0 = 0.0;
        for (int i = 0; i < n; ++i) derphi0 += W.gfk[i] * W.pk[i];
        const double phi0 = old_fval;
        double alpha_k = 0.0, fval_new = 0.0;
        bool have_grad = false;
        if (!line_search_wolfe12<NMAX>(eval_at, n, W, phi0, derphi0, have_old, old_old_fval, 1e-8, 50.0, 0.0, alpha_k,
                                       fval_new, have_grad)) {
            warnflag = 2;
            break;
        }
        old_old_fval = phi0;
        have_old = true;
        old_fval = fval_new;
        double l1 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double u = alpha_k * W.pk[i];
            W.xk[i] += u;
            l1 += fabs(u);
        }
        update_l1norm = l1;
        k += 1;
        have_g = have_grad;
        if (have_grad)
            for (int i = 0; i < n; ++i) W.gfk[i] = W.gnew[i];
    }
    if (warnflag == 0 && (old_fval != old_fval || update_l1norm != update_l1norm)) warnflag = 3;
    for (int i = 0; i < n; ++i) x[i] = W.xk[i];
    res.success = warnflag == 0;
    res.nit = k;
    res.status = warnflag;
    (void)phi_f;
    return res;
}

}  // namespace dsq
