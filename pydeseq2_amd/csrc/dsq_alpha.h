// dsq_alpha.h — per-gene dispersion fit (genewise MLE and MAP), one gene per wave.
//
// Replaces pydeseq2/utils.py:441-564 (fit_alpha_mle: loss :509-520, gradient :522-544,
// L-BFGS-B :546-554, grid fallback :556-564), utils.py:163-270 (nb_nll, dnb_nll) and
// pydeseq2/grid_search.py:54-142 (grid_fit_alpha) + :7-51 (vec_nb_nll).
//
// Data layout: counts y[n] (int32) and mu[n] (fp64) are rows of gene-major matrices, the
// design is passed transposed Xt[j*ldx + n] so that lane l reads sample l, l+W, ...
// with unit stride.  X^T W X and X^T dW X (p(p+1)/2 entries each) are accumulated
// per lane in registers and all-reduced across the wave; the p x p Cholesky / log-det /
// inverse runs redundantly in every lane (wave-uniform, no LDS round trip needed at
// these sizes).
#pragma once
#include "dsq_lbfgsb1d.h"
#include "dsq_linalg.h"
#include "dsq_wave.h"

namespace dsq {

struct AlphaArgs {
    const int32_t* y;   // [N]
    const double* mu;   // [N]
    const double* Xt;   // [P][ldx]
    int ldx;
    int N;
    double cst;         // sum lgamma(y+1) - sum y log(mu)   (alpha independent)
    double la_hat;      // log(alpha_hat)
    double prior_var;
    bool cr_reg, prior_reg;
};

// loss (and gradient) of the Cox-Reid / prior regularised NB negative log-likelihood at
// log_alpha.  GRAD = false is used by the grid search.
template <class Wv, int P, bool GRAD>
DSQ_HD void alpha_eval(const AlphaArgs& A, double la, bool cr_reg, bool prior_reg, double& f,
                       double& g) {
    constexpr int T = Tri<P>::N;
    const double alpha = exp(la);
    const double a = 1.0 / alpha;
    double lga, dga;
    lgamma_digamma<GRAD>(a, lga, dga);
    KSum accf;
    double accg = 0.0;
    double M[T], dM[T];
#pragma unroll
    for (int k = 0; k < T; ++k) { M[k] = 0.0; dM[k] = 0.0; }
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        const double m = A.mu[n];
        double lgy, dgy;
        lgamma_digamma<GRAD>(yv + a, lgy, dgy);
        const double lam = log(a + m);
        accf.add((lga - lgy) + (yv + a) * lam);
        if (GRAD) accg += dga - dgy + log(1.0 + m * alpha) + (yv - m) / (m + a);
        if (cr_reg) {
            const double w = m / (1.0 + m * alpha);
            const double dw = -(w * w);
            double x[P];
#pragma unroll
            for (int j = 0; j < P; ++j) x[j] = A.Xt[j * A.ldx + n];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const double xw = x[i] * w, xdw = x[i] * dw;
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    M[tri(i, j)] += xw * x[j];
                    if (GRAD) dM[tri(i, j)] += xdw * x[j];
                }
            }
        }
    }
    const double sumf = Wv::sum_comp(accf);
    if (GRAD) accg = Wv::sum(accg);
    f = A.N * a * log(alpha) + (sumf + A.cst);
    g = 0.0;
    if (GRAD) g = alpha * (-(a * a * accg));
    if (cr_reg) {
        Wv::template sum_n<T>(M);
        if (GRAD) Wv::template sum_n<T>(dM);
        chol<P>(M);
        f += 0.5 * chol_logdet<P>(M);
        if (GRAD) {
            double inv[T];
            chol_inverse<P>(M, inv);
            g += 0.5 * sym_frob<P>(inv, dM) * alpha;
        }
    }
    if (prior_reg) {
        const double dl = la - A.la_hat;
        f += dl * dl / (2.0 * A.prior_var);
        if (GRAD) g += dl / A.prior_var;
    }
}

// numpy.linspace(lo, hi, num)[i]
DSQ_HD double linspace_at(double lo, double hi, int num, int i) {
    if (i == num - 1) return hi;
    const double step = (hi - lo) / (double)(num - 1);
    return (double)i * step + lo;
}

// grid_fit_alpha (grid_search.py:54-142) as the reference calls it from fit_alpha_mle:
// Cox-Reid term on, prior OFF (utils.py:561 passes six positional arguments only).
template <class Wv, int P>
DSQ_HD double grid_fit_alpha(const AlphaArgs& A, double lo, double hi, int grid_length = 100) {
    double best = 0.0, g_unused;
    int kbest = 0;
    bool best_nan = false;
    for (int i = 0; i < grid_length; ++i) {
        double f;
        alpha_eval<Wv, P, false>(A, linspace_at(lo, hi, grid_length, i), true, false, f, g_unused);
        const bool isn = (f != f);
        if (i == 0 || (!best_nan && (isn || f < best))) { best = f; kbest = i; best_nan = isn; }
    }
    const double delta = linspace_at(lo, hi, grid_length, 1) - linspace_at(lo, hi, grid_length, 0);
    const double c = linspace_at(lo, hi, grid_length, kbest);
    const double flo = c - delta, fhi = c + delta;
    best_nan = false;
    for (int i = 0; i < grid_length; ++i) {
        double f;
        alpha_eval<Wv, P, false>(A, linspace_at(flo, fhi, grid_length, i), true, false, f, g_unused);
        const bool isn = (f != f);
        if (i == 0 || (!best_nan && (isn || f < best))) { best = f; kbest = i; best_nan = isn; }
    }
    return linspace_at(flo, fhi, grid_length, kbest);
}

struct AlphaOut {
    double alpha;
    int converged;  // scipy's res.success
    int nfev, nit, status;
};

// one gene: L-BFGS-B in log(alpha) from log(alpha_hat), grid search if it did not converge.
template <class Wv, int P>
DSQ_HD AlphaOut fit_alpha_gene(const int32_t* y, const double* mu, const double* Xt, int ldx, int N,
                               double alpha_hat, double min_disp, double max_disp,
                               double prior_var, bool cr_reg, bool prior_reg) {
    AlphaArgs A;
    A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.la_hat = log(alpha_hat);
    A.prior_var = prior_var;
    A.cr_reg = cr_reg; A.prior_reg = prior_reg;
    KSum c;
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double yv = (double)y[n];
        c.add(lgamma_pos(yv + 1.0) - yv * log(mu[n]));
    }
    A.cst = Wv::sum_comp(c);
    const double lo = log(min_disp), hi = log(max_disp);
    auto fg = [&](double la, double& f, double& g) {
        alpha_eval<Wv, P, true>(A, la, cr_reg, prior_reg, f, g);
    };
    const Lbfgsb1dResult r = lbfgsb_1d(fg, A.la_hat, lo, hi);
    AlphaOut o;
    o.converged = r.success ? 1 : 0;
    o.nfev = r.nfev; o.nit = r.nit; o.status = r.status;
    if (r.success) o.alpha = exp(r.x);
    else o.alpha = exp(grid_fit_alpha<Wv, P>(A, lo, hi));
    return o;
}

}  // namespace dsq
