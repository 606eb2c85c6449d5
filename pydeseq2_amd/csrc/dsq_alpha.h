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
#include "dsq_lgamma_int.h"
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

// lgamma(a) - lgamma(y + a) and digamma(a) - digamma(y + a) for a count y >= 0.
//   y <= 9 : exact recurrences  -log prod_{i<y}(a+i),  -sum_{i<y} 1/(a+i)   (y = 0 costs nothing)
//   y >= 10: Stirling series at z = y + a >= 10 minus the per-gene lgamma(a), digamma(a)
// (the reference evaluates gammaln / polygamma at both arguments and subtracts, utils.py:218,
// 263-264; the differences are what enters the likelihood).
constexpr int kSmallCount = 9;

template <class Wv, bool GRAD>
DSQ_HD void lgamma_digamma_diff(int yi, double a, double lga, double dga, double& dl, double& dd) {
    const bool small = yi <= kSmallCount;
    // small counts: prod = prod_{i<y}(a+i), num/prod = sum_{i<y} 1/(a+i)
    double prod = 1.0, num = 0.0;
    if (Wv::any(small && yi > 0)) {
#pragma unroll
        for (int i = 0; i < kSmallCount; ++i) {
            if (i < yi && small) {
                const double t = a + (double)i;
                if (GRAD) num = num * t + prod;
                prod *= t;
            }
        }
    }
    // ONE log and ONE reciprocal serve both branches (each lane needs only its own)
    const double z = (double)yi + a;
    const double arg = small ? prod : z;
    const double lg = flog(arg);
    const double rc = frcp(arg);
    if (small) {
        dl = -lg;
        dd = GRAD ? -(num * rc) : 0.0;
    } else {
        dl = lga - ((z - 0.5) * lg - z + kHalfLog2Pi + stirling_tail(rc));
        dd = GRAD ? dga - (lg + digamma_tail(rc)) : 0.0;
    }
}

// loss (and gradient) of the Cox-Reid / prior regularised NB negative log-likelihood at
// log_alpha.  GRAD = false is used by the grid search.
//
// Per sample, with L1 = log1p(mu*alpha) and a = 1/alpha, the reference's
//     n*a*log(alpha) + sum[ -logbinom + (y+a) log(a+mu) - y log mu ]          (utils.py:227-234)
// is evaluated as  sum[ (lgamma(a) - lgamma(y+a)) + y (L1 - log alpha) + a L1 ] + cst, i.e. the
// n*a*log(alpha) term is folded into the sum analytically (it cancels a*log(a+mu) to O(mu)), which
// removes the reference's largest rounding-noise source instead of reproducing it; one log1p
// serves the loss, the gradient's log(1 + mu alpha) (utils.py:265) and, through its argument's
// reciprocal, both W = mu/(1+mu alpha) and (y-mu)/(mu+a).
template <class Wv, int P, bool GRAD>
DSQ_HD void alpha_eval(const AlphaArgs& A, double la, bool cr_reg, bool prior_reg, double& f,
                       double& g) {
    constexpr int T = Tri<P>::N;
    la = Wv::uniform(la);
    const double alpha = Wv::uniform(exp(la));
    const double a = Wv::uniform(1.0 / alpha);
    // log of the ROUNDED alpha: keeps every term a function of the same alpha (using `la` itself
    // would leave an inconsistency of ulp(1) * sum(y) in the loss, i.e. line-search noise)
    const double lal = Wv::uniform(log(alpha));
    double lga, dga;
    lgamma_digamma<GRAD>(a, lga, dga);
    lga = Wv::uniform(lga);
    dga = Wv::uniform(dga);
    // Wave-level memo: lane k evaluates the two gamma-function differences for the COUNT k once
    // per evaluation; samples whose count is < 64 then fetch them with a cross-lane read instead
    // of recomputing log/Stirling/recurrences per sample (counts repeat heavily within a gene).
    // Only counts >= 64 take the per-sample Stirling path.  (Host build: table of one entry.)
    double tab_dl, tab_dd;
    lgamma_digamma_diff<Wv, GRAD>(Wv::lane(), a, lga, dga, tab_dl, tab_dd);
    KSum accf;
    double accg = 0.0;
    double M[T], dM[T];
#pragma unroll
    for (int k = 0; k < T; ++k) { M[k] = 0.0; dM[k] = 0.0; }
    for (int base = 0; base < A.N; base += Wv::W) {  // wave-uniform trip count: all lanes stay active
        const int n = base + Wv::lane();
        const bool valid = n < A.N;
        const int yi = valid ? A.y[n] : 0;
        const double yv = (double)yi;
        const double m = valid ? A.mu[n] : 1.0;
        const bool in_tab = yi < Wv::W;
        double dl = Wv::from_lane(tab_dl, in_tab ? yi : 0);
        double dd = GRAD ? Wv::from_lane(tab_dd, in_tab ? yi : 0) : 0.0;
        if (Wv::any(!in_tab)) {
            double dl2, dd2;
            lgamma_digamma_diff<Wv, GRAD>(in_tab ? 64 : yi, a, lga, dga, dl2, dd2);
            if (!in_tab) { dl = dl2; dd = dd2; }
        }
        const double ma = m * alpha;
        const double r1 = frcp(1.0 + ma);
        const double L1 = flog1p(ma);
        const double vz = valid ? 1.0 : 0.0;
        accf.add(vz * (dl + yv * (L1 - lal) + a * L1));
        if (GRAD) accg += vz * (dd + L1 + (yv - m) * alpha * r1);
        if (cr_reg) {
            const double w = vz * (m * r1);
            const double dw = -(w * w);
            double x[P];
#pragma unroll
            for (int j = 0; j < P; ++j) x[j] = valid ? A.Xt[j * A.ldx + n] : 0.0;
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
    f = sumf + A.cst;
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

// alpha-independent part of the NLL:  sum lgamma(y+1) - y log(mu)      (utils.py:227-234)
// lgamma(y+1) = log(y!) comes from a 256-entry table (correctly rounded) and from the Stirling
// series at z = y + 1 >= 257 beyond it; log(mu) through the lean log (mu >= min_mu > 0).
template <class Wv>
DSQ_HD double alpha_const(const int32_t* y, const double* mu, int N) {
    KSum c;
    for (int base = 0; base < N; base += Wv::W) {
        const int n = base + Wv::lane();
        const bool valid = n < N;
        const int yi = valid ? y[n] : 0;
        const double yv = (double)yi;
        const double m = valid ? mu[n] : 1.0;
        const bool in_tab = yi < kLgammaIntN;
        double lg = kLgammaInt[in_tab ? yi : 0];
        if (Wv::any(!in_tab)) {
            const double z = in_tab ? 300.0 : yv + 1.0;
            const double lz = flog(z);
            const double big = (z - 0.5) * lz - z + kHalfLog2Pi + stirling_tail(frcp(z));
            if (!in_tab) lg = big;
        }
        c.add(lg - yv * flog(m));
    }
    return Wv::sum_comp(c);
}

// one gene: L-BFGS-B in log(alpha) from log(alpha_hat).  RUN_GRID: on non-convergence run the
// reference's grid search right here (host simulation / single-kernel use); otherwise only report
// converged = 0 and the caller schedules grid_alpha_gene for the gene (device: second tiny kernel,
// which keeps the 200-evaluation grid code out of the main kernel's register budget).
template <class Wv, int P, bool RUN_GRID>
DSQ_HD AlphaOut fit_alpha_gene(const int32_t* y, const double* mu, const double* Xt, int ldx, int N,
                               double alpha_hat, double min_disp, double max_disp,
                               double prior_var, bool cr_reg, bool prior_reg, Lbfgsb1d& m,
                               const double* cst_in = nullptr, double* cst_out = nullptr) {
    AlphaArgs A;
    A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.la_hat = log(alpha_hat);
    A.prior_var = prior_var;
    A.cr_reg = cr_reg; A.prior_reg = prior_reg;
    // the constant depends on (y, mu) only: the MAP fit re-uses the one the MLE fit stored
    A.cst = cst_in != nullptr ? *cst_in : alpha_const<Wv>(y, mu, N);
    if (cst_out != nullptr && Wv::lane() == 0) *cst_out = A.cst;
    const double lo = log(min_disp), hi = log(max_disp);
    m.start(A.la_hat, lo, hi);
    while (!m.done) {
        double f, g;
        alpha_eval<Wv, P, true>(A, m.x, cr_reg, prior_reg, f, g);
        m.feed(f, g);
    }
    AlphaOut o;
    o.converged = m.success ? 1 : 0;
    o.nfev = m.nfev; o.nit = m.it; o.status = m.status;
    o.alpha = exp(m.x);
    if (RUN_GRID && !m.success) o.alpha = exp(grid_fit_alpha<Wv, P>(A, lo, hi));
    return o;
}

// grid-search fallback of one gene (utils.py:556-564)
template <class Wv, int P>
DSQ_HD double grid_alpha_gene(const int32_t* y, const double* mu, const double* Xt, int ldx, int N,
                              double min_disp, double max_disp) {
    AlphaArgs A;
    A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.la_hat = 0.0; A.prior_var = 1.0; A.cr_reg = true; A.prior_reg = false;
    A.cst = alpha_const<Wv>(y, mu, N);
    return exp(grid_fit_alpha<Wv, P>(A, log(min_disp), log(max_disp)));
}

}  // namespace dsq
