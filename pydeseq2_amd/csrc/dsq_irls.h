// dsq_irls.h — per-gene NB log-link GLM by iteratively reweighted least squares.
//
// Replaces pydeseq2/utils.py:273-438 (irls_solver).  The reference's loop is
//     while dev_ratio > beta_tol:
//         W = mu/(1+mu*disp); z = log(mu/sf) + (y-mu)/mu
//         beta_hat = solve(X^T W X + 1e-6 I, X^T W z)          (:368-371)
//         if any|beta_hat| > max_beta or i >= maxiter: -> bounded optimiser (:374-413)
//         beta = beta_hat; mu = max(sf*exp(X beta), min_mu)   (:415-416)
//         dev = -2*nb_nll(y, mu, disp); dev_ratio = |dev-old|/(|dev|+0.1)   (:418-421)
// One fused sweep over the samples per iteration produces the new mu, the deviance sum
// AND the next iteration's X^T W X, X^T W z (they depend only on the new mu), so each
// IRLS iteration reads the gene's counts exactly once.  The final X^T W X is also the
// one the hat diagonal needs (:427-433).  lgamma terms of the deviance do not depend on
// mu and are computed once per gene.
#pragma once
#include "dsq_alpha.h"
#include "dsq_lbfgsb.h"
#include "dsq_linalg.h"
#include "dsq_wave.h"

namespace dsq {

struct IrlsArgs {
    const int32_t* y;     // [N]
    const double* sf;     // [N]
    const double* lsf;    // [N] log(sf) (optional, nullptr -> computed on the fly)
    const double* Xt;     // [P][ldx]
    const double* pinvXt; // [P][ldx] rows of (X^T X)^-1 X^T  (QR initialisation, :349-353)
    int ldx, N;
    double disp, min_mu, beta_tol, min_beta, max_beta;
    int maxiter;
    bool full_rank;
};

// sweep: mu(beta) clamped, S = sum (y+a) log(a+mu) - y log mu, M = X^T W X, r = X^T W z
template <class Wv, int P>
DSQ_HD void irls_sweep(const IrlsArgs& A, const double (&beta)[P], double a, double& S,
                       double (&M)[Tri<P>::N], double (&r)[P]) {
    constexpr int T = Tri<P>::N;
    double s = 0.0;
    const double lmin = log(A.min_mu);
#pragma unroll
    for (int k = 0; k < T; ++k) M[k] = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) r[j] = 0.0;
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        const double sfn = A.sf[n];
        double x[P];
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * beta[j]; }
        // mu = max(sf exp(eta), min_mu).  While mu is not clamped, log(mu) = eta + log(sf) and
        // log(mu/sf) = eta hold exactly in real arithmetic, so only log(a + mu) needs a log.
        const double mu_raw = sfn * exp(eta);
        const bool clamped = !(mu_raw > A.min_mu);
        const double mu = clamped ? A.min_mu : mu_raw;
        const double lsfn = (A.lsf != nullptr) ? A.lsf[n] : flog(sfn);
        const double lmu = clamped ? lmin : eta + lsfn;
        const double rmu = frcp(mu);
        s += (yv + a) * flog(a + mu) - yv * lmu;
        const double w = mu * frcp(1.0 + mu * A.disp);
        const double z = (clamped ? lmin - lsfn : eta) + (yv - mu) * rmu;
        const double wz = w * z;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const double xw = x[i] * w;
            r[i] += x[i] * wz;
#pragma unroll
            for (int j = 0; j <= i; ++j) M[tri(i, j)] += xw * x[j];
        }
    }
    S = Wv::sum(s);
    Wv::template sum_n<T>(M);
    Wv::template sum_n<P>(r);
}

struct IrlsOut {
    int converged;
    int iters;
    int fallback;  // 1: IRLS diverged (|beta| > max_beta or maxiter) -> gene needs irls_rescue_gene
};

// initial beta (:349-357) and sum(logbinom), the mu-independent part of the NLL:
//   cst = sum[ lgamma(y+a) - lgamma(y+1) ] - N lgamma(a) = -sum[ (lgamma(a) - lgamma(y+a)) + log(y!) ]
// The gamma differences come from the wave memo of the dispersion kernel (lane k evaluates the
// count k once, counts >= 64 take the Stirling formula), log(y!) from the 256-entry table.  cst only
// enters the deviance's denominator |dev| + 0.1 (it cancels in dev - old_dev), so ulp-level
// differences from scipy's gammaln are immaterial.
template <class Wv, int P>
DSQ_HD void irls_init(const IrlsArgs& A, double a, double (&b0)[P], double& cst) {
#pragma unroll
    for (int j = 0; j < P; ++j) b0[j] = 0.0;
    double lga, dga_unused, tab_dl, tab_dd;
    lgamma_digamma<false>(a, lga, dga_unused);
    lgamma_digamma_diff<Wv, false>(Wv::lane(), a, lga, 0.0, tab_dl, tab_dd);
    double c = 0.0;
    for (int base = 0; base < A.N; base += Wv::W) {  // wave-uniform trip count (cross-lane memo reads)
        const int n = base + Wv::lane();
        const bool valid = n < A.N;
        const int yi = valid ? A.y[n] : 0;
        const double yv = (double)yi;
        const bool in_tab = yi < Wv::W;
        double dl = Wv::from_lane(tab_dl, in_tab ? yi : 0);
        if (Wv::any(!in_tab)) {
            double dl2, dd2;
            lgamma_digamma_diff<Wv, false>(in_tab ? 64 : yi, a, lga, 0.0, dl2, dd2);
            dl = in_tab ? dl : dl2;
        }
        const bool in_fact = yi < kLgammaIntN;
        double lf = kLgammaInt[in_fact ? yi : 0];
        if (Wv::any(!in_fact)) {
            const double z = in_fact ? 300.0 : yv + 1.0;
            const double big = (z - 0.5) * flog(z) - z + kHalfLog2Pi + stirling_tail(frcp(z));
            lf = in_fact ? lf : big;
        }
        c -= valid ? dl + lf : 0.0;
        if (valid) {
            if (A.full_rank) {
                // library log and true division: beta_init decides, on ill-conditioned genes, whether
                // IRLS diverges (and the gene goes to the rescue) exactly as it does in the reference
                const double ly = log(yv / A.sf[n] + 0.1);
#pragma unroll
                for (int j = 0; j < P; ++j) b0[j] += A.pinvXt[j * A.ldx + n] * ly;
            } else {
                b0[0] += log(yv / A.sf[n]);
            }
        }
    }
    cst = Wv::sum(c);
    Wv::template sum_n<P>(b0);
    if (!A.full_rank) b0[0] = b0[0] / (double)A.N;
}

// The same quantities with the reference's own arithmetic for cst (two lgamma evaluations per
// sample, utils.py:218-222).  The rescue of a diverged gene minimises f = nlogterm - cst + s with
// L-BFGS-B, whose line search and stopping tests react to the rounding of f itself on these
// ill-conditioned genes, so there cst is kept bit-for-bit what the parity tests were pinned on.
template <class Wv, int P>
DSQ_HD void irls_init_exact(const IrlsArgs& A, double a, double (&b0)[P], double& cst) {
#pragma unroll
    for (int j = 0; j < P; ++j) b0[j] = 0.0;
    double c = 0.0;
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        c += lgamma_pos(yv + a) - lgamma_pos(yv + 1.0);
        if (A.full_rank) {
            const double ly = log(yv / A.sf[n] + 0.1);
#pragma unroll
            for (int j = 0; j < P; ++j) b0[j] += A.pinvXt[j * A.ldx + n] * ly;
        } else {
            b0[0] += log(yv / A.sf[n]);
        }
    }
    cst = Wv::sum(c) - A.N * lgamma_pos(a);
    Wv::template sum_n<P>(b0);
    if (!A.full_rank) b0[0] = b0[0] / (double)A.N;
}

// hat diagonal (:427-433) from M = X^T W X at the final clamped mu, and unclamped mu (:435-437)
template <class Wv, int P>
DSQ_HD void irls_finish(const IrlsArgs& A, const double (&beta)[P], double (&M)[Tri<P>::N],
                        double* mu_out, double* H_out) {
    constexpr int T = Tri<P>::N;
    if (mu_out == nullptr && H_out == nullptr) return;
    double inv[T];
#pragma unroll
    for (int j = 0; j < P; ++j) M[tri(j, j)] += 1e-6;
    chol<P>(M);
    chol_inverse<P>(M, inv);
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        double x[P];
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * beta[j]; }
        const double mu_raw = A.sf[n] * exp(eta);
        if (mu_out != nullptr) mu_out[n] = mu_raw;
        if (H_out != nullptr) {
            const double mu = dmax(mu_raw, A.min_mu);
            const double w = mu / (1.0 + mu * A.disp);
            const double sw = sqrt(w);
            H_out[n] = sw * sym_quad<P>(inv, x) * sw;
        }
    }
}

// beta (out), mu_out[N] = UNclamped sf*exp(X beta), H_out[N] hat diagonal (either may be null).
// When IRLS diverges nothing is written and out.fallback = 1.
template <class Wv, int P>
DSQ_HD IrlsOut irls_gene(const IrlsArgs& A, double (&beta)[P], double* mu_out, double* H_out) {
    constexpr int T = Tri<P>::N;
    IrlsOut out;
    out.converged = 1; out.iters = 0; out.fallback = 0;
    const double a = 1.0 / A.disp;
    double cst;
    irls_init<Wv, P>(A, a, beta, cst);
    const double nlogterm = A.N * a * log(A.disp);
    double M[T], r[P], S;
    irls_sweep<Wv, P>(A, beta, a, S, M, r);
    double dev = 1000.0, ratio = 1.0;
    int i = 0;
    while (ratio > A.beta_tol) {
        double Hm[T];
#pragma unroll
        for (int k = 0; k < T; ++k) Hm[k] = M[k];
#pragma unroll
        for (int j = 0; j < P; ++j) Hm[tri(j, j)] += 1e-6;
        chol<P>(Hm);
        double bh[P];
#pragma unroll
        for (int j = 0; j < P; ++j) bh[j] = r[j];
        chol_solve<P>(Hm, bh);
        i += 1;
        bool bad = (i >= A.maxiter);
#pragma unroll
        for (int j = 0; j < P; ++j) bad = bad || (fabs(bh[j]) > A.max_beta);  // NaN is not "bad" (as in the reference)
        if (bad) {
            out.fallback = 1; out.converged = 0; out.iters = i;
            return out;
        }
#pragma unroll
        for (int j = 0; j < P; ++j) beta[j] = bh[j];
        irls_sweep<Wv, P>(A, beta, a, S, M, r);
        const double old = dev;
        dev = -2.0 * (nlogterm - cst + S);
        ratio = fabs(dev - old) / (fabs(dev) + 0.1);
    }
    out.iters = i;
    irls_finish<Wv, P>(A, beta, M, mu_out, H_out);
    return out;
}

// Workspace of the rescue (wave-private LDS on the device).
template <int P>
struct IrlsRescueWork {
    LbfgsbWork<P> lb;
    double x[P], l[P], u[P];
    int nbd[P];
};

// grid_fit_beta (grid_search.py:145-221): two-level 60 x 60 grid on [min_beta, max_beta]^2, P == 2.
template <class Wv>
DSQ_HD void grid_fit_beta2(const IrlsArgs& A, double a, double cst, double (&beta)[2],
                           int grid_length = 60) {
    auto loss = [&](double bx, double by) {
        double s = 0.0;
        for (int n = Wv::lane(); n < A.N; n += Wv::W) {
            const double yv = (double)A.y[n];
            const double eta = A.Xt[n] * bx + A.Xt[A.ldx + n] * by;
            const double mu = dmax(A.sf[n] * exp(eta), A.min_mu);
            s += (yv + a) * log(mu + a) - yv * log(mu);
        }
        s = Wv::sum(s);
        return (A.N * a * log(A.disp) - cst + s) + 0.5 * (1e-6 * bx * bx + 1e-6 * by * by);
    };
    double xlo = A.min_beta, xhi = A.max_beta, ylo = A.min_beta, yhi = A.max_beta;
    for (int level = 0; level < 2; ++level) {
        double best = 0.0;
        int bi = 0, bj = 0;
        bool best_nan = false, first = true;
        for (int i = 0; i < grid_length; ++i) {
            for (int j = 0; j < grid_length; ++j) {
                const double v = loss(linspace_at(xlo, xhi, grid_length, i), linspace_at(ylo, yhi, grid_length, j));
                const bool isn = (v != v);
                if (first || (!best_nan && (isn || v < best))) { best = v; bi = i; bj = j; best_nan = isn; first = false; }
            }
        }
        const double cx = linspace_at(xlo, xhi, grid_length, bi), cy = linspace_at(ylo, yhi, grid_length, bj);
        if (level == 0) {
            const double delta = linspace_at(xlo, xhi, grid_length, 1) - linspace_at(xlo, xhi, grid_length, 0);
            xlo = cx - delta; xhi = cx + delta; ylo = cy - delta; yhi = cy + delta;
        } else {
            beta[0] = cx; beta[1] = cy;
        }
    }
}

// Rescue for genes whose IRLS diverged (utils.py:374-413): L-BFGS-B on
//   f(b) = nb_nll(y, max(sf exp(X b), min_mu), disp) + 0.5 * 1e-6 |b|^2,  -30 <= b <= 30,
// restarted from beta_init, gradient as utils.py:381-387 (it ignores the clamp, as the
// reference does); if that does not converge and P <= 2 the 2-D grid search.  `converged`
// is scipy's res.success.
template <class Wv, int P>
DSQ_HD IrlsOut irls_rescue_gene(const IrlsArgs& A, IrlsRescueWork<P>& Wk, double (&beta)[P],
                                double* mu_out, double* H_out) {
    constexpr int T = Tri<P>::N;
    IrlsOut out;
    out.converged = 0; out.iters = 0; out.fallback = 1;
    const double a = 1.0 / A.disp;
    double cst, b0[P];
    irls_init_exact<Wv, P>(A, a, b0, cst);
    const double nlogterm = A.N * a * log(A.disp);
#pragma unroll
    for (int j = 0; j < P; ++j) { Wk.x[j] = b0[j]; Wk.l[j] = A.min_beta; Wk.u[j] = A.max_beta; Wk.nbd[j] = 2; }
    auto fg = [&](const double* xb, double& f, double* g) {
        double b[P];
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] = xb[j];
        double s = 0.0, gr[P];
#pragma unroll
        for (int j = 0; j < P; ++j) gr[j] = 0.0;
        for (int n = Wv::lane(); n < A.N; n += Wv::W) {
            const double yv = (double)A.y[n];
            double x[P];
            double eta = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * b[j]; }
            const double mu = dmax(A.sf[n] * exp(eta), A.min_mu);
            s += (yv + a) * log(a + mu) - yv * log(mu);
            const double gk = -yv + (a + yv) * mu / (a + mu);
#pragma unroll
            for (int j = 0; j < P; ++j) gr[j] += gk * x[j];
        }
        s = Wv::sum(s);
        Wv::template sum_n<P>(gr);
        double pen = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) { pen += 1e-6 * (b[j] * b[j]); g[j] = gr[j] + 1e-6 * b[j]; }
        f = (nlogterm - cst + s) + 0.5 * pen;
    };
    const LbfgsbResult res = lbfgsb_nd<P>(fg, P, Wk.x, Wk.l, Wk.u, Wk.nbd, Wk.lb);
#pragma unroll
    for (int j = 0; j < P; ++j) beta[j] = Wk.x[j];
    out.converged = res.success ? 1 : 0;
    out.iters = res.nit;
    if (!res.success && P <= 2) {
        if constexpr (P == 2) grid_fit_beta2<Wv>(A, a, cst, beta);
    }
    double M[T], r[P], S2;
    irls_sweep<Wv, P>(A, beta, a, S2, M, r);  // M = X^T W X at the final (clamped) mu
    irls_finish<Wv, P>(A, beta, M, mu_out, H_out);
    return out;
}

}  // namespace dsq
