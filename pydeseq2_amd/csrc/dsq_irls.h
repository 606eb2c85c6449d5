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
#include "dsq_linalg.h"
#include "dsq_wave.h"

namespace dsq {

struct IrlsArgs {
    const int32_t* y;     // [N]
    const double* sf;     // [N]
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
        const double mu = dmax(sfn * exp(eta), A.min_mu);
        s += (yv + a) * log(a + mu) - yv * log(mu);
        const double w = mu / (1.0 + mu * A.disp);
        const double z = log(mu / sfn) + (yv - mu) / mu;
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

// objective/gradient/Fisher matrix of the fallback problem (utils.py:376-387):
//   f = nb_nll(y, max(sf exp(X b), min_mu), disp) + 0.5*1e-6*|b|^2
template <class Wv, int P>
DSQ_HD void irls_fb_eval(const IrlsArgs& A, const double (&beta)[P], double a, double cst,
                         double& f, double (&grad)[P], double (&M)[Tri<P>::N]) {
    constexpr int T = Tri<P>::N;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < T; ++k) M[k] = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) grad[j] = 0.0;
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        double x[P];
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * beta[j]; }
        const double mu = dmax(A.sf[n] * exp(eta), A.min_mu);
        s += (yv + a) * log(a + mu) - yv * log(mu);
        const double gk = -yv + (a + yv) * mu / (a + mu);
        const double w = mu / (1.0 + mu * A.disp);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            grad[i] += gk * x[i];
            const double xw = x[i] * w;
#pragma unroll
            for (int j = 0; j <= i; ++j) M[tri(i, j)] += xw * x[j];
        }
    }
    s = Wv::sum(s);
    Wv::template sum_n<T>(M);
    Wv::template sum_n<P>(grad);
    double pen = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) { pen += beta[j] * beta[j]; grad[j] += 1e-6 * beta[j]; }
    f = A.N * a * log(A.disp) - cst + s + 0.5e-6 * pen;
}

struct IrlsOut {
    int converged;
    int iters;
    int fallback;  // 1: IRLS diverged (|beta| > max_beta or maxiter) -> gene needs irls_rescue_gene
};

// initial beta (:349-357) and sum(logbinom), the mu-independent part of the NLL
template <class Wv, int P>
DSQ_HD void irls_init(const IrlsArgs& A, double a, double (&b0)[P], double& cst) {
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

// Rescue for genes whose IRLS diverged: bounded, damped Fisher scoring on the reference's
// fallback objective, restarted from beta_init (utils.py:374-403).  The reference hands
// this problem to scipy's p-dimensional L-BFGS-B; both converge to the same bounded
// optimum but stop at slightly different points (see DESIGN.md "known deviations").
template <class Wv, int P>
DSQ_HD IrlsOut irls_rescue_gene(const IrlsArgs& A, double (&beta)[P], double* mu_out,
                                double* H_out) {
    constexpr int T = Tri<P>::N;
    IrlsOut out;
    out.converged = 0; out.iters = 0; out.fallback = 1;
    const double a = 1.0 / A.disp;
    double cst, b0[P];
    irls_init<Wv, P>(A, a, b0, cst);
#pragma unroll
    for (int j = 0; j < P; ++j) beta[j] = dmin(dmax(b0[j], A.min_beta), A.max_beta);
    double f, g[P], M[T];
    irls_fb_eval<Wv, P>(A, beta, a, cst, f, g, M);
    bool ok = false;
    int it = 0;
    for (; it < 200 && !ok; ++it) {
        double pg = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            double gj = g[j];
            if (gj < 0.0) gj = dmax(beta[j] - A.max_beta, gj);
            else gj = dmin(beta[j] - A.min_beta, gj);
            pg = dmax(pg, fabs(gj));
        }
        if (pg <= 1e-5) { ok = true; break; }
        double Hf[T];
#pragma unroll
        for (int k = 0; k < T; ++k) Hf[k] = M[k];
#pragma unroll
        for (int j = 0; j < P; ++j) Hf[tri(j, j)] += 1e-6;
        chol<P>(Hf);
        double d[P];
#pragma unroll
        for (int j = 0; j < P; ++j) d[j] = -g[j];
        chol_solve<P>(Hf, d);
        double t = 1.0;
        bool accepted = false;
        for (int h = 0; h < 30; ++h) {
            double bn[P], fn, gn[P], Mn[T];
#pragma unroll
            for (int j = 0; j < P; ++j)
                bn[j] = dmin(dmax(beta[j] + t * d[j], A.min_beta), A.max_beta);
            irls_fb_eval<Wv, P>(A, bn, a, cst, fn, gn, Mn);
            if (fn <= f) {
                const double df = f - fn;
#pragma unroll
                for (int j = 0; j < P; ++j) { beta[j] = bn[j]; g[j] = gn[j]; }
#pragma unroll
                for (int k = 0; k < T; ++k) M[k] = Mn[k];
                const double fm = dmax(fabs(f), dmax(fabs(fn), 1.0));
                f = fn;
                accepted = true;
                if (df <= 1e7 * kEps * fm) ok = true;
                break;
            }
            t *= 0.5;
        }
        if (!accepted) break;
    }
    out.converged = ok ? 1 : 0;
    out.iters = it;
    irls_finish<Wv, P>(A, beta, M, mu_out, H_out);
    return out;
}

}  // namespace dsq
