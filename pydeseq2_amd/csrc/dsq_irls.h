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
#include "dsq_bfgs.h"
#include "dsq_lbfgsb.h"
#include "dsq_linalg.h"
#include "dsq_stats.h"
#include "dsq_wave.h"

// The per-sample loops of this file are long dependent fp64 chains behind a handful of loads; two samples per
// trip give the scheduler two independent chains to interleave (the SQ counters showed these kernels stalled on
// instruction dependencies and waitcnt for two thirds of their cycles, not on issue bandwidth).  Measured (A/B
// builds, c5-shaped p = 8 design without cells): -12 % on the general loops; the cell-path loops (LDS atomics,
// per-cell selects) got slower with it and stay rolled.
#ifndef DSQ_IRLS_UNROLL
#define DSQ_IRLS_UNROLL 2
#endif
#define DSQ_PRAGMA_(x) _Pragma(#x)
#define DSQ_PRAGMA(x) DSQ_PRAGMA_(x)
#define DSQ_UNROLL2 DSQ_PRAGMA(unroll DSQ_IRLS_UNROLL)
// samples per trip of the sixteen-lane (RowWave) cell loops
#ifndef DSQ_IRLS_ROW_U
#define DSQ_IRLS_ROW_U 4
#endif

namespace dsq {

constexpr int kIrlsRowU = DSQ_IRLS_ROW_U;

// Optional fused tail of the LFC fit: the per-sample part of the Cook's distances (dds.py:986-1040 and the
// outlier bookkeeping of dds.py:1066-1110, 1325-1326; the robust dispersion comes from its own kernel, it does
// not depend on the fit) and the Wald statistics (ds.py:303-360) are computed while mu and the hat diagonal are
// still in registers, so the N x G layers need not be written and re-read (16 N bytes per gene each way).
struct LfcEpilogue {
    // Cook's (robust_disp < 0 or NaN handling as in cooks_gene): enabled when flags != nullptr
    const uint8_t* flags = nullptr;  // [N]
    double robust_disp = 0.0, cutoff = 0.0;
    double* cooks_row = nullptr;     // this gene's row of the cooks layer (nullable)
    CooksOut cooks;
    // Wald: enabled when ridge != nullptr
    const double* ridge = nullptr;     // [P*P]
    const double* contrast = nullptr;  // [P]
    double lfc_null = 0.0;
    int alt = 0;
    WaldOut wald;
};

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
    const CellDesign* cells = nullptr;  // CELL instantiations (designs with <= 64 distinct rows)
    const double* pinvc = nullptr;      // [C][P] column of (X^T X)^-1 X^T of each cell (LDS; k_irls_row, full rank)
    void* cell_ws = nullptr;            // this wave's CellWork<P>
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
    DSQ_UNROLL2
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
        s += (yv + a) * flog_t(a + mu) - yv * lmu;
        // w = mu / (1 + mu disp), w z = w (eta + (y - mu) / mu) = (mu eta + y - mu) / (1 + mu disp): one reciprocal
        const double rd = frcp(1.0 + mu * A.disp);
        const double w = mu * rd;
        const double wz = (mu * (clamped ? lmin - lsfn : eta) + (yv - mu)) * rd;
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

// sum_c col[c * LD] * acc[c]  (col: a column of a per-cell table in LDS, acc: per-cell sums in LDS)
template <int LD, class Col, class Acc>
DSQ_HD double cell_dot(Col col, const Acc& acc, int C) {
    double v0 = 0.0, v1 = 0.0;
    int c = 0;
#pragma nounroll
    for (; c + 3 < C; c += 4) {
        v0 += col[0] * acc[c];
        v1 += col[LD] * acc[c + 1];
        v0 += col[2 * LD] * acc[c + 2];
        v1 += col[3 * LD] * acc[c + 3];
        col += 4 * LD;
    }
#pragma nounroll
    for (; c < C; ++c) { v0 += col[0] * acc[c]; col += LD; }
    return v0 + v1;
}

// The same sweep for a design with few distinct rows (dsq_linalg.h, CellDesign): the linear predictor and its
// exponential are computed once per CELL, a sample fetches them by its cell index, adds its weight w and w z
// into the cell's accumulators, and X^T W X, X^T W z are rebuilt from the <= 64 cell sums entry-parallel.  The
// per-sample work no longer depends on P.
// LOADM = false (sixteen-lane rows): X^T W X and X^T W z stay in the workspace (Wk.ent), where row_chol_solve and the
// finish read them - no lane holds the whole matrix
template <class Wv, int P, bool LOADM = true>
DSQ_HD void irls_sweep_cell(const IrlsArgs& A, const double (&beta)[P], double a, double& S,
                            double (&M)[Tri<P>::N], double (&r)[P]) {
    constexpr int T = Tri<P>::N;
    const CellDesign& D = *A.cells;
    typedef DSQ_LDS_STRUCT(CellWork<P>) LdsWork;  // ds_read / ds_write instead of flat accesses
    LdsWork& Wk = *(LdsWork*)A.cell_ws;
    const auto Xc_ = DSQ_AS_LDS(double, D.Xc);
    const auto XX_ = DSQ_AS_LDS(double, D.XX);
    for (int c = Wv::lane(); c < D.C; c += Wv::W) {
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) eta += Xc_[c * P + j] * beta[j];
        Wk.tab[0][c] = eta;
        Wk.tab[1][c] = exp(eta);
        Wk.acc[0][c] = 0.0;
        Wk.acc[1][c] = 0.0;
    }
    Wv::sync();
    double s = 0.0;
    const double lmin = log(A.min_mu);
    DSQ_PHASE(3);
    // one sample: its term of the deviance sum, its weight and weight x working response
    auto sample = [&](double yv, double sfn, double lsfn, double eta, double e, double& w, double& wz) {
        const double mu_raw = sfn * e;
        const bool clamped = !(mu_raw > A.min_mu);
        const double mu = clamped ? A.min_mu : mu_raw;
        const double lmu = clamped ? lmin : eta + lsfn;
        s += (yv + a) * flog_t(a + mu) - yv * lmu;
        const double rd = frcp(1.0 + mu * A.disp);  // (see irls_sweep)
        w = mu * rd;
        wz = (mu * (clamped ? lmin - lsfn : eta) + (yv - mu)) * rd;
    };
    int n = Wv::lane();
    if constexpr (Wv::W == 16) {
        // sixteen lanes per gene take N / 16 trips: four samples per trip, loads first, so that four independent
        // chains hide the load and the transcendental latencies (two wavefronts per SIMD do not)
        constexpr int U = kIrlsRowU;
        for (; n + (U - 1) * Wv::W < A.N; n += U * Wv::W) {
            double yv[U], sfn[U], lsfn[U], eta[U], e[U], w[U], wz[U];
            int cell[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = n + u * Wv::W;
                yv[u] = (double)A.y[m];
                sfn[u] = A.sf[m];
                cell[u] = D.cell_of[m];
                lsfn[u] = (A.lsf != nullptr) ? A.lsf[m] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { eta[u] = Wk.tab[0][cell[u]]; e[u] = Wk.tab[1][cell[u]]; }
            if (A.lsf == nullptr) {
#pragma unroll
                for (int u = 0; u < U; ++u) lsfn[u] = flog(sfn[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) sample(yv[u], sfn[u], lsfn[u], eta[u], e[u], w[u], wz[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                Wv::cell_add(&Wk.acc[0][cell[u]], w[u]);
                Wv::cell_add(&Wk.acc[1][cell[u]], wz[u]);
            }
        }
    }
    for (; n < A.N; n += Wv::W) {
        const double sfn = A.sf[n];
        const int cell = D.cell_of[n];
        const double lsfn = (A.lsf != nullptr) ? A.lsf[n] : flog(sfn);
        double w, wz;
        sample((double)A.y[n], sfn, lsfn, Wk.tab[0][cell], Wk.tab[1][cell], w, wz);
        Wv::cell_add(&Wk.acc[0][cell], w);
        Wv::cell_add(&Wk.acc[1][cell], wz);
    }
    DSQ_PHASE(4);
    Wv::sync();
    // entries of X^T W X and X^T W z: a lane owns every W-th entry and walks the cells once for all of them (one
    // broadcast read of the cell's two sums serves its (T + P) / W independent chains)
    {
        constexpr int NE = (T + Wv::W - 1) / Wv::W, NR = (P + Wv::W - 1) / Wv::W;
        int em[NE], er[NR];
        double vm[NE], vr[NR];
#pragma unroll
        for (int k = 0; k < NE; ++k) { em[k] = Wv::lane() + k * Wv::W; em[k] = em[k] < T ? em[k] : T - 1; vm[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < NR; ++k) { er[k] = Wv::lane() + k * Wv::W; er[k] = er[k] < P ? er[k] : P - 1; vr[k] = 0.0; }
#pragma unroll 4
        for (int c = 0; c < D.C; ++c) {
            const double a0 = Wk.acc[0][c], a1 = Wk.acc[1][c];
#pragma unroll
            for (int k = 0; k < NE; ++k) vm[k] += XX_[c * T + em[k]] * a0;
#pragma unroll
            for (int k = 0; k < NR; ++k) vr[k] += Xc_[c * P + er[k]] * a1;
        }
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (Wv::lane() + k * Wv::W < T) Wk.ent[Wv::lane() + k * Wv::W] = vm[k];
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (Wv::lane() + k * Wv::W < P) Wk.ent[T + Wv::lane() + k * Wv::W] = vr[k];
    }
    Wv::sync();
    if constexpr (LOADM) {
#pragma unroll
        for (int k = 0; k < T; ++k) M[k] = Wk.ent[k];
#pragma unroll
        for (int j = 0; j < P; ++j) r[j] = Wk.ent[T + j];
        Wv::sync();  // ent is rewritten by the next sweep
    }
    S = Wv::sum(s);
}

// Designs with at most kSmallCells distinct rows (the two-group comparison of most experiments): the per-cell
// linear predictors and their exponentials are wave-uniform registers, a sample selects its cell's pair (no exp,
// no design loads, no outer product per sample) and adds its weight into per-cell register accumulators.
template <int CS>
DSQ_HD double cell_select(int cell, const double (&v)[CS]) {
    double r = v[0];
#pragma unroll
    for (int c = 1; c < CS; ++c) r = (cell == c) ? v[c] : r;
    return r;
}

template <class Wv, int P, int CS>
DSQ_HD void irls_sweep_cs(const IrlsArgs& A, const double (&beta)[P], double a, double& S,
                          double (&M)[Tri<P>::N], double (&r)[P], double (&e_c)[CS]) {
    constexpr int T = Tri<P>::N;
    const CellDesign& D = *A.cells;
    const auto Xc_ = DSQ_AS_LDS(double, D.Xc);
    const auto XX_ = DSQ_AS_LDS(double, D.XX);
    double eta_c[CS], sw[CS], swz[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        double eta = 0.0;
        if (c < D.C) {
#pragma unroll
            for (int j = 0; j < P; ++j) eta += Xc_[c * P + j] * beta[j];
        }
        eta_c[c] = Wv::uniform(eta);
        e_c[c] = Wv::uniform(exp(eta));
        sw[c] = 0.0;
        swz[c] = 0.0;
    }
    double s = 0.0;
    const double lmin = log(A.min_mu);
    DSQ_PHASE(3);
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        const double sfn = A.sf[n];
        const int cell = D.cell_of[n];
        const double eta = cell_select<CS>(cell, eta_c);
        const double mu_raw = sfn * cell_select<CS>(cell, e_c);
        const bool clamped = !(mu_raw > A.min_mu);
        const double mu = clamped ? A.min_mu : mu_raw;
        const double lsfn = (A.lsf != nullptr) ? A.lsf[n] : flog(sfn);
        const double lmu = clamped ? lmin : eta + lsfn;
        s += (yv + a) * flog_t(a + mu) - yv * lmu;
        const double rd = frcp(1.0 + mu * A.disp);  // (see irls_sweep)
        const double w = mu * rd;
        const double wz = (mu * (clamped ? lmin - lsfn : eta) + (yv - mu)) * rd;
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            sw[c] += (cell == c) ? w : 0.0;
            swz[c] += (cell == c) ? wz : 0.0;
        }
    }
    DSQ_PHASE(4);
    S = Wv::sum(s);
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        if (c < D.C) { sw[c] = Wv::sum(sw[c]); swz[c] = Wv::sum(swz[c]); }  // D.C is wave-uniform
    }
#pragma unroll
    for (int k = 0; k < T; ++k) M[k] = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) r[j] = 0.0;
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        if (c < D.C) {
#pragma unroll
            for (int k = 0; k < T; ++k) M[k] += XX_[c * T + k] * sw[c];
#pragma unroll
            for (int j = 0; j < P; ++j) r[j] += Xc_[c * P + j] * swz[c];
        }
    }
}

template <class Wv, int P, int CS>
DSQ_HD void irls_finish_cs(const IrlsArgs& A, const double (&beta)[P], double (&M)[Tri<P>::N],
                           const double (&e_c)[CS], double* mu_out, double* H_out, LfcEpilogue* E = nullptr) {
    constexpr int T = Tri<P>::N;
    const bool want_cooks = E != nullptr && E->flags != nullptr;
    const bool want_wald = E != nullptr && E->ridge != nullptr;
    if (mu_out == nullptr && H_out == nullptr && !want_cooks && !want_wald) return;
    const CellDesign& D = *A.cells;
    const auto Xc_ = DSQ_AS_LDS(double, D.Xc);
    const auto XX_ = DSQ_AS_LDS(double, D.XX);
    double q_c[CS], swu[CS];
    {
        double rinv[P];
#pragma unroll
        for (int j = 0; j < P; ++j) M[tri(j, j)] += 1e-6;
        chol<P>(M);
        chol_rinv<P>(M, rinv);
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            double q = 0.0;
            if (c < D.C) {
                double x[P];
#pragma unroll
                for (int j = 0; j < P; ++j) x[j] = Xc_[c * P + j];
                q = chol_quad<P>(M, rinv, x);
            }
            q_c[c] = Wv::uniform(q);
            swu[c] = 0.0;
        }
    }
    CooksAcc<Wv> acc(want_cooks ? E->robust_disp : 0.0, want_cooks ? E->cutoff : 0.0, P);
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const int cell = D.cell_of[n];
        const double mu_raw = A.sf[n] * cell_select<CS>(cell, e_c);
        if (mu_out != nullptr) mu_out[n] = mu_raw;
        double w = 0.0;
        const bool have_w = H_out != nullptr || want_cooks;
        if (have_w) {
            const double mu = dmax(mu_raw, A.min_mu);
            w = mu * frcp_g(1.0 + mu * A.disp);
            const double h = w * cell_select<CS>(cell, q_c);  // (sqrt(w) q sqrt(w) of utils.py:430-433, to an ulp)
            if (H_out != nullptr) H_out[n] = h;
            if (want_cooks) {
                const double ck = acc.add(n, (double)A.y[n], mu_raw, h, E->flags[n]);
                if (E->cooks_row != nullptr) E->cooks_row[n] = ck;
            }
        }
        if (want_wald) {
            // the Wald weight takes the UNclamped mu (ds.py:320-324): the same number unless a lane was clamped
            double wu = w;
            if (!have_w || Wv::any(!(mu_raw >= A.min_mu))) wu = mu_raw * frcp_g(1.0 + mu_raw * A.disp);
#pragma unroll
            for (int c = 0; c < CS; ++c) swu[c] += (cell == c) ? wu : 0.0;
        }
    }
    if (want_cooks) E->cooks = acc.finish(A.y, A.N);
    if (want_wald) {
        double Mw[T];
#pragma unroll
        for (int k = 0; k < T; ++k) Mw[k] = 0.0;
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            if (c < D.C) {
                const double sc = Wv::sum(swu[c]);
#pragma unroll
                for (int k = 0; k < T; ++k) Mw[k] += XX_[c * T + k] * sc;
            }
        }
        E->wald = wald_from_M<P>(Mw, beta, E->ridge, E->contrast, E->lfc_null, E->alt);
    }
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
    lgamma_digamma<false, true>(a, lga, dga_unused);
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
            const double z = in_fact ? 300.0 : yv + 1.0;  // z >= 257: the table logarithm and the truncated tail
            const double big = (z - 0.5) * flog_t(z) - z + kHalfLog2Pi + stirling_tail_big(frcp(z));
            lf = in_fact ? lf : big;
        }
        c -= valid ? dl + lf : 0.0;
        if (valid) {
            if (A.full_rank) {
                // log(y / sf + 0.1) (utils.py:351) with the lean log and a reciprocal (<= 1 ulp each; the sum over
                // the samples runs in another order than numpy's anyway): the library log + IEEE division here
                // were a sixth of the kernel
                const double ly = flog_t(yv * frcp(A.sf[n]) + 0.1);
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

// irls_init for a full-rank cell design on sixteen-lane rows: the columns of (X^T X)^-1 X^T are identical within a
// cell, so  b0 = sum_c pinv_c * sum_{n in c} log(y_n / sf_n + 0.1): a sample adds its logarithm into its cell's
// accumulator (no loads of the P pseudo-inverse rows per sample), the P entries of b0 are rebuilt entry-parallel.
// Four samples per trip, loads first (see irls_sweep_cell).  A.pinvc: [C][P] in LDS.
template <class Wv, int P>
DSQ_HD void irls_init_cell(const IrlsArgs& A, double a, double (&b0)[P], double& cst) {
    const CellDesign& D = *A.cells;
    typedef DSQ_LDS_STRUCT(CellWork<P>) LdsWork;
    LdsWork& Wk = *(LdsWork*)A.cell_ws;
    const auto pinvc_ = DSQ_AS_LDS(double, A.pinvc);
    for (int c = Wv::lane(); c < D.C; c += Wv::W) Wk.acc[0][c] = 0.0;
    Wv::sync();
    double lga, dga_unused, tab_dl, tab_dd;
    lgamma_digamma<false, true>(a, lga, dga_unused);
    lgamma_digamma_diff<Wv, false>(Wv::lane(), a, lga, 0.0, tab_dl, tab_dd);
    double csum = 0.0;
    constexpr int U = kIrlsRowU;
    for (int base = 0; base < A.N; base += U * Wv::W) {  // wave-uniform trip count (cross-lane memo reads)
        int yi[U], cell[U];
        double sfn[U], dl[U], lf[U];
        bool valid[U], in_tab[U], in_fact[U];
        bool any_big = false, any_huge = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = base + u * Wv::W + Wv::lane();
            valid[u] = n < A.N;
            yi[u] = valid[u] ? A.y[n] : 0;
            sfn[u] = valid[u] ? A.sf[n] : 1.0;
            cell[u] = valid[u] ? D.cell_of[n] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            in_tab[u] = yi[u] < Wv::W;
            in_fact[u] = yi[u] < kLgammaIntN;
            any_big = any_big || !in_tab[u];
            any_huge = any_huge || !in_fact[u];
            dl[u] = Wv::from_lane(tab_dl, in_tab[u] ? yi[u] : 0);
            lf[u] = kLgammaInt[in_fact[u] ? yi[u] : 0];
        }
        if (Wv::any(any_big)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double dl2, dd2;
                lgamma_digamma_diff<Wv, false>(in_tab[u] ? 64 : yi[u], a, lga, 0.0, dl2, dd2);
                dl[u] = in_tab[u] ? dl[u] : dl2;
            }
        }
        if (Wv::any(any_huge)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double z = in_fact[u] ? 300.0 : (double)yi[u] + 1.0;
                const double big = (z - 0.5) * flog_t(z) - z + kHalfLog2Pi + stirling_tail_big(frcp(z));
                lf[u] = in_fact[u] ? lf[u] : big;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            csum -= valid[u] ? dl[u] + lf[u] : 0.0;
            const double ly = flog_t((double)yi[u] * frcp(sfn[u]) + 0.1);
            if (valid[u]) Wv::cell_add(&Wk.acc[0][cell[u]], ly);
        }
    }
    cst = Wv::sum(csum);
    Wv::sync();
    if (Wv::lane() < P) {
        double v = 0.0;
        for (int c = 0; c < D.C; ++c) v += pinvc_[c * P + Wv::lane()] * Wk.acc[0][c];
        Wk.ent[Wv::lane()] = v;
    }
    Wv::sync();
#pragma unroll
    for (int j = 0; j < P; ++j) b0[j] = Wk.ent[j];
    Wv::sync();  // ent and acc are rewritten by the first sweep
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
    DSQ_UNROLL2
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

// hat diagonal (:427-433) from M = X^T W X at the final clamped mu, and unclamped mu (:435-437); with an
// epilogue also the Cook's bookkeeping and the Wald statistics of the gene (LfcEpilogue)
template <class Wv, int P>
DSQ_HD void irls_finish(const IrlsArgs& A, const double (&beta)[P], double (&M)[Tri<P>::N],
                        double* mu_out, double* H_out, LfcEpilogue* E = nullptr) {
    constexpr int T = Tri<P>::N;
    const bool want_cooks = E != nullptr && E->flags != nullptr;
    const bool want_wald = E != nullptr && E->ridge != nullptr;
    if (mu_out == nullptr && H_out == nullptr && !want_cooks && !want_wald) return;
    // few design columns: the Wald matrix (X^T W X at the UNclamped mu, ds.py:320-324) is accumulated in the
    // same pass; wider designs take a second pass so that inv[] and the accumulators are not live together
    constexpr bool kWaldInLoop = P <= 4;
    double rinv[P];
    if (H_out != nullptr || want_cooks) {
#pragma unroll
        for (int j = 0; j < P; ++j) M[tri(j, j)] += 1e-6;
        chol<P>(M);
        chol_rinv<P>(M, rinv);
    }
    CooksAcc<Wv> acc(want_cooks ? E->robust_disp : 0.0, want_cooks ? E->cutoff : 0.0, P);
    double Mw[T];
#pragma unroll
    for (int k = 0; k < T; ++k) Mw[k] = 0.0;
    if (mu_out != nullptr || H_out != nullptr || want_cooks || (want_wald && kWaldInLoop)) {
        DSQ_UNROLL2
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
            double x[P];
            double eta = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * beta[j]; }
            const double mu_raw = A.sf[n] * exp(eta);
            if (mu_out != nullptr) mu_out[n] = mu_raw;
            if (H_out != nullptr || want_cooks) {
                const double mu = dmax(mu_raw, A.min_mu);
                const double w = mu * frcp_g(1.0 + mu * A.disp);
                const double h = w * chol_quad<P>(M, rinv, x);  // (sqrt(w) q sqrt(w) of utils.py:430-433, to an ulp)
                if (H_out != nullptr) H_out[n] = h;
                if (want_cooks) {
                    const double ck = acc.add(n, (double)A.y[n], mu_raw, h, E->flags[n]);
                    if (E->cooks_row != nullptr) E->cooks_row[n] = ck;
                }
            }
            if (want_wald && kWaldInLoop) {
                const double wu = mu_raw * frcp_g(1.0 + mu_raw * A.disp);
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const double xw = x[i] * wu;
#pragma unroll
                    for (int j = 0; j <= i; ++j) Mw[tri(i, j)] += xw * x[j];
                }
            }
        }
    }
    if (want_cooks) E->cooks = acc.finish(A.y, A.N);
    if (want_wald) {
        if (!kWaldInLoop) {
            DSQ_UNROLL2
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
                double x[P];
                double eta = 0.0;
#pragma unroll
                for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * beta[j]; }
                const double m = A.sf[n] * exp(eta);
                const double wu = m * frcp_g(1.0 + m * A.disp);
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const double xw = x[i] * wu;
#pragma unroll
                    for (int j = 0; j <= i; ++j) Mw[tri(i, j)] += xw * x[j];
                }
            }
        }
        Wv::template sum_n<T>(Mw);
        E->wald = wald_from_M<P>(Mw, beta, E->ridge, E->contrast, E->lfc_null, E->alt);
    }
}

// The same for a cell design: the table of exp(x_c . beta) left by the last sweep (it ran at the final beta)
// gives mu, q_c = x_c^T (X^T W X + ridge)^-1 x_c per cell gives the hat diagonal h_n = w_n q_c, and the Wald
// matrix is rebuilt from the cells' sums of the unclamped weights.
template <class Wv, int P>
DSQ_HD void irls_finish_cell(const IrlsArgs& A, const double (&beta)[P], double (&M)[Tri<P>::N],
                             double* mu_out, double* H_out, LfcEpilogue* E = nullptr) {
    constexpr int T = Tri<P>::N;
    const bool want_cooks = E != nullptr && E->flags != nullptr;
    const bool want_wald = E != nullptr && E->ridge != nullptr;
    if (mu_out == nullptr && H_out == nullptr && !want_cooks && !want_wald) return;
    const CellDesign& D = *A.cells;
    typedef DSQ_LDS_STRUCT(CellWork<P>) LdsWork;  // ds_read / ds_write instead of flat accesses
    LdsWork& Wk = *(LdsWork*)A.cell_ws;
    const auto Xc_ = DSQ_AS_LDS(double, D.Xc);
    const auto XX_ = DSQ_AS_LDS(double, D.XX);
    {
        double rinv[P];
#pragma unroll
        for (int j = 0; j < P; ++j) M[tri(j, j)] += 1e-6;
        chol<P>(M);
        chol_rinv<P>(M, rinv);
        for (int c = Wv::lane(); c < D.C; c += Wv::W) {
            double x[P];
#pragma unroll
            for (int j = 0; j < P; ++j) x[j] = Xc_[c * P + j];
            Wk.acc[0][c] = chol_quad<P>(M, rinv, x);  // read-only from here on
            Wk.acc[1][c] = 0.0;                  // sums of the unclamped weights (Wald)
        }
    }
    Wv::sync();
    CooksAcc<Wv> acc(want_cooks ? E->robust_disp : 0.0, want_cooks ? E->cutoff : 0.0, P);
    // one sample; returns its unclamped Wald weight
    auto sample = [&](int n, int cell, double sfn, double e, double q, double yv, int fl) {
        const double mu_raw = sfn * e;
        if (mu_out != nullptr) mu_out[n] = mu_raw;
        double w = 0.0;
        const bool have_w = H_out != nullptr || want_cooks;
        if (have_w) {
            const double mu = dmax(mu_raw, A.min_mu);
            w = mu * frcp_g(1.0 + mu * A.disp);
            const double h = w * q;  // (sqrt(w) q sqrt(w) of utils.py:430-433, to an ulp)
            if (H_out != nullptr) H_out[n] = h;
            if (want_cooks) {
                const double ck = acc.add(n, yv, mu_raw, h, fl);
                if (E->cooks_row != nullptr) E->cooks_row[n] = ck;
            }
        }
        // (the same number unless a lane was clamped)
        if (!want_wald || (have_w && !Wv::any(!(mu_raw >= A.min_mu)))) return w;
        return mu_raw * frcp_g(1.0 + mu_raw * A.disp);
    };
    int n = Wv::lane();
    if constexpr (Wv::W == 16) {  // four samples per trip, loads first (see irls_sweep_cell)
        constexpr int U = kIrlsRowU;
        for (; n + (U - 1) * Wv::W < A.N; n += U * Wv::W) {
            int cell[U], fl[U];
            double sfn[U], e[U], q[U], yv[U], wu[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = n + u * Wv::W;
                cell[u] = D.cell_of[m];
                sfn[u] = A.sf[m];
                yv[u] = want_cooks ? (double)A.y[m] : 0.0;
                fl[u] = want_cooks ? E->flags[m] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { e[u] = Wk.tab[1][cell[u]]; q[u] = Wk.acc[0][cell[u]]; }
#pragma unroll
            for (int u = 0; u < U; ++u) wu[u] = sample(n + u * Wv::W, cell[u], sfn[u], e[u], q[u], yv[u], fl[u]);
            if (want_wald) {
#pragma unroll
                for (int u = 0; u < U; ++u) Wv::cell_add(&Wk.acc[1][cell[u]], wu[u]);
            }
        }
    }
    for (; n < A.N; n += Wv::W) {
        const int cell = D.cell_of[n];
        const double wu = sample(n, cell, A.sf[n], Wk.tab[1][cell], Wk.acc[0][cell],
                                 want_cooks ? (double)A.y[n] : 0.0, want_cooks ? E->flags[n] : 0);
        if (want_wald) Wv::cell_add(&Wk.acc[1][cell], wu);
    }
    if (want_cooks) E->cooks = acc.finish(A.y, A.N);
    if (want_wald) {
        Wv::sync();
        for (int e = Wv::lane(); e < T; e += Wv::W) {
            Wk.ent[e] = cell_dot<T>(XX_ + e, Wk.acc[1], D.C);
        }
        Wv::sync();
        double Mw[T];
#pragma unroll
        for (int k = 0; k < T; ++k) Mw[k] = Wk.ent[k];
        E->wald = wald_from_M<P>(Mw, beta, E->ridge, E->contrast, E->lfc_null, E->alt);
    }
}

// beta (out), mu_out[N] = UNclamped sf*exp(X beta), H_out[N] hat diagonal (either may be null).
// When IRLS diverges nothing is written and out.fallback = 1.
// CELL: 0 general design, 1 per-cell sums in LDS (5 .. 64 cells), 2 per-cell sums in registers (<= kSmallCells),
// 3 the same for at most two cells (the two-group comparison: half the selects and accumulators per sample)
template <class Wv, int P, int CELL = 0>
DSQ_HD IrlsOut irls_gene(const IrlsArgs& A, double (&beta)[P], double* mu_out, double* H_out,
                         LfcEpilogue* E = nullptr) {
    constexpr int T = Tri<P>::N;
    IrlsOut out;
    out.converged = 1; out.iters = 0; out.fallback = 0;
    const double a = 1.0 / A.disp;
    double cst;
    DSQ_PHASE(1);
    if constexpr (CELL == 1 && Wv::W == 16) {
        if (A.pinvc != nullptr) irls_init_cell<Wv, P>(A, a, beta, cst);
        else irls_init<Wv, P>(A, a, beta, cst);
    } else {
        irls_init<Wv, P>(A, a, beta, cst);
    }
    const double nlogterm = A.N * a * log(A.disp);
    double M[T], r[P], S;
    constexpr int CS = CELL == 3 ? 2 : kSmallCells;
    double e_c[CS];  // CELL == 2, 3: exp(x_c . beta) of the last sweep
    // sixteen-lane rows solve the normal equations cooperatively (row_chol_solve: two registers per design column and
    // lane instead of the whole matrix in every lane)
    constexpr bool kRowSolve = CELL == 1 && Wv::W == 16;
    auto sweep = [&]() {
        DSQ_PHASE(2);
        if constexpr (CELL == 1) irls_sweep_cell<Wv, P, !kRowSolve>(A, beta, a, S, M, r);
        else if constexpr (CELL >= 2) irls_sweep_cs<Wv, P, CS>(A, beta, a, S, M, r, e_c);
        else irls_sweep<Wv, P>(A, beta, a, S, M, r);
        DSQ_PHASE(5);
    };
    sweep();
    double dev = 1000.0, ratio = 1.0;
    int i = 0;
    while (ratio > A.beta_tol) {
        double bh[P];
        if constexpr (kRowSolve) {
            typedef DSQ_LDS_STRUCT(CellWork<P>) LdsWork;
            DSQ_PHASE(7);
            row_chol_solve<Wv, P>(((LdsWork*)A.cell_ws)->ent, 1e-6, bh);
            Wv::sync();  // ent is rewritten by the next sweep
        } else {
            double Hm[T];
#pragma unroll
            for (int k = 0; k < T; ++k) Hm[k] = M[k];
#pragma unroll
            for (int j = 0; j < P; ++j) Hm[tri(j, j)] += 1e-6;
            DSQ_PHASE(7);
            chol<P>(Hm);
            DSQ_PHASE(8);
#pragma unroll
            for (int j = 0; j < P; ++j) bh[j] = r[j];
            chol_solve<P>(Hm, bh);
        }
        DSQ_PHASE(5);
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
        sweep();
        const double old = dev;
        dev = -2.0 * (nlogterm - cst + S);
        ratio = fabs(dev - old) / (fabs(dev) + 0.1);
    }
    out.iters = i;
    DSQ_PHASE(6);
    if constexpr (kRowSolve) {  // X^T W X of the last sweep
        typedef DSQ_LDS_STRUCT(CellWork<P>) LdsWork;
        LdsWork& Wk = *(LdsWork*)A.cell_ws;
#pragma unroll
        for (int k = 0; k < T; ++k) M[k] = Wk.ent[k];
        Wv::sync();  // (the finish reuses ent)
    }
    if constexpr (CELL == 1) irls_finish_cell<Wv, P>(A, beta, M, mu_out, H_out, E);
    else if constexpr (CELL >= 2) irls_finish_cs<Wv, P, CS>(A, beta, M, e_c, mu_out, H_out, E);
    else irls_finish<Wv, P>(A, beta, M, mu_out, H_out, E);
    return out;
}

// Workspace of the rescue (wave-private LDS on the device).
template <int P>
struct IrlsRescueWork {
    union {
        LbfgsbWork<P> lb;
        BfgsWork<P> bf;  // optimizer="BFGS"
    };
    double x[P], l[P], u[P];
    int nbd[P];
};

// grid_fit_beta (grid_search.py:145-221): two-level 60 x 60 grid on [min_beta, max_beta]^2, P == 2.
template <class Wv>
DSQ_HD void grid_fit_beta2(const IrlsArgs& A, double a, double cst, double (&beta)[2],
                           int grid_length = 60) {
    auto loss = [&](double bx, double by) {
        double s = 0.0;
        DSQ_UNROLL2
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
                                double* mu_out, double* H_out, LfcEpilogue* E = nullptr, int optimizer = 0) {
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
        DSQ_UNROLL2
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
    bool ok;
    if (optimizer == 1) {  // scipy's unbounded BFGS (utils.py:389-399 with optimizer="BFGS")
        const BfgsResult rb = bfgs_min<P>(fg, P, Wk.x, Wk.bf);
        ok = rb.success;
        out.iters = rb.nit;
    } else {
        const LbfgsbResult res = lbfgsb_nd<P>(fg, P, Wk.x, Wk.l, Wk.u, Wk.nbd, Wk.lb);
        ok = res.success;
        out.iters = res.nit;
    }
#pragma unroll
    for (int j = 0; j < P; ++j) beta[j] = Wk.x[j];
    out.converged = ok ? 1 : 0;
    if (!ok && P <= 2) {
        if constexpr (P == 2) grid_fit_beta2<Wv>(A, a, cst, beta);
    }
    double M[T], r[P], S2;
    irls_sweep<Wv, P>(A, beta, a, S2, M, r);  // M = X^T W X at the final (clamped) mu
    irls_finish<Wv, P>(A, beta, M, mu_out, H_out, E);
    return out;
}

}  // namespace dsq
