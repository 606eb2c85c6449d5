// dsq_shrink.h — apeGLM MAP log-fold-change per gene, one gene per wave.
//
// Replaces pydeseq2/utils.py:990-1142 (nbinomGLM), :1145-1207 (nbinomFn) and
// pydeseq2/grid_search.py:224-320 (grid_fit_shrink_beta):
//   f(beta)   = [ prior(beta) - nll(beta) ] / cnst,          cnst = max(f_raw(0), 1)
//   nll       = sum_n  y_n eta_n - (y_n + size) logaddexp(eta_n + offset_n, log size),   eta = X beta
//   prior     = sum_{j != s} beta_j^2 / (2 sigma0^2) + log1p((beta_s / sigma)^2)
//   grad      = [ beta_j / sigma0^2 (j != s),  2 beta_s / (sigma^2 + beta_s^2) (j == s) ]
//               - sum_n [ y_n - (y_n + size) / (1 + size exp(-eta_n - offset_n)) ] x_n        , / cnst
// minimised with the unbounded L-BFGS-B (scipy options ftol = gtol = 1e-8) from
// beta_0 = 0.1 (-1)^j; on failure and p == 2 the reference's two-level 60 x 60 grid; then the inverse
// of the Hessian  X^T diag(frac) X + diag(h)  at the solution (unscaled), frac = (y + size) size e /
// (size + e)^2,  e = exp(eta + offset).
#pragma once
#include "dsq_alpha.h"  // linspace_at
#include "dsq_lbfgsb.h"
#include "dsq_lbfgsb_dense.h"
#include "dsq_lbfgsb_wave.h"
#include "dsq_bfgs.h"
#include "dsq_linalg.h"
#include "dsq_wave.h"

namespace dsq {

struct ShrinkArgs {
    const int32_t* y;      // [N]
    const double* offset;  // [N] log size factors
    const double* Xt;      // [P][ldx]
    int ldx, N;
    double size;           // 1 / dispersion
    double sigma0, sigma;  // prior_no_shrink_scale, prior_scale
    int shrink_index;
};

// up to 4 coefficients the dense-matrix L-BFGS-B (same iterates, a fraction of the scalar work and of
// the workspace) replaces the compact-form one
constexpr int kShrinkDenseMax = 4;
template <int P, bool DENSE = (P <= kShrinkDenseMax)>
struct ShrinkLb { typedef LbfgsbWork<P> type; };
template <int P>
struct ShrinkLb<P, true> { typedef LbfgsbDenseWork<P> type; };

template <int P, bool WAVE8 = false>
struct ShrinkWork {  // wave-private LDS on the device
    typename ShrinkLb<P>::type lb;
    double x[P], l[P], u[P];
    int nbd[P];
};
// optimizer = "BFGS" / "Newton-CG" (utils.py:1112-1121 hands the name to scipy.optimize.minimize): dsq_bfgs.h
template <int P>
struct ShrinkWorkAlt {
    NewtonCgWork<P> opt;
    double x[P];
};
template <class T>
struct IsAltWork { static constexpr bool value = false; };
template <int P>
struct IsAltWork<ShrinkWorkAlt<P>> { static constexpr bool value = true; };
// 5 ... 12 coefficients on the device: the inverse quasi-Newton matrix in the wavefront's registers (dsq_lbfgsb_wave.h:
// 8 x 8 with one entry per lane up to 8 coefficients, 16 x 16 with four per lane beyond); the host instantiation
// (tests/hostsim) keeps the compact form
constexpr int shrink_wave_rank(int p) { return p <= 8 ? 8 : 16; }
template <int P>
struct ShrinkWork<P, true> {
    LbfgsbWaveWorkT<shrink_wave_rank(P)> lb;
};
constexpr bool shrink_on_wave8(int p) { return p > kShrinkDenseMax && p <= 12; }
template <class T>
struct IsWave8Work { static constexpr bool value = false; };
template <int P>
struct IsWave8Work<ShrinkWork<P, true>> { static constexpr bool value = true; };

// prior - nll (unscaled) and, if g != nullptr, its gradient.
// With d = eta + offset - log(size) and e = exp(-|d|) (one exponential per sample):
//   logaddexp(eta + offset, log size) = max(.) + log1p(e)                      (numpy's formula)
//   (y + size) / (1 + size exp(-eta - offset)) = (y + size) / (1 + exp(-d)) = (y + size) * (d > 0 ? 1 : e) / (1 + e)
// so the likelihood and its gradient share the exponential, one lean log1p and one reciprocal.
template <class Wv, int P>
DSQ_HD double shrink_fn(const ShrinkArgs& A, const double (&b)[P], double* g) {
    const double lsz = log(A.size);
    double s = 0.0, gr[P];
#pragma unroll
    for (int j = 0; j < P; ++j) gr[j] = 0.0;
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        double x[P];
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * b[j]; }
        const double eo = eta + A.offset[n];
        const double d = eo - lsz;
        const double e = exp(-fabs(d));
        const double lae = (d > 0 ? eo : lsz) + flog1p(e);
        s += yv * eta - (yv + A.size) * lae;
        if (g != nullptr) {
            const double gk = yv - (yv + A.size) * ((d > 0 ? 1.0 : e) * frcp(1.0 + e));
#pragma unroll
            for (int j = 0; j < P; ++j) gr[j] += gk * x[j];
        }
    }
    s = Wv::sum(s);
    double prior = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j)
        if (j != A.shrink_index) prior += (b[j] * b[j]) / (2.0 * A.sigma0 * A.sigma0);
    double bs = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) bs = (j == A.shrink_index) ? b[j] : bs;
    const double q = bs / A.sigma;
    prior += log1p(q * q);
    if (g != nullptr) {
        Wv::template sum_n<P>(gr);
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double dp = (j == A.shrink_index) ? 2.0 * b[j] / (A.sigma * A.sigma + bs * bs)
                                                    : b[j] / (A.sigma0 * A.sigma0);
            g[j] = dp - gr[j];
        }
    }
    return prior - s;
}

// grid_fit_shrink_beta (grid_search.py:224-320), P == 2 (the reference's loss uses shrink_index = 1)
template <class Wv>
DSQ_HD void grid_fit_shrink2(const ShrinkArgs& A0, double cnst, double (&beta)[2], int grid_length = 60,
                             double min_beta = -30.0, double max_beta = 30.0) {
    ShrinkArgs A = A0;
    A.shrink_index = 1;
    double xlo = min_beta, xhi = max_beta, ylo = min_beta, yhi = max_beta;
    for (int level = 0; level < 2; ++level) {
        double best = 0.0;
        int bi = 0, bj = 0;
        bool best_nan = false, first = true;
        for (int i = 0; i < grid_length; ++i) {
            for (int j = 0; j < grid_length; ++j) {
                const double bb[2] = {linspace_at(xlo, xhi, grid_length, i), linspace_at(ylo, yhi, grid_length, j)};
                const double v = shrink_fn<Wv, 2>(A, bb, nullptr) / cnst;
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

// beta[P] (out), inv_hessian[P*P] row-major (out); returns scipy's res.success
// ih_entry (nullable): only inv_hessian[shrink_index][shrink_index] - all DeseqStats.lfc_shrink uses (ds.py:424-433)
// optimizer: 0 "L-BFGS-B" (what ds.py:407 passes), 1 "BFGS", 2 "Newton-CG" - the latter two with a ShrinkWorkAlt workspace
template <class Wv, int P, class Work>
DSQ_HD int shrink_gene(const ShrinkArgs& A, Work& Wk, double (&beta)[P], double* inv_hessian,
                       double* ih_entry = nullptr, int optimizer = 0) {
    constexpr int T = Tri<P>::N;
    double zero[P];
#pragma unroll
    for (int j = 0; j < P; ++j) zero[j] = 0.0;
    const double f0 = shrink_fn<Wv, P>(A, zero, nullptr);
    const double cnst = f0 > 1.0 ? f0 : 1.0;  // np.maximum(scale_cnst, 1): NaN propagates like numpy
    const double cn = (f0 != f0) ? f0 : cnst;
    if constexpr (IsAltWork<Work>::value) {
#pragma unroll
        for (int j = 0; j < P; ++j) Wk.x[j] = (j & 1) ? -0.1 : 0.1;
    } else if constexpr (!IsWave8Work<Work>::value) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            Wk.x[j] = (j & 1) ? -0.1 : 0.1;
            Wk.l[j] = 0.0; Wk.u[j] = 0.0; Wk.nbd[j] = 0;  // unbounded
        }
    }
    auto fg = [&](const double* xb, double& f, double* g) {
        double b[P], gg[P];
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] = xb[j];
        f = shrink_fn<Wv, P>(A, b, gg) / cn;
#pragma unroll
        for (int j = 0; j < P; ++j) g[j] = gg[j] / cn;
    };
    // the reference's Hessian of the scaled objective at b: (X^T F X + h broadcast over the rows) / cnst (utils.py:1091-1110;
    // the broadcasting quirk is described below) - what Newton-CG's conjugate gradients multiply with
    auto hess = [&](const double* xb, double* Hm) {
        double b[P], Mh[T];
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] = xb[j];
#pragma unroll
        for (int k = 0; k < T; ++k) Mh[k] = 0.0;
        for (int n = Wv::lane(); n < A.N; n += Wv::W) {
            const double yv = (double)A.y[n];
            double x[P];
            double eta = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * b[j]; }
            const double e = exp(eta + A.offset[n]);
            const double fr = (yv + A.size) * A.size * e / ((A.size + e) * (A.size + e));
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const double xw = x[i] * fr;
#pragma unroll
                for (int j = 0; j <= i; ++j) Mh[tri(i, j)] += xw * x[j];
            }
        }
        Wv::template sum_n<T>(Mh);
        double bsh = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) bsh = (j == A.shrink_index) ? b[j] : bsh;
        const double s2 = A.sigma * A.sigma, b2 = bsh * bsh;
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const double hdj = (j == A.shrink_index) ? 2.0 * (s2 - b2) / ((s2 + b2) * (s2 + b2))
                                                         : 1.0 / (A.sigma0 * A.sigma0);
                Hm[i * P + j] = (Mh[tri(i > j ? i : j, i > j ? j : i)] + hdj) / cn;
            }
    };
    // scipy: factr = ftol / eps, pgtol = gtol
    LbfgsbResult res;
    if constexpr (IsAltWork<Work>::value) {
        // (the options the reference passes are {"ftol": 1e-8, "gtol": 1e-8}: BFGS knows gtol, Newton-CG neither)
        const BfgsResult rb = optimizer == 1 ? bfgs_min<P>(fg, P, Wk.x, Wk.opt, 1e-8)
                                             : newton_cg_min<P>(fg, hess, P, Wk.x, Wk.opt);
        res.success = rb.success;
#pragma unroll
        for (int j = 0; j < P; ++j) beta[j] = Wk.x[j];
    } else if constexpr (IsWave8Work<Work>::value) {
        if (Wv::lane() < P) Wk.lb.x[Wv::lane()] = (Wv::lane() & 1) ? -0.1 : 0.1;
        res = lbfgsb_wave<P, shrink_wave_rank(P)>(fg, Wk.lb, 1e-8 / 2.220446049250313e-16, 1e-8);
#pragma unroll
        for (int j = 0; j < P; ++j) beta[j] = Wk.lb.x[j];
    } else {
        if constexpr (P <= kShrinkDenseMax)
            res = lbfgsb_dense<P>(fg, P, Wk.x, Wk.l, Wk.u, Wk.nbd, Wk.lb, 1e-8 / 2.220446049250313e-16, 1e-8);
        else
            // (the linear algebra between two evaluations spread over the wavefront's lanes: dsq_lbfgsb_par.h - same
            // iterates)
            res = lbfgsb_nd<P, decltype(fg)&, 10, Wv>(fg, P, Wk.x, Wk.l, Wk.u, Wk.nbd, Wk.lb,
                                                      1e-8 / 2.220446049250313e-16, 1e-8);
#pragma unroll
        for (int j = 0; j < P; ++j) beta[j] = Wk.x[j];
    }
    if (!res.success && P == 2) {
        if constexpr (P == 2) grid_fit_shrink2<Wv>(A, cn, beta);
    }
    // Hessian (cnst = 1) and its inverse (numpy.linalg.inv)
    double M[T];
#pragma unroll
    for (int k = 0; k < T; ++k) M[k] = 0.0;
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        double x[P];
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) { x[j] = A.Xt[j * A.ldx + n]; eta += x[j] * beta[j]; }
        const double e = exp(eta + A.offset[n]);
        const double fr = (yv + A.size) * A.size * e / ((A.size + e) * (A.size + e));
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const double xw = x[i] * fr;
#pragma unroll
            for (int j = 0; j <= i; ++j) M[tri(i, j)] += xw * x[j];
        }
    }
    Wv::template sum_n<T>(M);
    double bs = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) bs = (j == A.shrink_index) ? beta[j] : bs;
    // prior curvature h_j.  The reference adds it as  X^T F X + np.diag(h)  where h is ALREADY a diagonal
    // matrix, so np.diag(h) is the 1-D vector of its diagonal and numpy broadcasting adds h_j to EVERY
    // row of column j (utils.py:1099-1110).  Reproduced as is: the returned "inverse Hessian" (and the
    // shrunken lfcSE derived from it, ds.py:424-433) is the inverse of that non-symmetric matrix.
    double hd[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const double s2 = A.sigma * A.sigma, b2 = bs * bs;
        hd[j] = (j == A.shrink_index) ? 2.0 * (s2 - b2) / ((s2 + b2) * (s2 + b2)) : 1.0 / (A.sigma0 * A.sigma0);
    }
    if (inv_hessian != nullptr || ih_entry != nullptr) {
        // general inverse by Gauss-Jordan with partial pivoting on the full p x p matrix
        if constexpr (Wv::W > 1) {
            // One ROW per lane (lanes >= P idle along): the wave-redundant version below indexes its two p x p arrays with
            // the run-time pivot row, which puts them into scratch memory - ~3000 scratch accesses per gene in every lane.
            // Here a lane keeps its row of [H | I] in registers, rows are never moved: `pos` is the logical position of
            // the row a lane holds, a pivot exchange swaps two positions.  Same operations on the same numbers.
            const int ln = Wv::lane();
            const int me = ln < P ? ln : P - 1;  // (idle lanes shadow the last row; they never win a pivot search)
            double hr[P], ir[P];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < P; ++i) v = (i == me) ? M[tri(i > j ? i : j, i > j ? j : i)] : v;
                hr[j] = v + hd[j];
                ir[j] = (j == me) ? 1.0 : 0.0;
            }
            int pos = ln < P ? ln : 0x7fff;
            static_for<0, P>([&](auto CC) {
                constexpr int c = decltype(CC)::value;
                // pivot: the first row at position >= c with the largest |H[., c]| (the scalar scan starts at position c and
                // replaces on strictly greater: ties go to the smaller position)
                const int src_c = Wv::maxi(pos == c ? ln : -1);
                const double a_c = fabs(Wv::from_lane(hr[c], src_c));
                const double mine = fabs(hr[c]);
                // (NaN never wins a comparison: a NaN at position c keeps the pivot there, a NaN elsewhere is skipped)
                const double av = (pos > c && pos < P && mine > a_c) ? mine : -1.0;
                const double mx = Wv::max(av);
                const int cand = (mx >= 0.0 && av == mx) ? pos : 0x7fff;
                const int best = -Wv::maxi(-cand);  // smallest position among the maxima
                const int pv = mx >= 0.0 ? best : c;
                // exchange the positions c and pv
                if (pos == c) pos = pv;
                else if (pos == pv) pos = c;
                const bool is_piv = pos == c;
                // the pivot lane scales its row, every lane fetches it
                const int src = Wv::maxi(is_piv ? ln : -1);
                const double d = 1.0 / Wv::from_lane(hr[c], src);
                double ph[P], pi[P];
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const double hk = hr[k] * d, ik = ir[k] * d;
                    if (is_piv) { hr[k] = hk; ir[k] = ik; }
                    ph[k] = Wv::from_lane(hk, src);
                    pi[k] = Wv::from_lane(ik, src);
                }
                if (!is_piv) {
                    const double fct = hr[c];
#pragma unroll
                    for (int k = 0; k < P; ++k) { hr[k] -= fct * ph[k]; ir[k] -= fct * pi[k]; }
                }
            });
            if (ln < P) {
                if (inv_hessian != nullptr)
#pragma unroll
                    for (int j = 0; j < P; ++j) inv_hessian[pos * P + j] = ir[j];
                if (ih_entry != nullptr && pos == A.shrink_index) {
                    double v = ir[0];
#pragma unroll
                    for (int j = 1; j < P; ++j) v = (j == A.shrink_index) ? ir[j] : v;
                    *ih_entry = v;
                }
            }
        } else {
        double Hm[P][P], Iv[P][P];
#pragma unroll
        for (int i = 0; i < P; ++i)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                Hm[i][j] = M[tri(i > j ? i : j, i > j ? j : i)] + hd[j];
                Iv[i][j] = (i == j) ? 1.0 : 0.0;
            }
        for (int c = 0; c < P; ++c) {
            int pv = c;
            double mx = fabs(Hm[c][c]);
            for (int r = c + 1; r < P; ++r)
                if (fabs(Hm[r][c]) > mx) { mx = fabs(Hm[r][c]); pv = r; }
            if (pv != c)
                for (int k = 0; k < P; ++k) {
                    double t = Hm[c][k]; Hm[c][k] = Hm[pv][k]; Hm[pv][k] = t;
                    t = Iv[c][k]; Iv[c][k] = Iv[pv][k]; Iv[pv][k] = t;
                }
            const double d = 1.0 / Hm[c][c];
            for (int k = 0; k < P; ++k) { Hm[c][k] *= d; Iv[c][k] *= d; }
            for (int r = 0; r < P; ++r) {
                if (r == c) continue;
                const double fct = Hm[r][c];
                for (int k = 0; k < P; ++k) { Hm[r][k] -= fct * Hm[c][k]; Iv[r][k] -= fct * Iv[c][k]; }
            }
        }
        if (Wv::lane() == 0) {
            if (inv_hessian != nullptr)
                for (int i = 0; i < P; ++i)
                    for (int j = 0; j < P; ++j) inv_hessian[i * P + j] = Iv[i][j];
            if (ih_entry != nullptr) *ih_entry = Iv[A.shrink_index][A.shrink_index];
        }
        }
    }
    return res.success ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Designs of 13 ... 32 columns (the reference has no limit, utils.py:990-1142): the same fit with a RUN-TIME number of
// coefficients p <= PMAX.  The optimiser is the same lbfgsb_nd (its workspace is sized by PMAX), the Hessian is built
// one row per pass over the samples (p passes: the p (p + 1) / 2 accumulators of the narrow path would not fit the
// register file), and the inverse runs on two p x p matrices in the wave's LDS workspace, column-parallel.  One gene per
// wavefront; not tuned - what matters here is that wide designs run at all.
template <int PMAX, bool WAVE = false>
struct ShrinkWorkWide {  // wave-private LDS on the device
    LbfgsbWork<PMAX> lb;
    double x[PMAX], l[PMAX], u[PMAX];
    int nbd[PMAX];
    double Hm[PMAX * PMAX], Iv[PMAX * PMAX];
};
// the device build: the optimiser's inverse matrix in the wavefront's registers (dsq_lbfgsb_wave.h: 16 x 16 with four
// entries per lane, 32 x 32 with sixteen); x lives in its workspace
template <int PMAX>
struct ShrinkWorkWide<PMAX, true> {
    LbfgsbWaveWorkT<PMAX> lb;
    double Hm[PMAX * PMAX], Iv[PMAX * PMAX];
};
template <class T>
struct IsWaveWideWork { static constexpr bool value = false; };
template <int PMAX>
struct IsWaveWideWork<ShrinkWorkWide<PMAX, true>> { static constexpr bool value = true; };

// PB: the column loops walk PB <= PMAX columns (a multiple of 8 >= p: round 5 - at p = 24 the 32-column walk of the first
// version spent a quarter of its per-sample work on zeros); columns [p, PB) read as zero
template <class Wv, int PMAX, int PB = PMAX>
DSQ_HD double shrink_fn_wide(const ShrinkArgs& A, int p, const double* xb, double* g) {
    static_assert(PB <= PMAX && PB % 8 == 0, "column block");
    const double lsz = log(A.size);
    double b[PB], gr[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) { b[j] = j < p ? xb[j] : 0.0; gr[j] = 0.0; }
    double s = 0.0;
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        double x[PB];
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            x[j] = j < p ? A.Xt[j * A.ldx + n] : 0.0;
            eta += x[j] * b[j];  // (x[j] b[j] = 0 exactly beyond p: the sum's value and rounding are those of p terms)
        }
        const double eo = eta + A.offset[n];
        const double d = eo - lsz;
        const double e = exp(-fabs(d));
        const double lae = (d > 0 ? eo : lsz) + flog1p(e);
        s += yv * eta - (yv + A.size) * lae;
        if (g != nullptr) {
            const double gk = yv - (yv + A.size) * ((d > 0 ? 1.0 : e) * frcp(1.0 + e));
#pragma unroll
            for (int j = 0; j < PB; ++j) gr[j] += gk * x[j];
        }
    }
    s = Wv::sum(s);
    double prior = 0.0, bs = 0.0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        if (j < p && j != A.shrink_index) prior += (b[j] * b[j]) / (2.0 * A.sigma0 * A.sigma0);
        bs = (j == A.shrink_index) ? b[j] : bs;
    }
    const double q = bs / A.sigma;
    prior += log1p(q * q);
    if (g != nullptr) {
        Wv::template sum_n<PB>(gr);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            if (j < p) {
                const double dp = (j == A.shrink_index) ? 2.0 * b[j] / (A.sigma * A.sigma + bs * bs)
                                                        : b[j] / (A.sigma0 * A.sigma0);
                g[j] = dp - gr[j];
            }
        }
    }
    return prior - s;
}

// beta[p] (out), inv_hessian[p*p] row-major (out, nullable); returns scipy's res.success
template <class Wv, int PMAX, class Work, int PB = PMAX>
DSQ_HD int shrink_gene_wide(const ShrinkArgs& A, int p, Work& Wk, double* beta, double* inv_hessian) {
    constexpr bool kWave = IsWaveWideWork<Work>::value;
    double* const xw = [&]() { if constexpr (kWave) return Wk.lb.x; else return Wk.x; }();
    for (int j = 0; j < p; ++j) xw[j] = 0.0;
    Wv::sync();
    const double f0 = shrink_fn_wide<Wv, PMAX, PB>(A, p, xw, nullptr);
    const double cnst = f0 > 1.0 ? f0 : 1.0;  // np.maximum(scale_cnst, 1): NaN propagates like numpy
    const double cn = (f0 != f0) ? f0 : cnst;
    for (int j = 0; j < PMAX; ++j) {
        if constexpr (kWave) {
            xw[j] = j < p ? ((j & 1) ? -0.1 : 0.1) : 0.0;  // (columns beyond p: zero coefficients, zero gradient)
        } else if (j < p) {
            xw[j] = (j & 1) ? -0.1 : 0.1;
            Wk.l[j] = 0.0; Wk.u[j] = 0.0; Wk.nbd[j] = 0;  // unbounded
        }
    }
    Wv::sync();
    auto fg = [&](const double* xb, double& f, double* g) {
        double gg[PB];
        f = shrink_fn_wide<Wv, PMAX, PB>(A, p, xb, gg) / cn;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (j < p) g[j] = gg[j] / cn;
    };
    LbfgsbResult res;
    if constexpr (kWave)
        res = lbfgsb_wave<PMAX, PMAX>(fg, Wk.lb, 1e-8 / 2.220446049250313e-16, 1e-8);
    else
        res = lbfgsb_nd<PMAX, decltype(fg)&, 10, Wv>(fg, p, Wk.x, Wk.l, Wk.u, Wk.nbd, Wk.lb, 1e-8 / 2.220446049250313e-16,
                                                     1e-8);
    Wv::sync();
    if (Wv::lane() == 0)
        for (int j = 0; j < p; ++j) beta[j] = xw[j];
    if (inv_hessian == nullptr) return res.success ? 1 : 0;
    // Hessian (cnst = 1), TWO rows per pass over the samples (round 5: one row per pass recomputed the linear predictor and
    // its exponential p times - at p = 24 a third of the kernel's instructions):  X^T diag(frac) X + h  with the
    // reference's broadcasting quirk (h_j is added to every row of column j, utils.py:1099-1110; see shrink_gene).  Every
    // entry keeps its own accumulator and summation order: the matrix is bit-identical to the one-row version's.
    double b[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) b[j] = j < p ? xw[j] : 0.0;
    const double bs = xw[A.shrink_index];
    const double s2 = A.sigma * A.sigma, b2 = bs * bs;
    for (int i = 0; i < p; i += 2) {
        const bool two = i + 1 < p;
        double r0[PB], r1[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) { r0[j] = 0.0; r1[j] = 0.0; }
        for (int n = Wv::lane(); n < A.N; n += Wv::W) {
            const double yv = (double)A.y[n];
            double x[PB];
            double eta = 0.0;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                x[j] = j < p ? A.Xt[j * A.ldx + n] : 0.0;
                eta += x[j] * b[j];
            }
            const double e = exp(eta + A.offset[n]);
            const double fr = (yv + A.size) * A.size * e / ((A.size + e) * (A.size + e));
            const double xw0 = A.Xt[i * A.ldx + n] * fr;
            const double xw1 = two ? A.Xt[(i + 1) * A.ldx + n] * fr : 0.0;
#pragma unroll
            for (int j = 0; j < PB; ++j) { r0[j] += xw0 * x[j]; r1[j] += xw1 * x[j]; }
        }
        Wv::template sum_n<PB>(r0);
        Wv::template sum_n<PB>(r1);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            if (j < p && Wv::lane() == 0) {
                const double hd = (j == A.shrink_index) ? 2.0 * (s2 - b2) / ((s2 + b2) * (s2 + b2))
                                                        : 1.0 / (A.sigma0 * A.sigma0);
                Wk.Hm[i * p + j] = r0[j] + hd;
                Wk.Iv[i * p + j] = (i == j) ? 1.0 : 0.0;
                if (two) {
                    Wk.Hm[(i + 1) * p + j] = r1[j] + hd;
                    Wk.Iv[(i + 1) * p + j] = (i + 1 == j) ? 1.0 : 0.0;
                }
            }
        }
    }
    Wv::sync();
    // inverse by Gauss-Jordan with partial pivoting (as shrink_gene), the row operations spread over the lanes
    for (int c = 0; c < p; ++c) {
        int pv = c;
        double mx = fabs(Wk.Hm[c * p + c]);
        for (int r = c + 1; r < p; ++r) {
            const double v = fabs(Wk.Hm[r * p + c]);
            if (v > mx) { mx = v; pv = r; }
        }
        if (pv != c) {
            for (int k = Wv::lane(); k < p; k += Wv::W) {
                double t = Wk.Hm[c * p + k]; Wk.Hm[c * p + k] = Wk.Hm[pv * p + k]; Wk.Hm[pv * p + k] = t;
                t = Wk.Iv[c * p + k]; Wk.Iv[c * p + k] = Wk.Iv[pv * p + k]; Wk.Iv[pv * p + k] = t;
            }
            Wv::sync();
        }
        const double d = 1.0 / Wk.Hm[c * p + c];
        Wv::sync();  // (every lane has read the pivot before its column's owner scales it)
        for (int k = Wv::lane(); k < p; k += Wv::W) { Wk.Hm[c * p + k] *= d; Wk.Iv[c * p + k] *= d; }
        Wv::sync();
        for (int r = 0; r < p; ++r) {
            if (r == c) continue;
            const double fct = Wk.Hm[r * p + c];
            Wv::sync();  // (read by every lane before the owner of column c overwrites it)
            for (int k = Wv::lane(); k < p; k += Wv::W) {
                Wk.Hm[r * p + k] -= fct * Wk.Hm[c * p + k];
                Wk.Iv[r * p + k] -= fct * Wk.Iv[c * p + k];
            }
        }
        Wv::sync();
    }
    for (int k = Wv::lane(); k < p * p; k += Wv::W) inv_hessian[k] = Wk.Iv[k];
    return res.success ? 1 : 0;
}

}  // namespace dsq
