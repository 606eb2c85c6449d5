// dsq_wide.h — the per-gene fits for ANY design width up to kWideMaxP columns (run-time P).
//
// The register path (dsq_alpha.h / dsq_irls.h) keeps X^T W X in p(p+1)/2 registers per lane: fine up to a
// handful of columns, spilling beyond, impossible past 12.  Here the p x p matrices live in a wave-private LDS
// segment and are built by the matrix cores:
//   * Gram matrices  X^T diag(w) X  (and X^T diag(dw) X): per 64-sample chunk the lanes first do the per-sample
//     scalar work (lane = sample: log1p, lgamma differences, weights), leave w / dw in LDS, stage the chunk of the
//     design (P x 64) in LDS, and then issue v_mfma_f64_16x16x4_f64 on 16 x 16 tiles: A = (x_i w) [16 rows i, 4
//     samples], B = x_j [4 samples, 16 columns j], 16 k-steps per chunk, one accumulator fragment (4 doubles per
//     lane) per lower-triangle tile.  For P <= 8 the rows of X w and of X dw share ONE 16-row A operand, so a
//     single MFMA per k-step yields both matrices.  The fp64 matrix pipe of gfx950 has the vector FMA's flop rate
//     (tools/mfma_f64_probe.hip), so the point is not flops: the accumulators take 8 VGPRs per tile instead of
//     p(p+1) per lane, nothing spills, and the matrix pipe runs beside the vector pipe of the other resident waves.
//     (Designs with few distinct rows skip this: per-cell sums, entry-parallel rebuild — dsq_linalg.h CellDesign.)
//   * Cholesky / log-det / inverse / solves: lane-parallel over rows or columns of the LDS matrices (leading
//     dimension P|1: conflict-free column walks).
// Same formulas, same optimisers as the register path (alpha_eval, irls_gene: cited there); replaces
// pydeseq2/utils.py:273-438, 441-564 and 718-811 for designs of any width (the reference has no limit).
#pragma once
#include "dsq_alpha.h"
#include "dsq_irls.h"
#include "dsq_lbfgsb.h"

namespace dsq {

constexpr int kWideMaxP = 48;  // (round 6: 32 -> 48; 5 p x p matrices of a gene at p = 48 are 92 KB of the 160 KB of LDS)
constexpr int kWideXsLd = 65;  // leading dimension of the staged design chunk xs[j][n] (odd: conflict-free)

DSQ_HD int wide_ld(int P) { return P | 1; }
// doubles of wave-private workspace for a design of P columns
DSQ_HD int wide_work_doubles(int P) {
    const int rows = ((P + 15) / 16) * 16;  // xs is zero-padded to whole 16-row tiles
    return 5 * P * wide_ld(P) + rows * kWideXsLd + 2 * 64 + 8 * kWideMaxP + 4 * kMaxCells;
}

struct WideWork {
    int P, ld, rows;
    double *M, *dM, *L, *Li, *inv;  // P x P, leading dimension ld
    double* xs;                     // [rows][kWideXsLd] chunk of the design (rows >= P are zero)
    double* w;                      // [2][64] per-sample weights of the chunk
    double* vec;                    // [8][kWideMaxP] small vectors (beta, rhs, ...)
    double *acc, *tab;              // [2][kMaxCells] each: cell path
    DSQ_HD void bind(double* base, int P_) {
        P = P_; ld = wide_ld(P_); rows = ((P_ + 15) / 16) * 16;
        const int m = P * ld;
        M = base; dM = M + m; L = dM + m; Li = L + m; inv = Li + m;
        xs = inv + m;
        w = xs + rows * kWideXsLd;
        vec = w + 2 * 64;
        acc = vec + 8 * kWideMaxP;
        tab = acc + 2 * kMaxCells;
    }
    DSQ_HD double* v(int k) const { return vec + k * kWideMaxP; }
};

// ------------------------------------------------------------------ LDS linear algebra (lane-parallel)
// L = chol(A + diag_add I): row-parallel right-looking factorisation, same operation order per entry as chol<P>
template <class Wv>
DSQ_HD void wide_chol(const WideWork& W, const double* A, double* L, double diag_add) {
    const int P = W.P, ld = W.ld;
    for (int j = 0; j < P; ++j) {
        double d = A[j * ld + j] + diag_add;
        for (int k = 0; k < j; ++k) d -= L[j * ld + k] * L[j * ld + k];
        d = sqrt(d);
        const double r = 1.0 / d;
        for (int i = j + 1 + Wv::lane(); i < P; i += Wv::W) {
            double s = A[i * ld + j];
            for (int k = 0; k < j; ++k) s -= L[i * ld + k] * L[j * ld + k];
            L[i * ld + j] = s * r;
        }
        Wv::sync();
        if (Wv::lane() == 0) L[j * ld + j] = d;
        Wv::sync();
    }
}

template <class Wv>
DSQ_HD double wide_logdet(const WideWork& W, const double* L) {
    double s = 0.0;
    for (int j = 0; j < W.P; ++j) s += log(L[j * W.ld + j]);
    return 2.0 * s;
}

// solve (L L^T) x = b in place (b: P doubles in LDS)
template <class Wv>
DSQ_HD void wide_chol_solve(const WideWork& W, const double* L, double* b) {
    const int P = W.P, ld = W.ld;
    for (int i = 0; i < P; ++i) {  // forward, column oriented
        Wv::sync();
        const double xi = b[i] / L[i * ld + i];
        Wv::sync();
        if (Wv::lane() == 0) b[i] = xi;
        for (int k = i + 1 + Wv::lane(); k < P; k += Wv::W) b[k] -= L[k * ld + i] * xi;
    }
    for (int i = P - 1; i >= 0; --i) {  // backward
        Wv::sync();
        const double xi = b[i] / L[i * ld + i];
        Wv::sync();
        if (Wv::lane() == 0) b[i] = xi;
        for (int k = Wv::lane(); k < i; k += Wv::W) b[k] -= L[i * ld + k] * xi;
    }
    Wv::sync();
}

// inv = (L L^T)^-1 (full symmetric), Li = L^-1 as scratch
template <class Wv>
DSQ_HD void wide_inverse(const WideWork& W, const double* L, double* Li, double* inv) {
    const int P = W.P, ld = W.ld;
    for (int j = Wv::lane(); j < P; j += Wv::W) {  // column j of L^-1
        Li[j * ld + j] = 1.0 / L[j * ld + j];
        for (int i = j + 1; i < P; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s -= L[i * ld + k] * Li[k * ld + j];
            Li[i * ld + j] = s / L[i * ld + i];
        }
    }
    Wv::sync();
    for (int i = Wv::lane(); i < P; i += Wv::W) {  // row i of Li^T Li
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int k = i; k < P; ++k) s += Li[k * ld + i] * Li[k * ld + j];
            inv[i * ld + j] = s;
            inv[j * ld + i] = s;
        }
    }
    Wv::sync();
}

// sum_ij A_ij B_ij over the full symmetric matrices
template <class Wv>
DSQ_HD double wide_frob(const WideWork& W, const double* A, const double* B) {
    const int P = W.P, ld = W.ld;
    double s = 0.0;
    for (int i = Wv::lane(); i < P; i += Wv::W) {
        double r = 0.0;
        for (int j = 0; j < i; ++j) r += 2.0 * A[i * ld + j] * B[i * ld + j];
        s += r + A[i * ld + i] * B[i * ld + i];
    }
    return Wv::sum(s);
}

// q = x^T A x for the column `col` of the staged chunk xs (x_j = xs[j][col])
DSQ_HD double wide_quad_xs(const WideWork& W, const double* A, int col) {
    const int P = W.P, ld = W.ld;
    double s = 0.0;
    for (int i = 0; i < P; ++i) {
        double r = 0.0;
        for (int j = 0; j < P; ++j) r += A[i * ld + j] * W.xs[j * kWideXsLd + col];
        s += r * W.xs[i * kWideXsLd + col];
    }
    return s;
}

// ------------------------------------------------------------------ Gram matrices
// stage the design chunk of samples n0 .. n0+63 (zeros beyond N) into xs[j][lane]
template <class Wv>
DSQ_HD void wide_stage_x(const WideWork& W, const double* Xt, int ldx, int N, int n0) {
    for (int l = Wv::lane(); l < 64; l += Wv::W) {
        const int n = n0 + l;
        for (int j = 0; j < W.P; ++j) W.xs[j * kWideXsLd + l] = n < N ? Xt[j * ldx + n] : 0.0;
    }
}
template <class Wv>
DSQ_HD void wide_zero_pad_rows(const WideWork& W) {
    for (int l = Wv::lane(); l < 64; l += Wv::W)
        for (int j = W.P; j < W.rows; ++j) W.xs[j * kWideXsLd + l] = 0.0;
}

// Accumulator of X^T diag(w0) X (and, TWO, X^T diag(w1) X) over chunks; w0 / w1 of the current chunk are W.w[0..63]
// / W.w[64..127].  Device: MFMA fragments in registers; host: plain sums straight into W.M / W.dM.
template <class Wv, bool TWO>
struct WideGram {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef double d4 __attribute__((ext_vector_type(4)));
    // lower-triangle tiles (0,0), (1,0), (1,1), (2,0), (2,1), (2,2) of the two matrices: up to three 16-row blocks, P <= 48
    // (P <= 8 and TWO: only f0[0], stacked)
    static constexpr int kTiles = 6;
    d4 f0[kTiles], f1[kTiles];
#endif
    DSQ_HD void begin(const WideWork& W) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int t = 0; t < kTiles; ++t) { f0[t] = d4{0.0, 0.0, 0.0, 0.0}; f1[t] = d4{0.0, 0.0, 0.0, 0.0}; }
#else
        for (int e = 0; e < W.P * W.ld; ++e) { W.M[e] = 0.0; if (TWO) W.dM[e] = 0.0; }
#endif
    }
    // the chunk's xs and w must be in LDS (Wv::sync() by the caller before and after)
    DSQ_HD void add_chunk(const WideWork& W) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
        const bool stacked = TWO && W.P <= 8;
        const int nt = W.rows / 16;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int n = 4 * s + kq;
            const double w0 = W.w[n];
            const double x_lo = W.xs[r * kWideXsLd + n];
            if (stacked) {
                // A rows 0..7 = x_i w0, rows 8..15 = x_i w1; B columns 0..7 = x_j
                const double xa = W.xs[(r & 7) * kWideXsLd + n];
                const double a = xa * (r < 8 ? w0 : W.w[64 + n]);
                f0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x_lo, f0[0], 0, 0, 0);
            } else {
                const double w1 = TWO ? W.w[64 + n] : 0.0;
                f0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_lo * w0, x_lo, f0[0], 0, 0, 0);
                if (TWO) f1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_lo * w1, x_lo, f1[0], 0, 0, 0);
                if (nt > 1) {
                    const double x_hi = W.xs[(16 + r) * kWideXsLd + n];
                    f0[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_hi * w0, x_lo, f0[1], 0, 0, 0);
                    f0[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_hi * w0, x_hi, f0[2], 0, 0, 0);
                    if (TWO) {
                        f1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_hi * w1, x_lo, f1[1], 0, 0, 0);
                        f1[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_hi * w1, x_hi, f1[2], 0, 0, 0);
                    }
                    if (nt > 2) {  // 33 ... 48 columns: the third block of rows
                        const double x_3 = W.xs[(32 + r) * kWideXsLd + n];
                        f0[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_3 * w0, x_lo, f0[3], 0, 0, 0);
                        f0[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_3 * w0, x_hi, f0[4], 0, 0, 0);
                        f0[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_3 * w0, x_3, f0[5], 0, 0, 0);
                        if (TWO) {
                            f1[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_3 * w1, x_lo, f1[3], 0, 0, 0);
                            f1[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_3 * w1, x_hi, f1[4], 0, 0, 0);
                            f1[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(x_3 * w1, x_3, f1[5], 0, 0, 0);
                        }
                    }
                }
            }
        }
#else
        const int P = W.P, ld = W.ld;
        for (int i = 0; i < P; ++i)
            for (int j = 0; j <= i; ++j) {
                double s0 = 0.0, s1 = 0.0;
                for (int n = 0; n < 64; ++n) {
                    const double xx = W.xs[i * kWideXsLd + n] * W.xs[j * kWideXsLd + n];
                    s0 += xx * W.w[n];
                    if (TWO) s1 += xx * W.w[64 + n];
                }
                W.M[i * ld + j] += s0;
                if (TWO) W.dM[i * ld + j] += s1;
            }
#endif
    }
    // write the (symmetric) results to W.M (and W.dM)
    DSQ_HD void finish(const WideWork& W) {
        const int P = W.P, ld = W.ld;
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = threadIdx.x & 63, col = lane & 15, rq = lane >> 4;
        const bool stacked = TWO && P <= 8;
        if (stacked) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rq + 4 * q;  // 0..15: rows 0..7 -> M, 8..15 -> dM
                const int i = row & 7;
                if (i < P && col < P && col <= i) {
                    double* dst = row < 8 ? W.M : W.dM;
                    dst[i * ld + col] = f0[0][q];
                    dst[col * ld + i] = f0[0][q];
                }
            }
        } else {
            const int nt = W.rows / 16;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {  // static fragment indices (no dynamic register indexing)
                const int ti = t == 0 ? 0 : (t < 3 ? 1 : 2), tj = (t == 2 || t == 4) ? 1 : (t == 5 ? 2 : 0);
                if (ti >= nt) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * ti + rq + 4 * q, j = 16 * tj + col;
                    if (i < P && j <= i) {
                        W.M[i * ld + j] = f0[t][q];
                        W.M[j * ld + i] = f0[t][q];
                        if (TWO) { W.dM[i * ld + j] = f1[t][q]; W.dM[j * ld + i] = f1[t][q]; }
                    }
                }
            }
        }
#else
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < i; ++j) {
                W.M[j * ld + i] = W.M[i * ld + j];
                if (TWO) W.dM[j * ld + i] = W.dM[i * ld + j];
            }
#endif
        Wv::sync();
    }
};

// X^T diag(.) X from per-cell sums (cell designs): entry-parallel, both triangles
template <class Wv>
DSQ_HD void wide_gram_from_cells(const WideWork& W, const CellDesign& D, const double* cell_sum, double* Mout) {
    const int P = W.P, ld = W.ld;
    for (int e = Wv::lane(); e < P * P; e += Wv::W) {
        const int i = e / P, j = e % P;
        if (j > i) continue;
        double v = 0.0;
        for (int c = 0; c < D.C; ++c) v += (D.Xc[c * P + i] * D.Xc[c * P + j]) * cell_sum[c];
        Mout[i * ld + j] = v;
        Mout[j * ld + i] = v;
    }
}

// ------------------------------------------------------------------ dispersion fit
struct WideAlphaArgs {
    const int32_t* y;
    const double* mu;
    const double* Xt;
    int ldx, N;
    double cst, la_hat, prior_var;
    const CellDesign* cells;  // nullable
};

// loss / gradient of fit_alpha_mle at log_alpha (same formulas as alpha_eval, dsq_alpha.h)
template <class Wv, bool GRAD>
DSQ_HD void alpha_eval_wide(const WideAlphaArgs& A, const WideWork& W, double la, bool cr_reg, bool prior_reg,
                            double& f, double& g) {
    la = Wv::uniform(la);
    const double alpha = Wv::uniform(exp(la));
    const double a = Wv::uniform(frcp(alpha));
    const double lal = Wv::uniform(flog(alpha));
    double lga, dga;
    lgamma_digamma<GRAD>(a, lga, dga);
    lga = Wv::uniform(lga);
    dga = Wv::uniform(dga);
    const bool cell = A.cells != nullptr;
    KSum accf;
    double accg = 0.0;
    WideGram<Wv, GRAD> gram;
    if (cr_reg) {
        if (cell) {
            for (int c = Wv::lane(); c < kMaxCells; c += Wv::W) { W.acc[c] = 0.0; W.acc[kMaxCells + c] = 0.0; }
        } else {
            gram.begin(W);
            wide_zero_pad_rows<Wv>(W);
        }
        Wv::sync();
    }
    const int n_end = ((A.N + 63) / 64) * 64;
    for (int n0 = 0; n0 < n_end; n0 += 64) {
        for (int l = Wv::lane(); l < 64; l += Wv::W) {  // device: one iteration, lane = sample of the chunk
            const int n = n0 + l;
            const bool valid = n < A.N;
            const int yi = valid ? A.y[n] : 0;
            const double m = valid ? A.mu[n] : 0.0;
            const double yv = (double)yi;
            double dl, dd;
            lgamma_digamma_diff<Wv, GRAD>(yi, a, lga, dga, dl, dd);
            const double ma = m * alpha;
            const double r1 = frcp(1.0 + ma);
            const double L1 = flog1p(ma);
            accf.add(dl + yv * (L1 - lal) + a * L1);
            if (GRAD) accg += dd + L1 + (yv - m) * alpha * r1;
            if (cr_reg) {
                const double w = m * r1;
                const double dw = -(w * w);
                if (cell) {
                    const int c = valid ? A.cells->cell_of[n] : 0;
                    Wv::cell_add(&W.acc[c], w);
                    if (GRAD) Wv::cell_add(&W.acc[kMaxCells + c], dw);
                } else {
                    W.w[l] = w;
                    W.w[64 + l] = dw;
                }
            }
        }
        if (cr_reg && !cell) {
            wide_stage_x<Wv>(W, A.Xt, A.ldx, A.N, n0);
            Wv::sync();
            gram.add_chunk(W);
            Wv::sync();
        }
    }
    const double sumf = Wv::sum_comp(accf);
    if (GRAD) accg = Wv::sum(accg);
    f = sumf + A.cst;
    g = 0.0;
    if (GRAD) g = alpha * (-(a * a * accg));
    if (cr_reg) {
        if (cell) {
            Wv::sync();
            wide_gram_from_cells<Wv>(W, *A.cells, W.acc, W.M);
            if (GRAD) wide_gram_from_cells<Wv>(W, *A.cells, W.acc + kMaxCells, W.dM);
            Wv::sync();
        } else {
            gram.finish(W);
        }
        wide_chol<Wv>(W, W.M, W.L, 0.0);
        f += 0.5 * wide_logdet<Wv>(W, W.L);
        if (GRAD) {
            wide_inverse<Wv>(W, W.L, W.Li, W.inv);
            g += 0.5 * wide_frob<Wv>(W, W.inv, W.dM) * alpha;
        }
    }
    if (prior_reg) {
        const double dl = la - A.la_hat;
        f += dl * dl / (2.0 * A.prior_var);
        if (GRAD) g += dl / A.prior_var;
    }
}

// fit_alpha_mle (utils.py:441-564) incl. the grid-search fallback (grid_search.py:54-142), any P
template <class Wv>
DSQ_HD AlphaOut fit_alpha_wide(const int32_t* y, const double* mu, const double* Xt, int ldx, int N,
                               const WideWork& W, const CellDesign* cells, double alpha_hat, double min_disp,
                               double max_disp, double prior_var, bool cr_reg, bool prior_reg, Lbfgsb1d& m,
                               const double* cst_in, double* cst_out) {
    WideAlphaArgs A;
    A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N; A.cells = cells;
    A.la_hat = log(alpha_hat);
    A.prior_var = prior_var;
    A.cst = cst_in != nullptr ? *cst_in : alpha_const<Wv>(y, mu, N);
    if (cst_out != nullptr && Wv::lane() == 0) *cst_out = A.cst;
    const double lo = log(min_disp), hi = log(max_disp);
    m.start(A.la_hat, lo, hi);
    while (!m.done) {
        double f, g;
        alpha_eval_wide<Wv, true>(A, W, m.x, cr_reg, prior_reg, f, g);
        m.feed(f, g);
    }
    AlphaOut o;
    o.converged = m.success ? 1 : 0;
    o.nfev = m.nfev; o.nit = m.it; o.status = m.status;
    o.alpha = exp(m.x);
    if (!m.success) {  // grid search: Cox-Reid term on, no prior (the reference passes six positional arguments)
        double lohi[2] = {lo, hi};
        double best_la = 0.0;
        for (int level = 0; level < 2; ++level) {
            double best = 0.0;
            int kbest = 0;
            bool best_nan = false;
            for (int i = 0; i < 100; ++i) {
                double f, gu;
                alpha_eval_wide<Wv, false>(A, W, linspace_at(lohi[0], lohi[1], 100, i), true, false, f, gu);
                const bool isn = (f != f);
                if (i == 0 || (!best_nan && (isn || f < best))) { best = f; kbest = i; best_nan = isn; }
            }
            const double c = linspace_at(lohi[0], lohi[1], 100, kbest);
            const double delta = linspace_at(lohi[0], lohi[1], 100, 1) - linspace_at(lohi[0], lohi[1], 100, 0);
            best_la = c;
            lohi[0] = c - delta; lohi[1] = c + delta;
        }
        o.alpha = exp(best_la);
    }
    return o;
}

// ------------------------------------------------------------------ IRLS (utils.py:273-438) with the fused epilogue
// vec slots: 0 beta, 1 rhs / solution, 2 beta_init, 3 contrast work, 4 M Hc
template <class Wv>
DSQ_HD void irls_sweep_wide(const IrlsArgs& A, const WideWork& W, double a, double& S) {
    // on entry beta = W.v(0); on exit W.M = X^T W X, W.v(1) = X^T W z, S = sum of the deviance terms
    const int P = W.P;
    const bool cell = A.cells != nullptr;
    const double lmin = log(A.min_mu);
    double s = 0.0;
    WideGram<Wv, false> gram;
    double rpart = 0.0;  // lane j < P: r_j
    if (cell) {
        const CellDesign& D = *A.cells;
        for (int c = Wv::lane(); c < kMaxCells; c += Wv::W) {
            double eta = 0.0;
            if (c < D.C)
                for (int j = 0; j < P; ++j) eta += D.Xc[c * P + j] * W.v(0)[j];
            W.tab[c] = eta;
            W.tab[kMaxCells + c] = exp(eta);
            W.acc[c] = 0.0;
            W.acc[kMaxCells + c] = 0.0;
        }
    } else {
        gram.begin(W);
        wide_zero_pad_rows<Wv>(W);
#if !defined(__HIP_DEVICE_COMPILE__)
        for (int j = 0; j < P; ++j) W.v(1)[j] = 0.0;
#endif
    }
    Wv::sync();
    const int n_end = ((A.N + 63) / 64) * 64;
    for (int n0 = 0; n0 < n_end; n0 += 64) {
        if (!cell) {
            wide_stage_x<Wv>(W, A.Xt, A.ldx, A.N, n0);
            Wv::sync();
        }
        for (int l = Wv::lane(); l < 64; l += Wv::W) {
            const int n = n0 + l;
            const bool valid = n < A.N;
            double w = 0.0, wz = 0.0;
            int c = 0;
            if (valid) {
                const double yv = (double)A.y[n];
                const double sfn = A.sf[n];
                double eta, e;
                if (cell) {
                    c = A.cells->cell_of[n];
                    eta = W.tab[c];
                    e = W.tab[kMaxCells + c];
                } else {
                    eta = 0.0;
                    for (int j = 0; j < P; ++j) eta += W.xs[j * kWideXsLd + l] * W.v(0)[j];
                    e = exp(eta);
                }
                const double mu_raw = sfn * e;
                const bool clamped = !(mu_raw > A.min_mu);
                const double mu = clamped ? A.min_mu : mu_raw;
                const double lsfn = (A.lsf != nullptr) ? A.lsf[n] : flog(sfn);
                const double lmu = clamped ? lmin : eta + lsfn;
                s += (yv + a) * flog(a + mu) - yv * lmu;
                w = mu * frcp(1.0 + mu * A.disp);
                const double z = (clamped ? lmin - lsfn : eta) + (yv - mu) * frcp(mu);
                wz = w * z;
            }
            if (cell) {
                if (valid) { Wv::cell_add(&W.acc[c], w); Wv::cell_add(&W.acc[kMaxCells + c], wz); }
            } else {
                W.w[l] = w;
                W.w[64 + l] = wz;
            }
        }
        if (!cell) {
            Wv::sync();
            gram.add_chunk(W);
            // r_j += sum_n x_nj (w z)_n : lane j walks its row of the staged chunk
#if defined(__HIP_DEVICE_COMPILE__)
            const int j = threadIdx.x & 63;
            if (j < P)
                for (int n = 0; n < 64; ++n) rpart += W.xs[j * kWideXsLd + n] * W.w[64 + n];
#else
            for (int j = 0; j < P; ++j)
                for (int n = 0; n < 64; ++n) W.v(1)[j] += W.xs[j * kWideXsLd + n] * W.w[64 + n];
#endif
            Wv::sync();
        }
    }
    if (cell) {
        Wv::sync();
        const CellDesign& D = *A.cells;
        wide_gram_from_cells<Wv>(W, D, W.acc, W.M);
        for (int j = Wv::lane(); j < P; j += Wv::W) {
            double v = 0.0;
            for (int c = 0; c < D.C; ++c) v += D.Xc[c * P + j] * W.acc[kMaxCells + c];
            W.v(1)[j] = v;
        }
        Wv::sync();
    } else {
        gram.finish(W);
#if defined(__HIP_DEVICE_COMPILE__)
        if ((int)(threadIdx.x & 63) < P) W.v(1)[threadIdx.x & 63] = rpart;
#endif
        Wv::sync();
    }
    S = Wv::sum(s);
}

// Wald statistic from W.M = X^T W X at the unclamped mu (no ridge), beta = W.v(0)   (utils.py:718-811)
template <class Wv>
DSQ_HD WaldOut wald_wide(const WideWork& W, const double* ridge, const double* contrast, double lfc_null, int alt) {
    const int P = W.P, ld = W.ld;
    // Hm = M + ridge in W.dM, chol -> W.L, Hc = solve(Hm, c) in v(3), MHc = M Hc in v(4)
    for (int e = Wv::lane(); e < P * P; e += Wv::W) {
        const int i = e / P, j = e % P;
        W.dM[i * ld + j] = W.M[i * ld + j] + ridge[i * P + j];
    }
    for (int j = Wv::lane(); j < P; j += Wv::W) W.v(3)[j] = contrast[j];
    Wv::sync();
    wide_chol<Wv>(W, W.dM, W.L, 0.0);
    wide_chol_solve<Wv>(W, W.L, W.v(3));
    for (int i = Wv::lane(); i < P; i += Wv::W) {
        double r = 0.0;
        for (int j = 0; j < P; ++j) r += W.M[i * ld + j] * W.v(3)[j];
        W.v(4)[i] = r;
    }
    Wv::sync();
    double q = 0.0;
    for (int j = 0; j < P; ++j) q += W.v(3)[j] * W.v(4)[j];
    WaldOut o;
    o.se = sqrt(q);
    const double* beta = W.v(0);
    double stat = 0.0, pval;
    if (alt == ALT_NONE) {
        double t = 0.0;
        for (int j = 0; j < P; ++j) t += contrast[j] * (beta[j] - lfc_null);
        stat = t / o.se;
        pval = 2.0 * norm_sf(fabs(stat));
    } else if (alt == ALT_GREATER) {
        for (int j = 0; j < P; ++j) stat += contrast[j] * np_fmax((beta[j] - lfc_null) / o.se, 0.0);
        pval = norm_sf(stat);
    } else if (alt == ALT_LESS) {
        for (int j = 0; j < P; ++j) stat += contrast[j] * np_fmin((beta[j] - lfc_null) / o.se, 0.0);
        pval = norm_sf(fabs(stat));
    } else if (alt == ALT_GREATER_ABS) {
        for (int j = 0; j < P; ++j)
            stat += contrast[j] * (dsign(beta[j]) * np_fmax((fabs(beta[j]) - lfc_null) / o.se, 0.0));
        pval = 2.0 * norm_sf(fabs(stat));
    } else {
        const double an = fabs(lfc_null);
        double sa = 0.0, sb = 0.0;
        for (int j = 0; j < P; ++j) {
            sa += contrast[j] * np_fmax((beta[j] + an) / o.se, 0.0);
            sb += contrast[j] * np_fmin((beta[j] - an) / o.se, 0.0);
        }
        const double pa = norm_sf(sa), pb = norm_sf(fabs(sb));
        stat = (fabs(sb) < fabs(sa)) ? sb : sa;
        pval = (pb > pa) ? pb : pa;
    }
    o.stat = stat;
    o.p = pval;
    return o;
}

// hat diagonal, unclamped mu, fused Cook's bookkeeping and Wald statistics; on entry W.M = X^T W X at the final
// clamped mu and beta = W.v(0)
template <class Wv>
DSQ_HD void irls_finish_wide(const IrlsArgs& A, const WideWork& W, double* mu_out, double* H_out, LfcEpilogue* E) {
    const int P = W.P;
    const bool want_cooks = E != nullptr && E->flags != nullptr;
    const bool want_wald = E != nullptr && E->ridge != nullptr;
    if (mu_out == nullptr && H_out == nullptr && !want_cooks && !want_wald) return;
    const bool cell = A.cells != nullptr;
    const bool want_hat = H_out != nullptr || want_cooks;
    if (want_hat) {
        wide_chol<Wv>(W, W.M, W.L, 1e-6);
        wide_inverse<Wv>(W, W.L, W.Li, W.inv);
    }
    CooksAcc<Wv> acc(want_cooks ? E->robust_disp : 0.0, want_cooks ? E->cutoff : 0.0, P);
    WideGram<Wv, false> gram;
    if (cell) {
        const CellDesign& D = *A.cells;
        for (int c = Wv::lane(); c < kMaxCells; c += Wv::W) {
            double eta = 0.0;
            if (c < D.C)
                for (int j = 0; j < P; ++j) eta += D.Xc[c * P + j] * W.v(0)[j];
            W.tab[kMaxCells + c] = exp(eta);
            W.acc[kMaxCells + c] = 0.0;
        }
        Wv::sync();
        if (want_hat) {  // q_c = x_c^T inv x_c through the staging buffer (cells as columns)
            for (int c = Wv::lane(); c < kMaxCells; c += Wv::W)
                for (int j = 0; j < P; ++j) W.xs[j * kWideXsLd + c] = c < D.C ? D.Xc[c * P + j] : 0.0;
            Wv::sync();
            for (int c = Wv::lane(); c < kMaxCells; c += Wv::W) W.acc[c] = wide_quad_xs(W, W.inv, c);
            Wv::sync();
        }
    } else if (want_wald) {
        gram.begin(W);
        wide_zero_pad_rows<Wv>(W);
        Wv::sync();
    }
    const int n_end = ((A.N + 63) / 64) * 64;
    for (int n0 = 0; n0 < n_end; n0 += 64) {
        if (!cell) {
            wide_stage_x<Wv>(W, A.Xt, A.ldx, A.N, n0);
            Wv::sync();
        }
        for (int l = Wv::lane(); l < 64; l += Wv::W) {
            const int n = n0 + l;
            const bool valid = n < A.N;
            double wu = 0.0;
            if (valid) {
                double mu_raw, q = 0.0;
                int c = 0;
                if (cell) {
                    c = A.cells->cell_of[n];
                    mu_raw = A.sf[n] * W.tab[kMaxCells + c];
                    q = W.acc[c];
                } else {
                    double eta = 0.0;
                    for (int j = 0; j < P; ++j) eta += W.xs[j * kWideXsLd + l] * W.v(0)[j];
                    mu_raw = A.sf[n] * exp(eta);
                    if (want_hat) q = wide_quad_xs(W, W.inv, l);
                }
                if (mu_out != nullptr) mu_out[n] = mu_raw;
                if (want_hat) {
                    const double mu = dmax(mu_raw, A.min_mu);
                    const double w = mu / (1.0 + mu * A.disp);
                    const double sw = sqrt(w);
                    const double h = sw * q * sw;
                    if (H_out != nullptr) H_out[n] = h;
                    if (want_cooks) {
                        const double ck = acc.add(n, (double)A.y[n], mu_raw, h, E->flags[n]);
                        if (E->cooks_row != nullptr) E->cooks_row[n] = ck;
                    }
                }
                wu = mu_raw / (1.0 + mu_raw * A.disp);
                if (want_wald && cell) Wv::cell_add(&W.acc[kMaxCells + c], wu);
            }
            if (want_wald && !cell) W.w[l] = wu;
        }
        if (!cell) {
            Wv::sync();
            if (want_wald) gram.add_chunk(W);
            Wv::sync();
        }
    }
    if (want_cooks) E->cooks = acc.finish(A.y, A.N);
    if (want_wald) {
        if (cell) {
            Wv::sync();
            wide_gram_from_cells<Wv>(W, *A.cells, W.acc + kMaxCells, W.M);
            Wv::sync();
        } else {
            gram.finish(W);
        }
        E->wald = wald_wide<Wv>(W, E->ridge, E->contrast, E->lfc_null, E->alt);
    }
}

// irls_solver (utils.py:273-438) for any P; beta ends in W.v(0).  out.fallback = 1: IRLS diverged, call
// irls_rescue_wide (nothing has been written).
template <class Wv>
DSQ_HD IrlsOut irls_gene_wide(const IrlsArgs& A, const WideWork& W, double* mu_out, double* H_out, LfcEpilogue* E) {
    const int P = W.P;
    IrlsOut out;
    out.converged = 1; out.iters = 0; out.fallback = 0;
    const double a = 1.0 / A.disp;
    // beta_init (utils.py:349-357) and the mu-independent part of the NLL, as irls_init_exact
    double c = 0.0;
    for (int j = 0; j < P; ++j) {
        double b0 = 0.0;
        for (int n = Wv::lane(); n < A.N; n += Wv::W) {
            const double yv = (double)A.y[n];
            if (j == 0) c += lgamma_pos(yv + a) - lgamma_pos(yv + 1.0);
            if (A.full_rank) b0 += A.pinvXt[j * A.ldx + n] * log(yv / A.sf[n] + 0.1);
            else if (j == 0) b0 += log(yv / A.sf[n]);
        }
        b0 = Wv::sum(b0);
        if (!A.full_rank) b0 = j == 0 ? b0 / (double)A.N : 0.0;
        if (Wv::lane() == 0) { W.v(0)[j] = b0; W.v(2)[j] = b0; }
    }
    const double cst = Wv::sum(c) - A.N * lgamma_pos(a);
    Wv::sync();
    const double nlogterm = A.N * a * log(A.disp);
    double S;
    irls_sweep_wide<Wv>(A, W, a, S);
    double dev = 1000.0, ratio = 1.0;
    int i = 0;
    while (ratio > A.beta_tol) {
        wide_chol<Wv>(W, W.M, W.L, 1e-6);
        wide_chol_solve<Wv>(W, W.L, W.v(1));
        i += 1;
        bool bad = (i >= A.maxiter);
        for (int j = 0; j < P; ++j) bad = bad || (fabs(W.v(1)[j]) > A.max_beta);
        if (bad) {
            out.fallback = 1; out.converged = 0; out.iters = i;
            return out;
        }
        Wv::sync();
        for (int j = Wv::lane(); j < P; j += Wv::W) W.v(0)[j] = W.v(1)[j];
        Wv::sync();
        irls_sweep_wide<Wv>(A, W, a, S);
        const double old = dev;
        dev = -2.0 * (nlogterm - cst + S);
        ratio = fabs(dev - old) / (fabs(dev) + 0.1);
    }
    out.iters = i;
    irls_finish_wide<Wv>(A, W, mu_out, H_out, E);
    return out;
}

// rescue of a diverged gene (utils.py:374-413): bounded L-BFGS-B from beta_init = W.v(2)
template <class Wv>
DSQ_HD IrlsOut irls_rescue_wide(const IrlsArgs& A, const WideWork& W, LbfgsbWork<kWideMaxP>& Lb, double* xlu /*[3][32]*/,
                                int* nbd, double* mu_out, double* H_out, LfcEpilogue* E) {
    const int P = W.P;
    IrlsOut out;
    out.converged = 0; out.iters = 0; out.fallback = 1;
    const double a = 1.0 / A.disp;
    double c = 0.0;
    for (int n = Wv::lane(); n < A.N; n += Wv::W) {
        const double yv = (double)A.y[n];
        c += lgamma_pos(yv + a) - lgamma_pos(yv + 1.0);
    }
    const double cst = Wv::sum(c) - A.N * lgamma_pos(a);
    const double nlogterm = A.N * a * log(A.disp);
    double* x = xlu;
    double* lo = xlu + kWideMaxP;
    double* hi = xlu + 2 * kWideMaxP;
    Wv::sync();
    for (int j = Wv::lane(); j < P; j += Wv::W) { x[j] = W.v(2)[j]; lo[j] = A.min_beta; hi[j] = A.max_beta; nbd[j] = 2; }
    Wv::sync();
    auto fg = [&](const double* xb, double& f, double* g) {
        double s = 0.0;
        // per-sample: eta from global X (coalesced over lanes), gradient entries reduced one at a time
        for (int j = 0; j < P; ++j) {
            double gj = 0.0;
            for (int n = Wv::lane(); n < A.N; n += Wv::W) {
                const double yv = (double)A.y[n];
                double eta = 0.0;
                for (int k = 0; k < P; ++k) eta += A.Xt[k * A.ldx + n] * xb[k];
                const double mu = dmax(A.sf[n] * exp(eta), A.min_mu);
                if (j == 0) s += (yv + a) * log(a + mu) - yv * log(mu);
                gj += (-yv + (a + yv) * mu / (a + mu)) * A.Xt[j * A.ldx + n];
            }
            gj = Wv::sum(gj);
            g[j] = gj + 1e-6 * xb[j];
        }
        s = Wv::sum(s);
        double pen = 0.0;
        for (int j = 0; j < P; ++j) pen += 1e-6 * (xb[j] * xb[j]);
        f = (nlogterm - cst + s) + 0.5 * pen;
    };
    const LbfgsbResult res = lbfgsb_nd<kWideMaxP>(fg, P, x, lo, hi, nbd, Lb);
    Wv::sync();
    for (int j = Wv::lane(); j < P; j += Wv::W) W.v(0)[j] = x[j];
    Wv::sync();
    out.converged = res.success ? 1 : 0;
    out.iters = res.nit;
    double S2;
    irls_sweep_wide<Wv>(A, W, a, S2);
    irls_finish_wide<Wv>(A, W, mu_out, H_out, E);
    return out;
}

// ------------------------------------------------------------------ the cheap stages at any P
// rough + moments dispersions, normalised mean and (optionally) the linear-model mu_hat (mom_gene / lin_mu_gene);
// OLS coefficients in W.v(0)
template <class Wv>
DSQ_HD MomOut mom_wide(const int32_t* y, const double* sf, const double* Xt, const double* pinvXt, int ldx, int N,
                       const WideWork& W, double s_mean_inv, double min_disp, double max_disp, double min_mu,
                       double* mu_out) {
    const int P = W.P;
    double s = 0.0;
    for (int n = Wv::lane(); n < N; n += Wv::W) s += (double)y[n] / sf[n];
    s = Wv::sum(s);
    for (int j = 0; j < P; ++j) {
        double b = 0.0;
        for (int n = Wv::lane(); n < N; n += Wv::W) b += pinvXt[j * ldx + n] * ((double)y[n] / sf[n]);
        b = Wv::sum(b);
        if (Wv::lane() == 0) W.v(0)[j] = b;
    }
    Wv::sync();
    const double mean = s / (double)N;
    double ss = 0.0, rr = 0.0;
    const double dof = (double)(N - P);
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double sfn = sf[n];
        const double v = (double)y[n] / sfn;
        const double d = v - mean;
        ss += d * d;
        double yh = 0.0;
        for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * W.v(0)[j];
        if (mu_out != nullptr) mu_out[n] = dmax(sfn * yh, min_mu);
        yh = dmax(yh, 1.0);
        rr += ((v - yh) * (v - yh) - yh) / (dof * yh * yh);
    }
    ss = Wv::sum(ss);
    rr = Wv::sum(rr);
    MomOut o;
    o.normed_mean = mean;
    o.rough = dmax(rr, 0.0);
    const double var = ss / (double)(N - 1);
    double m = (var - s_mean_inv * mean) / (mean * mean);
    if (m != m) m = 0.0;
    else if (m == INFINITY) m = DBL_MAX;
    else if (m == -INFINITY) m = -DBL_MAX;
    o.moments = m;
    o.mom = dmin(dmax(dmin(o.rough, o.moments), min_disp), max_disp);
    return o;
}

// wald_test on given coefficients (Inference.wald_test / another contrast): M from the caller's mu row or from
// mu = sf exp(X beta); beta must be in W.v(0)
template <class Wv>
DSQ_HD WaldOut wald_gene_wide(const double* mu, const double* sf, const double* Xt, int ldx, int N, double disp,
                              const WideWork& W, const double* ridge, const double* contrast, double lfc_null,
                              int alt) {
    const int P = W.P;
    WideGram<Wv, false> gram;
    gram.begin(W);
    wide_zero_pad_rows<Wv>(W);
    Wv::sync();
    const int n_end = ((N + 63) / 64) * 64;
    for (int n0 = 0; n0 < n_end; n0 += 64) {
        wide_stage_x<Wv>(W, Xt, ldx, N, n0);
        Wv::sync();
        for (int l = Wv::lane(); l < 64; l += Wv::W) {
            const int n = n0 + l;
            double w = 0.0;
            if (n < N) {
                double m;
                if (mu != nullptr) m = mu[n];
                else {
                    double eta = 0.0;
                    for (int j = 0; j < P; ++j) eta += W.xs[j * kWideXsLd + l] * W.v(0)[j];
                    m = sf[n] * exp(eta);
                }
                w = m / (1.0 + m * disp);
            }
            W.w[l] = w;
        }
        Wv::sync();
        gram.add_chunk(W);
        Wv::sync();
    }
    gram.finish(W);
    return wald_wide<Wv>(W, ridge, contrast, lfc_null, alt);
}

}  // namespace dsq
