// dsq_lbfgsb_par.h — the dense linear algebra of L-BFGS-B's compact representation (dsq_lbfgsb.h: formk, subsm, matupd,
// formt; LINPACK dpofa / dtrsl) spread over the lanes of a wavefront WITHOUT changing any element's arithmetic.
//
// Why.  dsq_lbfgsb.h restates scipy's L-BFGS-B routine by routine as scalar code on a wave-private LDS workspace: every
// lane executes the same ~3500 dependent LDS operations per iteration (two Cholesky factorisations of up to 10 x 10, up
// to 10 + 2 triangular solves, a Schur complement, the updates of the 2m x 2m middle matrix).  With the objective of the
// apeGLM shrinkage (utils.py:990-1207) that is 10 x the cost of the ~30 function evaluations: k_shrink<8> spent 64 ms on
// 60 000 genes where the evaluations need ~5 (profiles/r03_c4.txt).
// The iterates must stay those of scipy (the reference returns wherever that optimiser stops), so nothing here may
// reassociate a sum.  But the routines have independent OUTPUTS: the entries of a row of a Cholesky factor, the right-hand
// sides of a batch of triangular solves, the entries of a Schur complement, the components of a solution vector that
// receive the same column update.  Each output is owned by one lane, which performs exactly the operations, in exactly
// the order, that the scalar routine performs for that output; lanes meet at wave barriers between the steps of a sweep.
// A policy with one lane (OneLane below, or tests/hostsim's HostWave) runs the same code sequentially - the host build,
// which tests/test_hostsim.py holds to scipy's iterates, therefore validates the arithmetic of the device build.
#pragma once
#include "dsq_math.h"

namespace dsq {

// every "lane" does everything: the behaviour of the scalar routines when a whole wavefront executes them redundantly
struct OneLane {
    static constexpr int W = 1;
    DSQ_HD static int lane() { return 0; }
    DSQ_HD static void sync() {}
};

namespace lbp {

// LINPACK dpofa (upper triangle, leading dimension lda, 1-based (i,j)): rows of the factor one after the other, the
// entries of a row in parallel.  Lane j computes A(k,j) = (A(k,j) - sum_{q<k} A(q,k) A(q,j)) / A(k,k) for k = 1 .. j-1 in
// that order and accumulates s_j = sum_k A(k,j)^2 in that order (sacc[j-1], touched by its lane only); then
// A(j,j) = sqrt(A(j,j) - s_j).  Returns j at the first non-positive pivot, like the original (the caller discards the
// matrix then), else 0.
template <class Wv>
DSQ_HD int dpofa(double* a, int lda, int n, double* sacc) {
#define A_(i, j) a[((i)-1) + ((j)-1) * lda]
    for (int j = 1 + Wv::lane(); j <= n; j += Wv::W) sacc[j - 1] = 0.0;
    for (int k = 1; k <= n; ++k) {
        // pivot of row k (its lane has finished column k's off-diagonal entries in the previous steps)
        Wv::sync();
        {
            const double s = A_(k, k) - sacc[k - 1];
            if (s <= 0.0) return k;  // (every lane reads the same two numbers: a uniform exit)
            Wv::sync();              // all lanes have read the old diagonal
            if (Wv::lane() == (k - 1) % Wv::W) A_(k, k) = sqrt(s);
        }
        Wv::sync();
        const double dkk = A_(k, k);
        for (int j = k + 1 + Wv::lane(); j <= n; j += Wv::W) {
            double t = A_(k, j);
            for (int q = 1; q <= k - 1; ++q) t -= A_(q, k) * A_(q, j);
            t = t / dkk;
            A_(k, j) = t;
            sacc[j - 1] += t * t;
        }
    }
    Wv::sync();
    return 0;
#undef A_
}

// LINPACK dtrsl, upper-triangular t, ONE right-hand side, components in parallel.
// job 11 (t' x = b): b_j = (b_j - sum_{q<j} T(q,j) b_q) / T(j,j): lane j adds T(q,j) b_q to its sum (sacc[j-1]) as soon
// as b_q is final, q = 1, 2, ... - the order of the original's inner loop.
// job 01 (t x = b): the original's column sweep b_q += (-b_{j+1}) T(q, j+1), j + 1 = n .. 2, with lane q owning b_q.
template <class Wv>
DSQ_HD int dtrsl_upper(const double* t, int ldt, int n, double* b, int job, double* sacc) {
#define T_(i, j) t[((i)-1) + ((j)-1) * ldt]
    for (int j = 1; j <= n; ++j)
        if (T_(j, j) == 0.0) return j;
    if (job == 1) {  // t x = b
        Wv::sync();
        if (Wv::lane() == (n - 1) % Wv::W) b[n - 1] = b[n - 1] / T_(n, n);
        for (int jj = 2; jj <= n; ++jj) {
            const int j = n - jj + 1;
            Wv::sync();
            const double temp = -b[j];  // b(j+1), final
            for (int q = 1 + Wv::lane(); q <= j; q += Wv::W) {
                double v = b[q - 1] + temp * T_(q, j + 1);
                if (q == j) v = v / T_(j, j);
                b[q - 1] = v;
            }
        }
    } else {  // t' x = b
        for (int j = 1 + Wv::lane(); j <= n; j += Wv::W) sacc[j - 1] = 0.0;
        Wv::sync();
        if (Wv::lane() == 0) b[0] = b[0] / T_(1, 1);
        for (int q = 1; q <= n - 1; ++q) {
            Wv::sync();
            const double bq = b[q - 1];  // final
            for (int j = q + 1 + Wv::lane(); j <= n; j += Wv::W) {
                const double s = sacc[j - 1] + T_(q, j) * bq;
                sacc[j - 1] = s;
                if (j == q + 1) {
                    double v = b[j - 1] - s;
                    v = v / T_(j, j);
                    b[j - 1] = v;
                }
            }
        }
    }
    Wv::sync();
    return 0;
#undef T_
}

// the scalar dtrsl (dsq_lbfgsb.h) on a right-hand side that belongs to ONE lane (batches: a lane per right-hand side)
DSQ_HD void dtrsl_upper_t_own(const double* t, int ldt, int n, double* b) {  // job 11
#define T_(i, j) t[((i)-1) + ((j)-1) * ldt]
    b[0] = b[0] / T_(1, 1);
    for (int j = 2; j <= n; ++j) {
        double s = 0.0;
        for (int q = 1; q <= j - 1; ++q) s += T_(q, j) * b[q - 1];
        b[j - 1] = b[j - 1] - s;
        b[j - 1] = b[j - 1] / T_(j, j);
    }
#undef T_
}

}  // namespace lbp
}  // namespace dsq
