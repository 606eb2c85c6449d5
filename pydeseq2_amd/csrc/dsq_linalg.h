// dsq_linalg.h — tiny dense symmetric P x P algebra held entirely in registers.
//
// The p x p normal-equation matrices X^T W X of the NB-GLM (p = design
// columns, 1..16) are symmetric positive definite; they are stored packed
// (lower triangle, tri(i,j) = i(i+1)/2 + j, i >= j).  P is a template parameter
// so every loop unrolls and nothing is indexed dynamically (no scratch).
// Replaces numpy.linalg.slogdet / inv and scipy.linalg.solve(assume_a="pos")
// at pydeseq2/utils.py:370-371, 428-430, 515, 532, 772-776.
#pragma once
#include <type_traits>
#include <cstdint>

#include "dsq_math.h"

namespace dsq {

template <int P>
struct Tri {
    static constexpr int N = P * (P + 1) / 2;
};

DSQ_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }          // i >= j
DSQ_HD constexpr int tris(int i, int j) { return i >= j ? tri(i, j) : tri(j, i); }  // any order

// Designs whose rows take few distinct values ("design cells": every purely categorical design): X^T W X is
// sum_c (sum_{n in cell c} w_n) x_c x_c^T, so the per-sample work of the fits does not depend on P at all.
// The per-gene routines then add w_n into per-cell accumulators (wave-private LDS) and rebuild the matrix
// entry-parallel: lane e owns entry e and walks the <= 64 cells.
constexpr int kMaxCells = 64;
constexpr int kSmallCells = 4;  // up to this many cells: per-cell sums in registers (IRLS, dsq_irls.h irls_sweep_cs)
struct CellDesign {
    const int32_t* cell_of;  // [ldx] design cell of every sample (0 beyond N)
    const double* Xc;        // [C][P] the cells' design rows
    const double* XX;        // [C][T] x_i x_j of the cells' rows, packed lower triangle
    int C;
};
// wave-private workspace of the cell path (LDS on the device)
template <int P>
struct CellWork {
    double acc[2][kMaxCells];        // per-cell sums of the current sweep
    double tab[2][kMaxCells];        // per-cell table fetched by the samples (linear predictor, its exponential, ...)
    double ent[2 * (P * (P + 1) / 2)];  // matrix entries on their way from the lane that computed them to all lanes
};

// The p x p algebra runs redundantly in every lane between the sample loops; at p = 8 and N = 500 it used to cost
// three times the sample loop itself, almost all of it in IEEE divisions (~30 instructions each on gfx950) and
// library logarithms (~120).  Reciprocals therefore go through frcp (v_rcp_f64 + two Newton steps, <= 1 ulp), the
// log-determinant takes ONE log of the product of the pivots, and a pivot's reciprocal is computed once.
//
// in-place Cholesky A = L L^T (lower, packed).  Non-SPD input yields NaNs (sqrt of <0).
template <int P>
DSQ_HD void chol(double (&a)[Tri<P>::N]) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double d = a[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= a[tri(j, k)] * a[tri(j, k)];
        d = sqrt(d);
        a[tri(j, j)] = d;
        const double r = frcp(d);
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double s = a[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= a[tri(i, k)] * a[tri(j, k)];
            a[tri(i, j)] = s * r;
        }
    }
}

// compile-time loop (the index reaches DPP controls, which are immediates)
template <int I, int N, class F>
DSQ_HD void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Solve (A + ridge I) x = b for one gene held by a SIXTEEN-LANE ROW (RowWave, P <= 16): lane i keeps row i of the
// lower triangle (P registers) and its entry of b; pivots, columns and solution entries travel by row broadcasts.
// The register kernels keep the whole p x p matrix in every lane (p (p + 1) registers at fp64 - the reason k_irls_row
// spilled from p = 6 on); here the matrix costs 2 p registers per lane for the same ~p^3 / 3 instructions.
//   ent: packed lower triangle of A followed by b (T + P doubles, LDS);  x: the solution, in every lane
template <class Wv, int P, class Ent>
DSQ_HD void row_chol_solve(const Ent& ent, double ridge, double (&x)[P]) {
    constexpr int T = Tri<P>::N;
    const int rl = Wv::lane();
    const int ri = rl < P ? rl : P - 1;  // (lanes beyond the matrix shadow its last row)
    double a[P], rinv[P];
#pragma unroll
    for (int j = 0; j < P; ++j) a[j] = ent[tri(ri, j <= ri ? j : ri)] + (j == ri ? ridge : 0.0);
    double b = ent[T + ri];
    // right-looking Cholesky: after step j, a[j] of lane i >= j is L_ij
    static_for<0, P>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const double r = frsq(Wv::template row_bcast<j>(a[j]));
        rinv[j] = r;
        a[j] *= r;
        static_for<j + 1, P>([&](auto K) {
            constexpr int k = decltype(K)::value;
            a[k] -= a[j] * Wv::template row_bcast<k>(a[j]);  // (meaningful in lanes i >= k)
        });
    });
    // forward substitution: y_k is final in lane k, every later row takes its term
    static_for<0, P>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const double yk = Wv::template row_bcast<k>(b) * rinv[k];
        b = rl == k ? yk : (rl > k ? b - a[k] * yk : b);
    });
    // back substitution with L^T: x_i final in lane i, lane k < i takes L_ik x_i (L_ik lives in lane i)
    static_for<0, P>([&](auto I) {
        constexpr int i = P - 1 - decltype(I)::value;
        const double xi = Wv::template row_bcast<i>(b) * rinv[i];
        x[i] = xi;
        b = rl == i ? xi : b;
        static_for<0, i>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const double lik = Wv::template row_bcast<i>(a[k]);
            b = rl == k ? b - lik * xi : b;
        });
    });
}

// log det(L L^T) = 2 log prod_j L_jj.  The pivots are square roots of weighted sums of squares of design entries
// (1e-4 .. 1e5 for any realistic fit), so the product of up to 12 of them stays far inside the double range; a
// product out of that range is split in two; a NaN pivot (matrix not positive definite) gives NaN.
template <int P>
DSQ_HD double chol_logdet(const double (&l)[Tri<P>::N]) {
    // two half products (the fallback used to be P library logarithms, which the compiler evaluated ahead of the range
    // test on every call: ~80 instructions each)
    double p1 = 1.0, p2 = 1.0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        if (j < (P + 1) / 2) p1 *= l[tri(j, j)];
        else p2 *= l[tri(j, j)];
    }
    const double prod = p1 * p2;
    if (prod > 1e-280 && prod < 1e280) return 2.0 * flog(prod);
    if (p1 == 0.0 || p2 == 0.0) return -INFINITY;                       // a zero pivot: log 0
    if (!(p1 > 0.0 && p1 < INFINITY && p2 > 0.0 && p2 < INFINITY)) return NAN;  // a NaN pivot (not positive definite)
    return 2.0 * (flog(p1) + flog(p2));
}

// solve (L L^T) x = b in place
template <int P>
DSQ_HD void chol_solve(const double (&l)[Tri<P>::N], double (&b)[P]) {
    double rinv[P];
#pragma unroll
    for (int i = 0; i < P; ++i) rinv[i] = frcp(l[tri(i, i)]);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= l[tri(i, k)] * b[k];
        b[i] = s * rinv[i];
    }
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < P; ++k) s -= l[tri(k, i)] * b[k];
        b[i] = s * rinv[i];
    }
}

// x^T (L L^T)^-1 x = |L^-1 x|^2 by one forward substitution (rinv: reciprocals of the pivots): no inverse matrix
template <int P>
DSQ_HD void chol_rinv(const double (&l)[Tri<P>::N], double (&rinv)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) rinv[i] = frcp(l[tri(i, i)]);
}
template <int P>
DSQ_HD double chol_quad(const double (&l)[Tri<P>::N], const double (&rinv)[P], const double (&x)[P]) {
    double t[P], q = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= l[tri(i, k)] * t[k];
        t[i] = s * rinv[i];
        q += t[i] * t[i];
    }
    return q;
}

// inv = (L L^T)^-1, packed symmetric
template <int P>
DSQ_HD void chol_inverse(const double (&l)[Tri<P>::N], double (&inv)[Tri<P>::N]) {
    // Linv (lower) first
    double li[Tri<P>::N];
#pragma unroll
    for (int j = 0; j < P; ++j) li[tri(j, j)] = frcp(l[tri(j, j)]);
#pragma unroll
    for (int j = 0; j < P; ++j) {
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s -= l[tri(i, k)] * li[tri(k, j)];
            li[tri(i, j)] = s * li[tri(i, i)];
        }
    }
    // inv = Linv^T Linv
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = i; k < P; ++k) s += li[tri(k, i)] * li[tri(k, j)];
            inv[tri(i, j)] = s;
        }
    }
}

// packed lower triangle of L^-1 for L from chol<P> (in place is not possible: rows are read while columns are written)
template <int P>
DSQ_HD void tri_inverse(const double (&l)[Tri<P>::N], double (&li)[Tri<P>::N]) {
#pragma unroll
    for (int j = 0; j < P; ++j) li[tri(j, j)] = frcp(l[tri(j, j)]);
#pragma unroll
    for (int j = 0; j < P; ++j) {
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s -= l[tri(i, k)] * li[tri(k, j)];
            li[tri(i, j)] = s * li[tri(i, i)];
        }
    }
}

// tr((L L^T)^-1 B) = sum_k l_k B l_k^T with l_k = row k of L^-1 (B symmetric, packed)
template <int P>
DSQ_HD double trace_inv_times(const double (&li)[Tri<P>::N], const double (&b)[Tri<P>::N]) {
    double tr = 0.0;
#pragma unroll
    for (int k = 0; k < P; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j <= k; ++j) r += b[tris(i, j)] * li[tri(k, j)];
            tr += r * li[tri(k, i)];
        }
    }
    return tr;
}

// x^T A x for packed symmetric A
template <int P>
DSQ_HD double sym_quad(const double (&a)[Tri<P>::N], const double (&x)[P]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) r += a[tris(i, j)] * x[j];
        s += r * x[i];
    }
    return s;
}

// y = A x for packed symmetric A
template <int P>
DSQ_HD void sym_matvec(const double (&a)[Tri<P>::N], const double (&x)[P], double (&y)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) r += a[tris(i, j)] * x[j];
        y[i] = r;
    }
}

// sum_ij A_ij B_ij for packed symmetric A, B (off-diagonals count twice)
template <int P>
DSQ_HD double sym_frob(const double (&a)[Tri<P>::N], const double (&b)[Tri<P>::N]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) s += 2.0 * a[tri(i, j)] * b[tri(i, j)];
        s += a[tri(i, i)] * b[tri(i, i)];
    }
    return s;
}

}  // namespace dsq
