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
#include "dsq_bfgs.h"
#include "dsq_lbfgsb1d.h"
#include "dsq_lgamma_int.h"
#include "dsq_linalg.h"
#include "dsq_wave.h"

// The sample loop is software-pipelined by hand (the reads of the next iteration are issued before this one's
// arithmetic); rolled, the rotation of the prefetched values costs ~35 register moves per trip (22 % of the loop's
// instructions).  Two trips per loop iteration (DSQ_ALPHA_UNROLL 2) let the compiler rename instead of move:
// measured -13 % instructions in the loop but only -1.4 % time (the loop is bound by its dependent fp64 chains, not
// by issue slots), and the larger live set spills: 680 MB of scratch writes per full-size launch instead of 48 MB.
// Kept rolled.
#ifndef DSQ_ALPHA_UNROLL
#define DSQ_ALPHA_UNROLL 1
#endif

namespace dsq {

struct CellCtx {
    CellDesign D;
    void* ws;
};

struct AlphaArgs {
    const int32_t* y;   // [N]
    const double* mu;   // [N]
    const double* Xt;   // [P][ldx]
    int ldx;
    int N;
    double cst;         // sum lgamma(y+1) - sum y log(mu)   (alpha independent)
    double la_hat;      // log(alpha_hat)
    double prior_var;
    // CELL instantiations: the design's cells and this wave's CellWork<P>.  (16 dwords in all: the evaluation is an
    // out-of-line call and an argument struct of up to 16 registers travels in VGPRs; one more field and the whole
    // struct goes through scratch memory on every call - measured 0.64 GB of scratch writes per launch.)
    const CellCtx* cell = nullptr;
};

// lgamma(a) - lgamma(y + a) and digamma(a) - digamma(y + a) for a count y >= 0.
//   y <= 9 : exact recurrences  -log prod_{i<y}(a+i),  -sum_{i<y} 1/(a+i)   (y = 0 costs nothing)
//   y >= 10: Stirling series at z = y + a >= 10 minus the per-gene lgamma(a), digamma(a)
// (the reference evaluates gammaln / polygamma at both arguments and subtracts, utils.py:218,
// 263-264; the differences are what enters the likelihood).
constexpr int kSmallCount = 9;

constexpr int kMemoBlocks = 4;  // memo of up to 256 counts: lane k holds counts k, k+64, k+128, k+192

template <class Wv, bool GRAD, bool BIG = false>
DSQ_HD void lgamma_digamma_diff(int yi, double a, double lga, double dga, double& dl, double& dd) {
    if (BIG) {  // caller guarantees yi >= 256: Stirling only, truncated tails
        const double z = (double)yi + a;
        const double lg = flog_t(z);
        const double rc = frcp(z);
        dl = lga - ((z - 0.5) * lg - z + kHalfLog2Pi + stirling_tail_big(rc));
        dd = GRAD ? dga - (lg + digamma_tail_big(rc)) : 0.0;
        return;
    }
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
    const double lg = flog_t(arg);  // (every kernel that builds the memo has filled the table, dsq_math.h)
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
#if defined(__HIPCC__)
#define DSQ_EVAL_FN __host__ __device__ __attribute__((noinline))
#else
#define DSQ_EVAL_FN inline
#endif
// The evaluation is an out-of-line function per (P, memo size, cell mode) for narrow designs - one inlined kernel
// spilled at the 168-register budget of 3 waves/SIMD - whose argument struct (16 dwords) and two results travel in
// registers.  The cell-path evaluation of P >= DSQ_EVAL_INLINE_MIN_P is inlined: at those widths the callee touches
// most of the callee-saved VGPRs (v40-47, v56-63, ...) and its prologue / epilogue saved and restored 61 of them
// through scratch on EVERY call (rocprofv3: 4.9 GB of WRITE_SIZE per k_alpha<8, cells> launch against 0.5 MB
// algorithmic; same run time either way - the scratch stays in L2/MALL).  The general wide evaluation stays out of
// line: inlined it is 16 % slower at N = 5000 (register allocation of the p(p+1) accumulators).
#ifndef DSQ_EVAL_INLINE_MIN_P
#define DSQ_EVAL_INLINE_MIN_P 7
#endif
struct EvalOut {
    double f, g;
};
template <class Wv, int P, bool GRAD, bool PAD = false, int NB = 1, bool CELL = false>
DSQ_HD EvalOut alpha_eval_body(const AlphaArgs& A, double la, bool cr_reg, bool prior_reg) {
    double f, g;
    constexpr int T = Tri<P>::N;
    constexpr bool kSplitDM = P >= 9 && !CELL;
    DSQ_PHASE(2);
    la = Wv::uniform(la);
    const double alpha = Wv::uniform(exp(la));
    const double a = Wv::uniform(frcp(alpha));
    // log of the ROUNDED alpha: keeps every term a function of the same alpha (using `la` itself
    // would leave an inconsistency of ulp(1) * sum(y) in the loss, i.e. line-search noise)
    const double lal = Wv::uniform(flog_t(alpha));
    double lga, dga;
    lgamma_digamma<GRAD, true>(a, lga, dga);
    lga = Wv::uniform(lga);
    dga = Wv::uniform(dga);
    // Wave-level memo: lane k evaluates the two gamma-function differences for the COUNTS k, k+64,
    // ... once per evaluation; samples whose count is < 64 * memo_blocks then fetch them with a
    // cross-lane read instead of recomputing log/Stirling/recurrences per sample (counts repeat
    // heavily within a gene).  memo_blocks follows the gene's largest count (wave-uniform), so a
    // low-count gene builds one block only; counts >= 256 take the per-sample Stirling path.
    // (Host build: table of one entry.)
    // NB (1, 2 or 4 blocks of 64 counts) is a template parameter so that the sample loop carries no
    // branches: fit_alpha_gene picks the instantiation from the gene's largest count.
    static_assert(NB >= 1 && NB <= kMemoBlocks, "memo blocks");
    double tab_dl[NB], tab_dd[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t)
        lgamma_digamma_diff<Wv, GRAD>(Wv::lane() + t * Wv::W, a, lga, dga, tab_dl[t], tab_dd[t]);
    constexpr int tab_n = NB * Wv::W;
    KSum accf;
    double accg = 0.0;
    double M[T], dM[T];
#pragma unroll
    for (int k = 0; k < T; ++k) { M[k] = 0.0; dM[k] = 0.0; }
    // CELL: per-cell sums of w and dw instead of p(p+1) accumulators per lane (dsq_linalg.h, CellDesign)
    typedef DSQ_LDS_STRUCT(CellWork<P>) LdsWork;  // ds_read / ds_write instead of flat accesses
    LdsWork* const Wk = CELL ? (LdsWork*)A.cell->ws : nullptr;
    const int32_t* const cell_of = CELL ? A.cell->D.cell_of : nullptr;
    if (CELL && cr_reg) {
        for (int c = Wv::lane(); c < kMaxCells; c += Wv::W) { Wk->acc[0][c] = 0.0; Wk->acc[1][c] = 0.0; }
        Wv::sync();
    }
    DSQ_PHASE(3);
    // PAD: y / mu are padded to a multiple of the wave width with (0, 0.0), which makes every
    // per-sample contribution exactly zero (L1 = 0, w = 0, memo[0] = 0) without masking; without PAD
    // out-of-range lanes are given the same (0, 0.0).
    //
    // The loop is software-pipelined: iteration i issues the LDS / global reads of iteration i+1
    // (counts: i+2) and the cross-lane memo fetch of iteration i+1 before it starts its own ~100
    // dependent fp64 instructions, so that none of those latencies is exposed (3 waves per SIMD are
    // not enough to hide a serial  ds_read -> ds_bpermute -> use  chain per iteration).
    const int n_end = PAD ? ((A.N + Wv::W - 1) / Wv::W) * Wv::W : A.N;
    const int n_last = n_end - Wv::W + Wv::lane();  // PAD: this lane's slot of the last iteration
    // PAD rows are this wave's LDS segment (k_alpha stages them), the others and the design are global
    auto load_y = [&](int n) -> int {
        if constexpr (PAD) return DSQ_AS_LDS(int32_t, A.y)[n < n_last ? n : n_last];
        else return n < A.N ? DSQ_AS_GLOBAL(int32_t, A.y)[n] : 0;
    };
    auto load_m = [&](int n) -> double {
        if constexpr (PAD) return DSQ_AS_LDS(double, A.mu)[n < n_last ? n : n_last];
        else return n < A.N ? DSQ_AS_GLOBAL(double, A.mu)[n] : 0.0;
    };
    const auto Xg = DSQ_AS_GLOBAL(double, A.Xt);
    auto memo_issue = [&](int yi, double (&rl)[NB], double (&rd)[NB]) {
        const int src = yi & (Wv::W - 1);
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            rl[t] = Wv::from_lane(tab_dl[t], src);
            rd[t] = GRAD ? Wv::from_lane(tab_dd[t], src) : 0.0;
        }
    };
    auto memo_pick = [&](int yi, const double (&rl)[NB], const double (&rd)[NB], double& dl, double& dd) {
        const int blk = yi / Wv::W;
        dl = rl[0]; dd = rd[0];
#pragma unroll
        for (int t = 1; t < NB; ++t) {
            dl = (blk == t) ? rl[t] : dl;
            dd = (blk == t) ? rd[t] : dd;
        }
    };
    int nl = Wv::lane();
    int y1 = load_y(nl), y2 = load_y(nl + Wv::W);
    double m1 = load_m(nl);
    double x1[P];
    int c1 = 0;
    if (cr_reg) {
        if constexpr (CELL) {
            c1 = DSQ_AS_GLOBAL(int32_t, cell_of)[nl < A.N ? nl : A.N - 1];
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) x1[j] = Xg[j * A.ldx + (nl < A.N ? nl : A.N - 1)];
        }
    }
    // the 4-block memo keeps 16 fetched values in flight per prefetch, which costs more registers
    // than the 168 a wave may use at 3 waves per SIMD: it fetches at the point of use instead
    constexpr bool kPrefetchMemo = NB <= 2;
    double dl1 = 0.0, dd1 = 0.0;
    if (kPrefetchMemo) {
        double rl[NB], rd[NB];
        memo_issue(y1, rl, rd);
        memo_pick(y1, rl, rd, dl1, dd1);
    }
    auto trip = [&]() {
        const int yi = y1;
        const double m = m1, dl0 = dl1, dd0 = dd1;
        double x[P];
        const int cell = c1;
        if constexpr (!CELL) {
#pragma unroll
            for (int j = 0; j < P; ++j) x[j] = x1[j];
        }
        // ---- issue the next iteration's reads
        nl += Wv::W;
        y1 = y2;
        y2 = load_y(nl + Wv::W);
        m1 = load_m(nl);
        if (cr_reg) {
            const int nx = nl < A.N ? nl : A.N - 1;  // padded / out-of-range samples have w = 0
            if constexpr (CELL) {
                c1 = DSQ_AS_GLOBAL(int32_t, cell_of)[nx];
            } else {
#pragma unroll
                for (int j = 0; j < P; ++j) x1[j] = Xg[j * A.ldx + nx];
            }
        }
        double rl[NB], rd[NB];
        memo_issue(kPrefetchMemo ? y1 : yi, rl, rd);
        // ---- this iteration
        const double yv = (double)yi;
        double dl = dl0, dd = dd0;
        if (!kPrefetchMemo) memo_pick(yi, rl, rd, dl, dd);
        // counts beyond the memo exist only when the gene's largest count is >= 256 (NB == 4);
        // HostWave's memo holds NB counts, so the host instantiation takes the general formula
        const bool in_tab = yi < tab_n;
        if ((NB == kMemoBlocks || Wv::W == 1) && Wv::any(!in_tab)) {
            double dl2, dd2;
            if (Wv::W == 1)
                lgamma_digamma_diff<Wv, GRAD>(yi, a, lga, dga, dl2, dd2);
            else
                lgamma_digamma_diff<Wv, GRAD, true>(in_tab ? 256 : yi, a, lga, dga, dl2, dd2);
            dl = in_tab ? dl : dl2;
            dd = in_tab ? dd : dd2;
        }
        const double ma = m * alpha;
        const double r1 = frcp(1.0 + ma);
        const double L1 = flog1p_t(ma, r1);
        accf.add(dl + yv * (L1 - lal) + a * L1);
        if (GRAD) accg += dd + L1 + (yv - m) * alpha * r1;
        if (cr_reg) {
            const double w = m * r1;
            const double dw = -(w * w);
            if constexpr (CELL) {
                Wv::cell_add(&Wk->acc[0][cell], w);
                if (GRAD) Wv::cell_add(&Wk->acc[1][cell], dw);
            } else {
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const double xw = x[i] * w, xdw = x[i] * dw;
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        M[tri(i, j)] += xw * x[j];
                        if (GRAD && !kSplitDM) dM[tri(i, j)] += xdw * x[j];
                    }
                }
            }
        }
        // ---- the memo values fetched above are consumed by the next iteration
        if (kPrefetchMemo) memo_pick(y1, rl, rd, dl1, dd1);
    };
    // wave-uniform trip count (all lanes stay active); DSQ_ALPHA_UNROLL trips per loop iteration, written out by
    // hand: the loop holds cross-lane reads (convergent), which the compiler will not unroll with a remainder
    for (int base = 0; base < n_end; base += DSQ_ALPHA_UNROLL * Wv::W) {
        trip();
#if DSQ_ALPHA_UNROLL >= 2
        if (base + Wv::W < n_end) trip();
#endif
    }
    if (GRAD && kSplitDM && cr_reg) {
#pragma unroll
        for (int k = 0; k < T; ++k) dM[k] = 0.0;
        // wide designs: X^T dW X in a second sweep, so that only p(p+1)/2 accumulators are live per
        // sweep (both sets together exceed the register file and spill heavily from p = 9 on)
        for (int base = 0; base < n_end; base += Wv::W) {
            const int n = base + Wv::lane();
            const double m = load_m(n);
            const double w = m * frcp(1.0 + m * alpha);
            const double dw = -(w * w);
            const int nx = n < A.N ? n : A.N - 1;
            double x[P];
#pragma unroll
            for (int j = 0; j < P; ++j) x[j] = Xg[j * A.ldx + nx];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const double xdw = x[i] * dw;
#pragma unroll
                for (int j = 0; j <= i; ++j) dM[tri(i, j)] += xdw * x[j];
            }
        }
    }
    DSQ_PHASE(4);
    const double sumf = Wv::sum_comp(accf);
    if (GRAD) accg = Wv::sum(accg);
    f = sumf + A.cst;
    g = 0.0;
    if (GRAD) g = alpha * (-(a * a * accg));
    if (cr_reg) {
        if constexpr (CELL) {
            // entry-parallel: lane e owns entry e of X^T W X (and of X^T dW X) and walks the cells
            Wv::sync();
            const CellDesign& D = A.cell->D;
            const auto XXg = DSQ_AS_LDS(double, D.XX);  // the kernel stages the cells' tables in LDS
            for (int e = Wv::lane(); e < T; e += Wv::W) {
                double me = 0.0, de = 0.0;
                for (int c = 0; c < D.C; ++c) {
                    const double xx = XXg[c * T + e];
                    me += xx * Wk->acc[0][c];
                    if (GRAD) de += xx * Wk->acc[1][c];
                }
                Wk->ent[e] = me;
                if (GRAD) Wk->ent[T + e] = de;
            }
            Wv::sync();
#pragma unroll
            for (int k = 0; k < T; ++k) {
                M[k] = Wk->ent[k];
                if (GRAD) dM[k] = Wk->ent[T + k];
            }
        } else {
            Wv::template sum_n<T>(M);
            if (GRAD) Wv::template sum_n<T>(dM);
        }
        DSQ_PHASE(5);
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
    return EvalOut{f, g};
}

template <class Wv, int P, bool GRAD, bool PAD = false, int NB = 1, bool CELL = false>
DSQ_EVAL_FN EvalOut alpha_eval_v(const AlphaArgs A, double la, bool cr_reg, bool prior_reg) {
    return alpha_eval_body<Wv, P, GRAD, PAD, NB, CELL>(A, la, cr_reg, prior_reg);
}

template <class Wv, int P, bool GRAD, bool PAD = false, int NB = 1, bool CELL = false>
DSQ_HD void alpha_eval(const AlphaArgs& A, double la, bool cr_reg, bool prior_reg, double& f, double& g) {
    EvalOut o;
    if constexpr (P >= DSQ_EVAL_INLINE_MIN_P && CELL) o = alpha_eval_body<Wv, P, GRAD, PAD, NB, CELL>(A, la, cr_reg, prior_reg);
    else o = alpha_eval_v<Wv, P, GRAD, PAD, NB, CELL>(A, la, cr_reg, prior_reg);
    f = o.f;
    g = o.g;
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
        c.add(lg - yv * flog_t(m));  // (every kernel that gets here has filled the table: k_alpha*, the wide and BFGS ones)
    }
    return Wv::sum_comp(c);
}

// alpha_const and the gene's largest count in ONE pass, the rows fetched four samples per lane at a time (the grid
// fallback's wavefronts walk global rows alone: a pass is a chain of load latencies, so it pays to have fewer passes
// and more loads in flight).  Same terms in the same order per lane as alpha_const: the same constant, bit for bit.
template <class Wv>
DSQ_HD double alpha_const_max(const int32_t* y, const double* mu, int N, int& max_count) {
    constexpr int U = 4;
    KSum c;
    int mx = 0;
    for (int base = 0; base < N; base += Wv::W * U) {
        int yi[U];
        double m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = base + u * Wv::W + Wv::lane();
            const int nn = n < N ? n : N - 1;
            yi[u] = y[nn];
            m[u] = mu[nn];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * Wv::W >= N) break;  // (alpha_const stops at the last partial group of W samples, too)
            const bool valid = base + u * Wv::W + Wv::lane() < N;
            const int yv_i = valid ? yi[u] : 0;
            const double yv = (double)yv_i;
            const double mm = valid ? m[u] : 1.0;
            mx = yv_i > mx ? yv_i : mx;
            const bool in_tab = yv_i < kLgammaIntN;
            double lg = kLgammaInt[in_tab ? yv_i : 0];
            if (Wv::any(!in_tab)) {
                const double z = in_tab ? 300.0 : yv + 1.0;
                const double lz = flog(z);
                const double big = (z - 0.5) * lz - z + kHalfLog2Pi + stirling_tail(frcp(z));
                if (!in_tab) lg = big;
            }
            c.add(lg - yv * flog_t(mm));
        }
    }
    max_count = Wv::maxi(mx);
    return Wv::sum_comp(c);
}

// one gene: L-BFGS-B in log(alpha) from log(alpha_hat).  RUN_GRID: on non-convergence run the
// reference's grid search right here (host simulation / single-kernel use); otherwise only report
// converged = 0 and the caller schedules grid_alpha_gene for the gene (device: second tiny kernel,
// which keeps the 200-evaluation grid code out of the main kernel's register budget).
template <class Wv, int P, bool RUN_GRID, bool PAD = false, bool CELL = false>
DSQ_HD AlphaOut fit_alpha_gene(const int32_t* y, const double* mu, const double* Xt, int ldx, int N,
                               double alpha_hat, double min_disp, double max_disp,
                               double prior_var, bool cr_reg, bool prior_reg, Lbfgsb1d& m,
                               const double* cst_in = nullptr, double* cst_out = nullptr,
                               int memo_blocks = 1, const CellCtx* cell = nullptr, int eval_cap = 0,
                               bool resume = false) {
    // PAD rows are LDS-staged; the in-place grid search evaluates with PAD = false (global rows)
    static_assert(!(RUN_GRID && PAD), "the in-place grid search expects un-staged rows");
    AlphaArgs A;
    A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.la_hat = log(alpha_hat);
    A.prior_var = prior_var;
    A.cell = cell;
    // the constant depends on (y, mu) only: the MAP fit re-uses the one the MLE fit stored
    DSQ_PHASE(1);
    A.cst = cst_in != nullptr ? *cst_in : alpha_const<Wv>(y, mu, N);
    if (cst_out != nullptr && Wv::lane() == 0) *cst_out = A.cst;
    const double lo = log(min_disp), hi = log(max_disp);
    // eval_cap > 0: stop after that many evaluations in THIS call with the optimiser's state left in `m` (o.status = -1):
    // the caller resumes the gene later (resume = true: `m` holds that state).  The sequence of iterates is unchanged.
    if (!resume) m.start(A.la_hat, lo, hi);
    int budget = eval_cap > 0 ? eval_cap : 0x7fffffff;
    while (!m.done && budget > 0) {
        --budget;
        double f, g;
        if (memo_blocks <= 1) alpha_eval<Wv, P, true, PAD, 1, CELL>(A, m.x, cr_reg, prior_reg, f, g);
        else if (memo_blocks == 2) alpha_eval<Wv, P, true, PAD, 2, CELL>(A, m.x, cr_reg, prior_reg, f, g);
        else alpha_eval<Wv, P, true, PAD, 4, CELL>(A, m.x, cr_reg, prior_reg, f, g);
        DSQ_PHASE(6);
        m.feed(f, g);
    }
    DSQ_PHASE(7);
    AlphaOut o;
    o.converged = m.success ? 1 : 0;
    o.nfev = m.nfev; o.nit = m.it; o.status = m.done ? m.status : -1;
    o.alpha = exp(m.x);
    if (RUN_GRID && m.done && !m.success) {
        A.cell = nullptr;  // the (rare) grid search runs the general evaluation
        o.alpha = exp(grid_fit_alpha<Wv, P>(A, lo, hi));
    }
    return o;
}

// optimizer="BFGS" (utils.py:546-554): scipy's unbounded BFGS in log(alpha) from log(alpha_hat); on success = False
// the same grid search as above.  General evaluation on un-staged rows (not a hot path: no caller in dds.py selects it).
template <class Wv, int P, bool RUN_GRID>
DSQ_HD AlphaOut fit_alpha_gene_bfgs(const int32_t* y, const double* mu, const double* Xt, int ldx, int N,
                                    double alpha_hat, double min_disp, double max_disp, double prior_var,
                                    bool cr_reg, bool prior_reg, int memo_blocks) {
    AlphaArgs A;
    A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.la_hat = log(alpha_hat);
    A.prior_var = prior_var;
    A.cell = nullptr;
    A.cst = alpha_const<Wv>(y, mu, N);
    auto fg = [&](const double* x, double& f, double* g) {
        if (memo_blocks <= 1) alpha_eval<Wv, P, true, false, 1>(A, x[0], cr_reg, prior_reg, f, g[0]);
        else if (memo_blocks == 2) alpha_eval<Wv, P, true, false, 2>(A, x[0], cr_reg, prior_reg, f, g[0]);
        else alpha_eval<Wv, P, true, false, 4>(A, x[0], cr_reg, prior_reg, f, g[0]);
    };
    BfgsWork<1> W;
    double x = A.la_hat;
    const BfgsResult r = bfgs_min<1>(fg, 1, &x, W);
    AlphaOut o;
    o.converged = r.success ? 1 : 0;
    o.nfev = r.nfev; o.nit = r.nit; o.status = r.status;
    o.alpha = exp(x);
    if (RUN_GRID && !r.success) o.alpha = exp(grid_fit_alpha<Wv, P>(A, log(min_disp), log(max_disp)));
    return o;
}

// grid-search fallback of one gene (utils.py:556-564)
template <class Wv, int P>
DSQ_HD double grid_alpha_gene(const int32_t* y, const double* mu, const double* Xt, int ldx, int N,
                              double min_disp, double max_disp) {
    AlphaArgs A;
    A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.la_hat = 0.0; A.prior_var = 1.0;
    A.cst = alpha_const<Wv>(y, mu, N);
    return exp(grid_fit_alpha<Wv, P>(A, log(min_disp), log(max_disp)));
}

}  // namespace dsq
