// dsq_lbfgsb.h — bound-constrained limited-memory BFGS (L-BFGS-B) for n <= NMAX variables.
//
// The reference rescues a diverging IRLS with
//   scipy.optimize.minimize(f, beta_init, jac=df, method="L-BFGS-B", bounds=[(-30, 30)]*p)
// (pydeseq2/utils.py:374-403) and keeps whatever iterate that optimiser stops at — for
// low-count genes this is usually NOT the optimum (objective has kinks at mu = min_mu,
// the relative-reduction test fires early, or the line search aborts).  To return the
// same numbers this file restates the published algorithm that scipy ships
// (L-BFGS-B 3.0, Byrd-Lu-Nocedal-Zhu 1995; Morales-Nocedal 2011; scipy 1.15.3 is a C
// translation of it) routine by routine: active / projgr / cauchy (+ hpsolb) / freev /
// formk / cmprlb / subsm / lnsrlb (+ MINPACK-2 dcsrch, dsq_lbfgsb1d.h) / matupd / formt /
// bmv, and LINPACK dpofa / dtrsl, with scipy's defaults m = 10, factr = 1e7,
// pgtol = 1e-5, maxls = 20, maxiter = maxfun = 15000.
//
// Indices below are 1-based through small accessor helpers so the control flow can be
// compared line by line with the published algorithm.  All state lives in a caller
// supplied workspace (wave-private LDS on the device; every lane executes the same
// scalar code on it, function/gradient evaluations are the only wave-parallel part).
#pragma once
#include "dsq_lbfgsb1d.h"
#include "dsq_lbfgsb_par.h"

namespace dsq {

template <int NMAX, int M = 10>
struct LbfgsbWork {
    double ws[NMAX * M], wy[NMAX * M];
    double sy[M * M], ss[M * M], wt[M * M];
    double wn[4 * M * M], wn1[4 * M * M];
    double z[NMAX], r[NMAX], d[NMAX], t[NMAX], xp[NMAX], wa[8 * M];
    double g[NMAX];
    double sacc[2 * M];  // per-lane accumulators of the lane-parallel factorisations / solves (dsq_lbfgsb_par.h)
    int index[NMAX], iwhere[NMAX], indx2[NMAX];
};

struct LbfgsbResult {
    double f;
    bool success;
    int nfev, nit;
    int status;  // 0/1 pgtol, 2 ftol, 3 abnormal, 4 maxiter/maxfun
};

namespace lb {

// LINPACK dpofa on the upper triangle of a (leading dimension lda), 1-based (i,j)
DSQ_HD int dpofa(double* a, int lda, int n) {
#define A_(i, j) a[((i)-1) + ((j)-1) * lda]
    for (int j = 1; j <= n; ++j) {
        double s = 0.0;
        for (int k = 1; k <= j - 1; ++k) {
            double t = A_(k, j);
            for (int q = 1; q <= k - 1; ++q) t -= A_(q, k) * A_(q, j);
            t = t / A_(k, k);
            A_(k, j) = t;
            s += t * t;
        }
        s = A_(j, j) - s;
        if (s <= 0.0) return j;
        A_(j, j) = sqrt(s);
    }
    return 0;
#undef A_
}

// LINPACK dtrsl for upper-triangular t: job 01 solves t x = b, job 11 solves t' x = b
DSQ_HD int dtrsl_upper(const double* t, int ldt, int n, double* b, int job) {
#define T_(i, j) t[((i)-1) + ((j)-1) * ldt]
    for (int j = 1; j <= n; ++j)
        if (T_(j, j) == 0.0) return j;
    if (job == 1) {  // t x = b
        b[n - 1] = b[n - 1] / T_(n, n);
        for (int jj = 2; jj <= n; ++jj) {
            const int j = n - jj + 1;
            const double temp = -b[j];  // b(j+1)
            for (int q = 1; q <= j; ++q) b[q - 1] += temp * T_(q, j + 1);
            b[j - 1] = b[j - 1] / T_(j, j);
        }
    } else {  // t' x = b
        b[0] = b[0] / T_(1, 1);
        for (int j = 2; j <= n; ++j) {
            double s = 0.0;
            for (int q = 1; q <= j - 1; ++q) s += T_(q, j) * b[q - 1];
            b[j - 1] = b[j - 1] - s;
            b[j - 1] = b[j - 1] / T_(j, j);
        }
    }
    return 0;
#undef T_
}

// heap maintenance of the breakpoints (hpsolb)
DSQ_HD void hpsolb(int n, double* t, int* iorder, int iheap) {
    if (iheap == 0) {
        for (int k = 2; k <= n; ++k) {
            const double ddum = t[k - 1];
            const int indxin = iorder[k - 1];
            int i = k;
            while (i > 1) {
                const int j = i / 2;
                if (ddum < t[j - 1]) {
                    t[i - 1] = t[j - 1];
                    iorder[i - 1] = iorder[j - 1];
                    i = j;
                } else {
                    break;
                }
            }
            t[i - 1] = ddum;
            iorder[i - 1] = indxin;
        }
    }
    if (n > 1) {
        int i = 1;
        const double out = t[0];
        const int indxou = iorder[0];
        const double ddum = t[n - 1];
        const int indxin = iorder[n - 1];
        for (;;) {
            int j = i + i;
            if (j <= n - 1) {
                if (t[j] < t[j - 1]) j = j + 1;
                if (t[j - 1] < ddum) {
                    t[i - 1] = t[j - 1];
                    iorder[i - 1] = iorder[j - 1];
                    i = j;
                    continue;
                }
            }
            break;
        }
        t[i - 1] = ddum;
        iorder[i - 1] = indxin;
        t[n - 1] = out;
        iorder[n - 1] = indxou;
    }
}

}  // namespace lb

// FG: void(const double* x, double& f, double* g)
// Wv: how the dense linear algebra between two evaluations is executed (dsq_lbfgsb_par.h).  OneLane: every lane of the
// calling wavefront runs all of it redundantly (the scalar routines as written); a wave policy (DeviceWave): the outputs of
// formk / subsm / matupd / formt are spread over the lanes, each with its scalar arithmetic - same iterates.  The rare
// paths (Cauchy point with its bmv products, projections at bounds) stay scalar in either mode.
template <int NMAX, class FG, int M = 10, class Wv = OneLane>
DSQ_HD LbfgsbResult lbfgsb_nd(FG&& fg, int n, double* x, const double* l, const double* u,
                              const int* nbd, LbfgsbWork<NMAX, M>& W, double factr = 1e7,
                              double pgtol = 1e-5, int maxls = 20, int maxiter = 15000,
                              int maxfun = 15000) {
    constexpr int m = M;
    constexpr int m2 = 2 * M;
    const double epsmch = kEps;
    const double tol = factr * epsmch;
#define WS(i, j) W.ws[((i)-1) + ((j)-1) * NMAX]
#define WY(i, j) W.wy[((i)-1) + ((j)-1) * NMAX]
#define SY(i, j) W.sy[((i)-1) + ((j)-1) * m]
#define SS(i, j) W.ss[((i)-1) + ((j)-1) * m]
#define WT(i, j) W.wt[((i)-1) + ((j)-1) * m]
#define WN(i, j) W.wn[((i)-1) + ((j)-1) * m2]
#define WN1(i, j) W.wn1[((i)-1) + ((j)-1) * m2]
    double* g = W.g;
    double* z = W.z;
    double* r = W.r;
    double* d = W.d;
    double* t = W.t;
    double* xp = W.xp;
    double* wa = W.wa;
    int* index = W.index;
    int* iwhere = W.iwhere;
    int* indx2 = W.indx2;

    {   // scipy hands setulb zero-initialised work arrays (numpy.zeros in _minimize_lbfgsb) and the routine, as written in
        // Fortran, reads entries of them it has not written yet (bounded problems: seen on a gene whose IRLS rescue changed
        // its iterates from pass to pass with whatever the previous workgroup had left in LDS): they are 0 there, and here
#if !defined(DSQ_LBFGSB_NO_ZERO)  // (developer builds only: does an input notice what the workspace held?)
        uint32_t* wz = (uint32_t*)&W;
#if defined(DSQ_LBFGSB_POISON)     // (developer builds only, -DDSQ_LBFGSB_POISON=0xFFFFFFFFu: that word instead of zeros - a
                                   // result that changes has read an entry before writing it)
        for (int i = Wv::lane(); i < (int)(sizeof(W) / 4); i += Wv::W) wz[i] = (uint32_t)(DSQ_LBFGSB_POISON);
#else
        for (int i = Wv::lane(); i < (int)(sizeof(W) / 4); i += Wv::W) wz[i] = 0u;
#endif
        Wv::sync();
#endif
    }
    LbfgsbResult R;
    int col = 0, head = 1, itail = 0, iupdat = 0, iter = 0, nfev = 0, nfree = n, nenter = 0,
        ileave = 0, nseg = 0;
    double theta = 1.0, f = 0.0, fold = 0.0, gd = 0.0, gdold = 0.0, stp = 0.0, dtd = 0.0,
           sbgnrm = 0.0;
    bool updatd = false, wrk = false;
    (void)nseg; (void)itail;

    // ---- active
    bool cnstnd = false, boxed = true;
    for (int i = 1; i <= n; ++i) {
        if (nbd[i - 1] > 0) {
            if (nbd[i - 1] <= 2 && x[i - 1] <= l[i - 1]) {
                if (x[i - 1] < l[i - 1]) x[i - 1] = l[i - 1];
            } else if (nbd[i - 1] >= 2 && x[i - 1] >= u[i - 1]) {
                if (x[i - 1] > u[i - 1]) x[i - 1] = u[i - 1];
            }
        }
    }
    for (int i = 1; i <= n; ++i) {
        if (nbd[i - 1] != 2) boxed = false;
        if (nbd[i - 1] == 0) {
            iwhere[i - 1] = -1;
        } else {
            cnstnd = true;
            if (nbd[i - 1] == 2 && u[i - 1] - l[i - 1] <= 0.0) iwhere[i - 1] = 3;
            else iwhere[i - 1] = 0;
        }
    }
    auto projgr = [&]() {
        double s = 0.0;
        for (int i = 1; i <= n; ++i) {
            double gi = g[i - 1];
            if (nbd[i - 1] != 0) {
                if (gi < 0.0) {
                    if (nbd[i - 1] >= 2) gi = dmax(x[i - 1] - u[i - 1], gi);
                } else {
                    if (nbd[i - 1] <= 2) gi = dmin(x[i - 1] - l[i - 1], gi);
                }
            }
            s = dmax(s, fabs(gi));
        }
        return s;
    };
    // product of the 2m x 2m middle matrix with v -> p  (bmv)
    auto bmv = [&](const double* v, double* p) -> int {
        if (col == 0) return 0;
        p[col] = v[col];
        for (int i = 2; i <= col; ++i) {
            const int i2 = col + i;
            double sum = 0.0;
            for (int k = 1; k <= i - 1; ++k) sum += SY(i, k) * v[k - 1] / SY(k, k);
            p[i2 - 1] = v[i2 - 1] + sum;
        }
        int info = lb::dtrsl_upper(W.wt, m, col, p + col, 11);
        if (info != 0) return info;
        for (int i = 1; i <= col; ++i) p[i - 1] = v[i - 1] / sqrt(SY(i, i));
        info = lb::dtrsl_upper(W.wt, m, col, p + col, 1);
        if (info != 0) return info;
        for (int i = 1; i <= col; ++i) p[i - 1] = -p[i - 1] / sqrt(SY(i, i));
        for (int i = 1; i <= col; ++i) {
            double sum = 0.0;
            for (int k = i + 1; k <= col; ++k) sum += SY(k, i) * p[col + k - 1] / SY(i, i);
            p[i - 1] += sum;
        }
        return 0;
    };
    auto refresh = [&]() {
        col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = false;
    };

    fg(x, f, g);
    nfev = 1;
    sbgnrm = projgr();
    if (sbgnrm <= pgtol) {
        R = {f, true, nfev, 0, 0};
        return R;
    }

    for (;;) {  // ---------------------------------------------------------------- iterations
        bool done_dir = false;
        while (!done_dir) {  // label 222
            int info = 0;
            if (!cnstnd && col > 0) {
                Wv::sync();
                for (int i = Wv::lane(); i < n; i += Wv::W) z[i] = x[i];
                Wv::sync();
                wrk = updatd;
                nseg = 0;
            } else {
                // ------------------------------------------------------------ cauchy
                double* p = wa;            // wa(1)
                double* c = wa + 2 * m;    // wa(2m+1)
                double* wbp = wa + 4 * m;  // wa(4m+1)
                double* v = wa + 6 * m;    // wa(6m+1)
                double* xcp = z;
                int* iorder = indx2;
                double* tb = t;  // breakpoint times
                do {
                    if (sbgnrm <= 0.0) {
                        for (int i = 0; i < n; ++i) xcp[i] = x[i];
                        break;
                    }
                    bool bnded = true;
                    int nfree_c = n + 1, nbreak = 0, ibkmin = 0;
                    double bkmin = 0.0;
                    const int col2 = 2 * col;
                    double f1 = 0.0;
                    for (int i = 0; i < col2; ++i) p[i] = 0.0;
                    for (int i = 1; i <= n; ++i) {
                        const double neggi = -g[i - 1];
                        double tl = 0.0, tu = 0.0;
                        if (iwhere[i - 1] != 3 && iwhere[i - 1] != -1) {
                            if (nbd[i - 1] <= 2) tl = x[i - 1] - l[i - 1];
                            if (nbd[i - 1] >= 2) tu = u[i - 1] - x[i - 1];
                            const bool xlower = nbd[i - 1] <= 2 && tl <= 0.0;
                            const bool xupper = nbd[i - 1] >= 2 && tu <= 0.0;
                            iwhere[i - 1] = 0;
                            if (xlower) {
                                if (neggi <= 0.0) iwhere[i - 1] = 1;
                            } else if (xupper) {
                                if (neggi >= 0.0) iwhere[i - 1] = 2;
                            } else {
                                if (fabs(neggi) <= 0.0) iwhere[i - 1] = -3;
                            }
                        }
                        int pointr = head;
                        if (iwhere[i - 1] != 0 && iwhere[i - 1] != -1) {
                            d[i - 1] = 0.0;
                        } else {
                            d[i - 1] = neggi;
                            f1 -= neggi * neggi;
                            for (int j = 1; j <= col; ++j) {
                                p[j - 1] += WY(i, pointr) * neggi;
                                p[col + j - 1] += WS(i, pointr) * neggi;
                                pointr = pointr % m + 1;
                            }
                            if (nbd[i - 1] <= 2 && nbd[i - 1] != 0 && neggi < 0.0) {
                                nbreak += 1;
                                iorder[nbreak - 1] = i;
                                tb[nbreak - 1] = tl / (-neggi);
                                if (nbreak == 1 || tb[nbreak - 1] < bkmin) { bkmin = tb[nbreak - 1]; ibkmin = nbreak; }
                            } else if (nbd[i - 1] >= 2 && neggi > 0.0) {
                                nbreak += 1;
                                iorder[nbreak - 1] = i;
                                tb[nbreak - 1] = tu / neggi;
                                if (nbreak == 1 || tb[nbreak - 1] < bkmin) { bkmin = tb[nbreak - 1]; ibkmin = nbreak; }
                            } else {
                                nfree_c -= 1;
                                iorder[nfree_c - 1] = i;
                                if (fabs(neggi) > 0.0) bnded = false;
                            }
                        }
                    }
                    if (theta != 1.0)
                        for (int i = 0; i < col; ++i) p[col + i] *= theta;
                    for (int i = 0; i < n; ++i) xcp[i] = x[i];
                    if (nbreak == 0 && nfree_c == n + 1) break;
                    for (int j = 0; j < col2; ++j) c[j] = 0.0;
                    double f2 = -theta * f1;
                    const double f2_org = f2;
                    if (col > 0) {
                        info = bmv(p, v);
                        if (info != 0) break;
                        double s = 0.0;
                        for (int j = 0; j < col2; ++j) s += v[j] * p[j];
                        f2 -= s;
                    }
                    double dtm = -f1 / f2;
                    double tsum = 0.0;
                    nseg = 1;
                    bool goto999 = false;
                    if (nbreak != 0) {
                        int nleft = nbreak, it2 = 1;
                        double tj = 0.0;
                        for (;;) {  // 777
                            const double tj0 = tj;
                            int ibp;
                            if (it2 == 1) {
                                tj = bkmin;
                                ibp = iorder[ibkmin - 1];
                            } else {
                                if (it2 == 2) {
                                    if (ibkmin != nbreak) {
                                        tb[ibkmin - 1] = tb[nbreak - 1];
                                        iorder[ibkmin - 1] = iorder[nbreak - 1];
                                    }
                                }
                                lb::hpsolb(nleft, tb, iorder, it2 - 2);
                                tj = tb[nleft - 1];
                                ibp = iorder[nleft - 1];
                            }
                            const double dt = tj - tj0;
                            if (dtm < dt) break;  // goto 888
                            tsum += dt;
                            nleft -= 1;
                            it2 += 1;
                            const double dibp = d[ibp - 1];
                            d[ibp - 1] = 0.0;
                            double zibp;
                            if (dibp > 0.0) {
                                zibp = u[ibp - 1] - x[ibp - 1];
                                xcp[ibp - 1] = u[ibp - 1];
                                iwhere[ibp - 1] = 2;
                            } else {
                                zibp = l[ibp - 1] - x[ibp - 1];
                                xcp[ibp - 1] = l[ibp - 1];
                                iwhere[ibp - 1] = 1;
                            }
                            if (nleft == 0 && nbreak == n) {
                                dtm = dt;
                                goto999 = true;
                                break;
                            }
                            nseg += 1;
                            const double dibp2 = dibp * dibp;
                            f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
                            f2 = f2 - theta * dibp2;
                            if (col > 0) {
                                for (int j = 0; j < col2; ++j) c[j] += dt * p[j];
                                int pointr = head;
                                for (int j = 1; j <= col; ++j) {
                                    wbp[j - 1] = WY(ibp, pointr);
                                    wbp[col + j - 1] = theta * WS(ibp, pointr);
                                    pointr = pointr % m + 1;
                                }
                                info = bmv(wbp, v);
                                if (info != 0) break;
                                double wmc = 0.0, wmp = 0.0, wmw = 0.0;
                                for (int j = 0; j < col2; ++j) { wmc += c[j] * v[j]; wmp += p[j] * v[j]; wmw += wbp[j] * v[j]; }
                                for (int j = 0; j < col2; ++j) p[j] += -dibp * wbp[j];
                                f1 = f1 + dibp * wmc;
                                f2 = f2 + 2.0 * dibp * wmp - dibp2 * wmw;
                            }
                            f2 = dmax(epsmch * f2_org, f2);
                            if (nleft > 0) {
                                dtm = -f1 / f2;
                                continue;
                            } else if (bnded) {
                                f1 = 0.0; f2 = 0.0; dtm = 0.0;
                            } else {
                                dtm = -f1 / f2;
                            }
                            break;
                        }
                        if (info != 0) break;
                    }
                    if (!goto999) {  // 888
                        if (dtm <= 0.0) dtm = 0.0;
                        tsum += dtm;
                        for (int i = 0; i < n; ++i) xcp[i] += tsum * d[i];
                    }
                    if (col > 0)  // 999
                        for (int j = 0; j < col2; ++j) c[j] += dtm * p[j];
                } while (false);
                if (info != 0) { refresh(); continue; }
                // ------------------------------------------------------------ freev
                nenter = 0;
                ileave = n + 1;
                if (iter > 0 && cnstnd) {
                    for (int i = 1; i <= nfree; ++i) {
                        const int k = index[i - 1];
                        if (iwhere[k - 1] > 0) { ileave -= 1; indx2[ileave - 1] = k; }
                    }
                    for (int i = 1 + nfree; i <= n; ++i) {
                        const int k = index[i - 1];
                        if (iwhere[k - 1] <= 0) { nenter += 1; indx2[nenter - 1] = k; }
                    }
                }
                wrk = (ileave < n + 1) || (nenter > 0) || updatd;
                nfree = 0;
                int iact = n + 1;
                for (int i = 1; i <= n; ++i) {
                    if (iwhere[i - 1] <= 0) { nfree += 1; index[nfree - 1] = i; }
                    else { iact -= 1; index[iact - 1] = i; }
                }
            }
            // ---------------------------------------------------------------- 333
            if (nfree != 0 && col != 0) {
                if (wrk) {
                    // -------------------------------------------------------- formk
                    // (lane-parallel: every output entry keeps the scalar routine's operations and their order; see
                    // dsq_lbfgsb_par.h.  The barriers separate steps whose inputs other lanes wrote.)
                    Wv::sync();
                    if (updatd) {
                        if (iupdat > m) {
                            // the three blocks of WN1 move one row up and one column left: new(r, c) = old(r + 1, c + 1) -
                            // the sequential loops never read an entry they have already overwritten, so this is a
                            // simultaneous move: read, barrier, write
                            constexpr int NSH = ((M - 1) * M + (M - 1) * (M - 1) + Wv::W - 1) / Wv::W;
                            double tmp[NSH];
                            auto walk = [&](auto&& fn) {
                                int e = 0, slot = 0;
                                for (int jy = 1; jy <= m - 1; ++jy) {
                                    const int js = m + jy;
                                    for (int q = 0; q < m - jy; ++q, ++e)
                                        if (e % Wv::W == Wv::lane()) fn(slot++, jy + q, jy, jy + 1 + q, jy + 1);
                                    for (int q = 0; q < m - jy; ++q, ++e)
                                        if (e % Wv::W == Wv::lane()) fn(slot++, js + q, js, js + 1 + q, js + 1);
                                    for (int q = 0; q < m - 1; ++q, ++e)
                                        if (e % Wv::W == Wv::lane()) fn(slot++, m + 1 + q, jy, m + 2 + q, jy + 1);
                                }
                            };
                            if constexpr (Wv::W == 1) {
                                (void)tmp;
                                walk([&](int, int dr_, int dc_, int sr_, int sc_) { WN1(dr_, dc_) = WN1(sr_, sc_); });
                            } else {
#pragma unroll
                                for (int i = 0; i < NSH; ++i) tmp[i] = 0.0;
                                walk([&](int slot, int, int, int sr_, int sc_) {
#pragma unroll
                                    for (int i = 0; i < NSH; ++i)
                                        if (i == slot) tmp[i] = WN1(sr_, sc_);
                                });
                                Wv::sync();
                                walk([&](int slot, int dr_, int dc_, int, int) {
                                    double v = 0.0;
#pragma unroll
                                    for (int i = 0; i < NSH; ++i)
                                        if (i == slot) v = tmp[i];
                                    WN1(dr_, dc_) = v;
                                });
                                Wv::sync();
                            }
                        }
                        const int pbegin = 1, pend = nfree, dbegin = nfree + 1, dend = n;
                        const int iy = col, is0 = m + col;
                        int ipntr = head + col - 1;
                        if (ipntr > m) ipntr -= m;
                        for (int jy = 1 + Wv::lane(); jy <= col; jy += Wv::W) {  // row `col` of the Y'Y, S'S and R blocks
                            const int js = m + jy;
                            const int jpntr = (head + jy - 2) % m + 1;
                            double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0;
                            for (int k = pbegin; k <= pend; ++k) { const int k1 = index[k - 1]; temp1 += WY(k1, ipntr) * WY(k1, jpntr); }
                            for (int k = dbegin; k <= dend; ++k) {
                                const int k1 = index[k - 1];
                                temp2 += WS(k1, ipntr) * WS(k1, jpntr);
                                temp3 += WS(k1, ipntr) * WY(k1, jpntr);
                            }
                            WN1(iy, jy) = temp1;
                            WN1(is0, js) = temp2;
                            WN1(is0, jy) = temp3;
                        }
                        Wv::sync();  // (the next loop overwrites WN1(m + col, col))
                        const int jyc = col;
                        int jpntr = head + col - 1;
                        if (jpntr > m) jpntr -= m;
                        for (int i = 1 + Wv::lane(); i <= col; i += Wv::W) {  // column `col` of the R block
                            const int is = m + i;
                            const int ip = (head + i - 2) % m + 1;
                            double temp3 = 0.0;
                            for (int k = pbegin; k <= pend; ++k) { const int k1 = index[k - 1]; temp3 += WS(k1, ip) * WY(k1, jpntr); }
                            WN1(is, jyc) = temp3;
                        }
                        Wv::sync();
                    }
                    const int upcl = updatd ? col - 1 : col;
                    {
                        // variables that entered / left the free set since the last iteration (none without bounds: the
                        // sums are empty, the entries still take their "+ 0 - 0")
                        for (int e = Wv::lane(); e < upcl * upcl; e += Wv::W) {
                            const int iy = e / upcl + 1, jy = e % upcl + 1;
                            if (jy > iy) continue;
                            const int is = m + iy, js = m + jy;
                            const int ipntr = (head + iy - 2) % m + 1, jpntr = (head + jy - 2) % m + 1;
                            double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0, temp4 = 0.0;
                            for (int k = 1; k <= nenter; ++k) {
                                const int k1 = indx2[k - 1];
                                temp1 += WY(k1, ipntr) * WY(k1, jpntr);
                                temp2 += WS(k1, ipntr) * WS(k1, jpntr);
                            }
                            for (int k = ileave; k <= n; ++k) {
                                const int k1 = indx2[k - 1];
                                temp3 += WY(k1, ipntr) * WY(k1, jpntr);
                                temp4 += WS(k1, ipntr) * WS(k1, jpntr);
                            }
                            WN1(iy, jy) = WN1(iy, jy) + temp1 - temp3;
                            WN1(is, js) = WN1(is, js) - temp2 + temp4;
                        }
                        for (int e = Wv::lane(); e < upcl * upcl; e += Wv::W) {
                            const int is = m + e / upcl + 1, jy = e % upcl + 1;
                            const int ipntr = (head + (is - m) - 2) % m + 1, jpntr = (head + jy - 2) % m + 1;
                            double temp1 = 0.0, temp3 = 0.0;
                            for (int k = 1; k <= nenter; ++k) { const int k1 = indx2[k - 1]; temp1 += WS(k1, ipntr) * WY(k1, jpntr); }
                            for (int k = ileave; k <= n; ++k) { const int k1 = indx2[k - 1]; temp3 += WS(k1, ipntr) * WY(k1, jpntr); }
                            if (is <= jy + m) WN1(is, jy) = WN1(is, jy) + temp1 - temp3;
                            else WN1(is, jy) = WN1(is, jy) - temp1 + temp3;
                        }
                        Wv::sync();
                    }
                    for (int iy = 1 + Wv::lane(); iy <= col; iy += Wv::W) {  // the columns iy and col + iy of WN
                        const int is = col + iy, is1 = m + iy;
                        for (int jy = 1; jy <= iy; ++jy) {
                            const int js = col + jy, js1 = m + jy;
                            WN(jy, iy) = WN1(iy, jy) / theta;
                            WN(js, is) = WN1(is1, js1) * theta;
                        }
                        for (int jy = 1; jy <= iy - 1; ++jy) WN(jy, is) = -WN1(is1, jy);
                        for (int jy = iy; jy <= col; ++jy) WN(jy, is) = WN1(is1, jy);
                        WN(iy, iy) = WN(iy, iy) + SY(iy, iy);
                    }
                    Wv::sync();
                    info = lbp::dpofa<Wv>(W.wn, m2, col, W.sacc);
                    if (info != 0) {
                        info = -1;
                    } else {
                        const int col2 = 2 * col;
                        // col right-hand sides, one per lane (each solve is the scalar one)
                        for (int js = col + 1 + Wv::lane(); js <= col2; js += Wv::W) lbp::dtrsl_upper_t_own(W.wn, m2, col, &WN(1, js));
                        Wv::sync();
                        for (int e = Wv::lane(); e < col * col; e += Wv::W) {  // Schur complement, one entry per lane
                            const int is = col + 1 + e / col, js = col + 1 + e % col;
                            if (js < is) continue;
                            double sacc_ = 0.0;
                            for (int q = 1; q <= col; ++q) sacc_ += WN(q, is) * WN(q, js);
                            WN(is, js) = WN(is, js) + sacc_;
                        }
                        Wv::sync();
                        info = lbp::dpofa<Wv>(&WN(col + 1, col + 1), m2, col, W.sacc);
                        if (info != 0) info = -2;
                    }
                }
                if (info != 0) { refresh(); continue; }
                // ------------------------------------------------------------ cmprlb
                if (!cnstnd && col > 0) {
                    Wv::sync();
                    for (int i = Wv::lane(); i < n; i += Wv::W) r[i] = -g[i];
                    Wv::sync();
                } else {
                    for (int i = 1; i <= nfree; ++i) {
                        const int k = index[i - 1];
                        r[i - 1] = -theta * (z[k - 1] - x[k - 1]) - g[k - 1];
                    }
                    info = bmv(wa + 2 * m, wa);
                    if (info != 0) {
                        info = -8;
                    } else {
                        int pointr = head;
                        for (int j = 1; j <= col; ++j) {
                            const double a1 = wa[j - 1], a2 = theta * wa[col + j - 1];
                            for (int i = 1; i <= nfree; ++i) {
                                const int k = index[i - 1];
                                r[i - 1] = r[i - 1] + WY(k, pointr) * a1 + WS(k, pointr) * a2;
                            }
                            pointr = pointr % m + 1;
                        }
                    }
                }
                // ------------------------------------------------------------ subsm
                if (info == 0 && nfree > 0) {
                    const int nsub = nfree;
                    double* wv = wa;
                    double* ds = r;   // Newton direction on the free variables
                    double* xs = z;   // Cauchy point in, subspace minimiser out
                    Wv::sync();
                    for (int i = 1 + Wv::lane(); i <= col; i += Wv::W) {  // W' r: one pair of entries per lane
                        const int pointr = (head + i - 2) % m + 1;
                        double temp1 = 0.0, temp2 = 0.0;
                        for (int j = 1; j <= nsub; ++j) {
                            const int k = index[j - 1];
                            temp1 += WY(k, pointr) * ds[j - 1];
                            temp2 += WS(k, pointr) * ds[j - 1];
                        }
                        wv[i - 1] = temp1;
                        wv[col + i - 1] = theta * temp2;
                    }
                    Wv::sync();
                    const int col2 = 2 * col;
                    info = lbp::dtrsl_upper<Wv>(W.wn, m2, col2, wv, 11, W.sacc);
                    if (info == 0) {
                        for (int i = Wv::lane(); i < col; i += Wv::W) wv[i] = -wv[i];
                        Wv::sync();
                        info = lbp::dtrsl_upper<Wv>(W.wn, m2, col2, wv, 1, W.sacc);
                    }
                    if (info == 0) {
                        for (int i = 1 + Wv::lane(); i <= nsub; i += Wv::W) {  // the Newton step, one component per lane
                            const int k = index[i - 1];
                            double dsi = ds[i - 1];
                            int pointr = head;
                            for (int jy = 1; jy <= col; ++jy) {
                                const int js = col + jy;
                                dsi = dsi + WY(k, pointr) * wv[jy - 1] / theta + WS(k, pointr) * wv[js - 1];
                                pointr = pointr % m + 1;
                            }
                            ds[i - 1] = dsi * (1.0 / theta);
                        }
                        Wv::sync();
                        int iword = 0;
                        for (int i = Wv::lane(); i < n; i += Wv::W) xp[i] = xs[i];
                        Wv::sync();
                        if (!cnstnd) {  // no bounds: the projection is the plain step, one component per lane
                            for (int i = 1 + Wv::lane(); i <= nsub; i += Wv::W) {
                                const int k = index[i - 1];
                                xs[k - 1] = xs[k - 1] + ds[i - 1];
                            }
                            Wv::sync();
                        } else
                        for (int i = 1; i <= nsub; ++i) {
                            const int k = index[i - 1];
                            const double dk = ds[i - 1];
                            double xk = xs[k - 1];
                            if (nbd[k - 1] != 0) {
                                if (nbd[k - 1] == 1) {
                                    xs[k - 1] = dmax(l[k - 1], xk + dk);
                                    if (xs[k - 1] == l[k - 1]) iword = 1;
                                } else if (nbd[k - 1] == 2) {
                                    xk = dmax(l[k - 1], xk + dk);
                                    xs[k - 1] = dmin(u[k - 1], xk);
                                    if (xs[k - 1] == l[k - 1] || xs[k - 1] == u[k - 1]) iword = 1;
                                } else if (nbd[k - 1] == 3) {
                                    xs[k - 1] = dmin(u[k - 1], xk + dk);
                                    if (xs[k - 1] == u[k - 1]) iword = 1;
                                }
                            } else {
                                xs[k - 1] = xk + dk;
                            }
                        }
                        if (iword != 0) {
                            double dd_p = 0.0;
                            for (int i = 0; i < n; ++i) dd_p += (xs[i] - x[i]) * g[i];
                            if (dd_p > 0.0) {
                                for (int i = 0; i < n; ++i) xs[i] = xp[i];
                                double alpha = 1.0, temp1 = alpha;
                                int ibd = 0;
                                for (int i = 1; i <= nsub; ++i) {
                                    const int k = index[i - 1];
                                    const double dk = ds[i - 1];
                                    if (nbd[k - 1] != 0) {
                                        if (dk < 0.0 && nbd[k - 1] <= 2) {
                                            const double temp2 = l[k - 1] - xs[k - 1];
                                            if (temp2 >= 0.0) temp1 = 0.0;
                                            else if (dk * alpha < temp2) temp1 = temp2 / dk;
                                        } else if (dk > 0.0 && nbd[k - 1] >= 2) {
                                            const double temp2 = u[k - 1] - xs[k - 1];
                                            if (temp2 <= 0.0) temp1 = 0.0;
                                            else if (dk * alpha > temp2) temp1 = temp2 / dk;
                                        }
                                        if (temp1 < alpha) { alpha = temp1; ibd = i; }
                                    }
                                }
                                if (alpha < 1.0) {
                                    const double dk = ds[ibd - 1];
                                    const int k = index[ibd - 1];
                                    if (dk > 0.0) { xs[k - 1] = u[k - 1]; ds[ibd - 1] = 0.0; }
                                    else if (dk < 0.0) { xs[k - 1] = l[k - 1]; ds[ibd - 1] = 0.0; }
                                }
                                for (int i = 1; i <= nsub; ++i) {
                                    const int k = index[i - 1];
                                    xs[k - 1] = xs[k - 1] + alpha * ds[i - 1];
                                }
                            }
                        }
                    }
                }
                if (info != 0) { refresh(); continue; }
            }
            // ---------------------------------------------------------------- 555: line search
            Wv::sync();
            for (int i = Wv::lane(); i < n; i += Wv::W) d[i] = z[i] - x[i];
            Wv::sync();
            dtd = 0.0;
            for (int i = 0; i < n; ++i) dtd += d[i] * d[i];
            const double dnorm = sqrt(dtd);
            double stpmx = 1e10;
            if (cnstnd) {
                if (iter == 0) {
                    stpmx = 1.0;
                } else {
                    for (int i = 0; i < n; ++i) {
                        const double a1 = d[i];
                        if (nbd[i] != 0) {
                            if (a1 < 0.0 && nbd[i] <= 2) {
                                const double a2 = l[i] - x[i];
                                if (a2 >= 0.0) stpmx = 0.0;
                                else if (a1 * stpmx < a2) stpmx = a2 / a1;
                            } else if (a1 > 0.0 && nbd[i] >= 2) {
                                const double a2 = u[i] - x[i];
                                if (a2 <= 0.0) stpmx = 0.0;
                                else if (a1 * stpmx > a2) stpmx = a2 / a1;
                            }
                        }
                    }
                }
            }
            if (iter == 0 && !boxed) stp = dmin(1.0 / dnorm, stpmx);
            else stp = 1.0;
            for (int i = Wv::lane(); i < n; i += Wv::W) { t[i] = x[i]; r[i] = g[i]; }
            Wv::sync();
            fold = f;
            int ifun = 0;
            bool lsfail = false;
            gd = 0.0;
            for (int i = 0; i < n; ++i) gd += g[i] * d[i];
            gdold = gd;
            Dcsrch ls;
            if (gd >= 0.0) {
                lsfail = true;
            } else {
                if (ls.start(f, gd, stp, stpmx) == Dcsrch::ERR) lsfail = true;
                while (!lsfail) {
                    ifun += 1;
                    if (ifun - 1 >= maxls) { lsfail = true; break; }
                    Wv::sync();
                    if (stp == 1.0) {
                        for (int i = Wv::lane(); i < n; i += Wv::W) x[i] = z[i];
                    } else {
                        for (int i = Wv::lane(); i < n; i += Wv::W) {
                            double xi = stp * d[i] + t[i];
                            if (nbd[i] == 1 || nbd[i] == 2) xi = dmax(xi, l[i]);
                            if (nbd[i] == 2 || nbd[i] == 3) xi = dmin(xi, u[i]);
                            x[i] = xi;
                        }
                    }
                    Wv::sync();
                    fg(x, f, g);
                    Wv::sync();
                    nfev += 1;
                    gd = 0.0;
                    for (int i = 0; i < n; ++i) gd += g[i] * d[i];
                    if (ls.step(f, gd, stp) != Dcsrch::FG) break;
                }
            }
            if (lsfail) {
                Wv::sync();
                for (int i = Wv::lane(); i < n; i += Wv::W) { x[i] = t[i]; g[i] = r[i]; }
                Wv::sync();
                f = fold;
                if (col == 0) {
                    R = {f, false, nfev, iter, 3};
                    return R;
                }
                refresh();
                continue;
            }
            done_dir = true;
        }
        // -------------------------------------------------------------------- NEW_X
        iter += 1;
        sbgnrm = projgr();
        // scipy driver: iteration / evaluation limits are checked when NEW_X is reported
        if (iter >= maxiter || nfev > maxfun) {
            R = {f, false, nfev, iter, 4};
            return R;
        }
        if (sbgnrm <= pgtol) {
            R = {f, true, nfev, iter, 1};
            return R;
        }
        {
            const double ddum = dmax(fabs(fold), dmax(fabs(f), 1.0));
            if ((fold - f) <= tol * ddum) {
                R = {f, true, nfev, iter, 2};
                return R;
            }
        }
        Wv::sync();
        for (int i = Wv::lane(); i < n; i += Wv::W) r[i] = g[i] - r[i];
        Wv::sync();
        double rr = 0.0;
        for (int i = 0; i < n; ++i) rr += r[i] * r[i];
        double dr, ddum;
        if (stp == 1.0) {
            dr = gd - gdold;
            ddum = -gdold;
        } else {
            dr = (gd - gdold) * stp;
            for (int i = Wv::lane(); i < n; i += Wv::W) d[i] *= stp;
            Wv::sync();
            ddum = -gdold * stp;
        }
        if (dr <= epsmch * ddum) {
            updatd = false;
            continue;
        }
        updatd = true;
        iupdat += 1;
        // -------------------------------------------------------------------- matupd
        if (iupdat <= m) {
            col = iupdat;
            itail = (head + iupdat - 2) % m + 1;
        } else {
            itail = itail % m + 1;
            head = head % m + 1;
        }
        Wv::sync();
        for (int i = 1 + Wv::lane(); i <= n; i += Wv::W) { WS(i, itail) = d[i - 1]; WY(i, itail) = r[i - 1]; }
        theta = rr / dr;
        if (iupdat > m) {
            // SS moves one row up and one column left, SY likewise: simultaneous moves (read, barrier, write)
            constexpr int NSH = ((M - 1) * M + Wv::W - 1) / Wv::W;
            double tmp[NSH];
            auto walk = [&](auto&& fn) {
                int e = 0, slot = 0;
                for (int j = 1; j <= col - 1; ++j) {
                    for (int q = 0; q < j; ++q, ++e)
                        if (e % Wv::W == Wv::lane()) fn(slot++, 0, 1 + q, j, 2 + q, j + 1);
                    for (int q = 0; q < col - j; ++q, ++e)
                        if (e % Wv::W == Wv::lane()) fn(slot++, 1, j + q, j, j + 1 + q, j + 1);
                }
            };
            if constexpr (Wv::W == 1) {
                (void)tmp;
                walk([&](int, int which, int dr_, int dc_, int sr_, int sc_) {
                    if (which == 0) SS(dr_, dc_) = SS(sr_, sc_);
                    else SY(dr_, dc_) = SY(sr_, sc_);
                });
            } else {
#pragma unroll
                for (int i = 0; i < NSH; ++i) tmp[i] = 0.0;
                walk([&](int slot, int which, int, int, int sr_, int sc_) {
                    const double v = which == 0 ? SS(sr_, sc_) : SY(sr_, sc_);
#pragma unroll
                    for (int i = 0; i < NSH; ++i)
                        if (i == slot) tmp[i] = v;
                });
                Wv::sync();
                walk([&](int slot, int which, int dr_, int dc_, int, int) {
                    double v = 0.0;
#pragma unroll
                    for (int i = 0; i < NSH; ++i)
                        if (i == slot) v = tmp[i];
                    if (which == 0) SS(dr_, dc_) = v;
                    else SY(dr_, dc_) = v;
                });
            }
        }
        Wv::sync();
        {
            for (int j = 1 + Wv::lane(); j <= col - 1; j += Wv::W) {  // last row of SY, last column of SS
                const int pointr = (head + j - 2) % m + 1;
                double s1 = 0.0, s2 = 0.0;
                for (int i = 1; i <= n; ++i) { s1 += d[i - 1] * WY(i, pointr); s2 += WS(i, pointr) * d[i - 1]; }
                SY(col, j) = s1;
                SS(j, col) = s2;
            }
            if (Wv::lane() == 0) {
                if (stp == 1.0) SS(col, col) = dtd;
                else SS(col, col) = stp * stp * dtd;
                SY(col, col) = dr;
            }
        }
        Wv::sync();
        // -------------------------------------------------------------------- formt
        for (int e = Wv::lane(); e < col * col; e += Wv::W) {  // one entry of the upper triangle per lane
            const int i = e / col + 1, j = e % col + 1;
            if (j < i) continue;
            if (i == 1) {
                WT(1, j) = theta * SS(1, j);
            } else {
                const int k1 = i - 1;  // min(i, j) - 1
                double ddum2 = 0.0;
                for (int k = 1; k <= k1; ++k) ddum2 += SY(i, k) * SY(j, k) / SY(k, k);
                WT(i, j) = ddum2 + theta * SS(i, j);
            }
        }
        Wv::sync();
        if (lbp::dpofa<Wv>(W.wt, m, col, W.sacc) != 0) refresh();
    }
#undef WS
#undef WY
#undef SY
#undef SS
#undef WT
#undef WN
#undef WN1
}

}  // namespace dsq
