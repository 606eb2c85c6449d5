// dsq_lbfgsb_dense.h — L-BFGS-B for a HANDFUL of variables with the quasi-Newton matrix held densely.
//
// Same algorithm and the same iterates (up to rounding) as dsq_lbfgsb.h / scipy's L-BFGS-B — limited
// memory BFGS matrix B_k built from the last m = 10 pairs on B_0 = theta*I, generalized Cauchy point,
// subspace minimisation with projection, MINPACK-2 line search, scipy's stopping rules — but for
// n <= NMAX <= 4 the matrix B_k (n x n) is formed explicitly by replaying the stored BFGS updates
// (Byrd-Nocedal-Schnabel: the compact representation theta*I - W M W^T IS that matrix), which turns
// the O(m^3) compact-form algebra on 2m x 2m matrices into a few dozen flops held in registers.
// Used for the 2-coefficient dispersion-trend fit, where the scalar optimiser logic — not the data
// passes — dominated the launch.  (The compact-form bookkeeping quirks of the reference code cannot
// trigger here: they need a variable that sits on a bound while correction pairs exist.)
#pragma once
#include "dsq_lbfgsb.h"

namespace dsq {

#if defined(DSQ_TREND_PHASES) && defined(__HIPCC__)
__device__ long long g_lbd_phase[8];
#endif
#if defined(DSQ_TREND_PHASES) && defined(__HIP_DEVICE_COMPILE__)
#define LBD_PH(k)                                                        \
    {                                                                    \
        const long long now_ = clock64();                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_lbd_phase[k] += now_ - lbd_t_; \
        lbd_t_ = now_;                                                   \
    }
#else
#define LBD_PH(k)
#endif

template <int NMAX, int M = 10>
struct LbfgsbDenseWork {
    double S[M][NMAX], Y[M][NMAX];  // circular pair storage
    double RHO[M];                  // 1 / y^T s of a pair (the inverse update of the interior iterations)
};

// FG: void(const double* x, double& f, double* g)
template <int NMAX, class FG, int M = 10>
DSQ_HD LbfgsbResult lbfgsb_dense(FG&& fg, int n, double* x, const double* l, const double* u,
                                 const int* nbd, LbfgsbDenseWork<NMAX, M>& W, double factr = 1e7,
                                 double pgtol = 1e-5, int maxls = 20, int maxiter = 15000,
                                 int maxfun = 15000) {
    const double epsmch = kEps, tol = factr * epsmch;
    double g[NMAX], z[NMAX], d[NMAX], t[NMAX], r[NMAX], B[NMAX][NMAX];
    int iwhere[NMAX];
    int col = 0, head = 0, iter = 0, nfev = 0;
    double theta = 1.0, f = 0.0, fold = 0.0, gd = 0.0, gdold = 0.0, stp = 0.0, dtd = 0.0;
    LbfgsbResult R;

    bool cnstnd = false, boxed = true;
    for (int i = 0; i < n; ++i) {
        if (nbd[i] > 0) {
            if (nbd[i] <= 2 && x[i] <= l[i]) x[i] = l[i];
            else if (nbd[i] >= 2 && x[i] >= u[i]) x[i] = u[i];
        }
        if (nbd[i] != 2) boxed = false;
        if (nbd[i] != 0) cnstnd = true;
    }
    auto projgr = [&]() {
        double s = 0.0;
        for (int i = 0; i < n; ++i) {
            double gi = g[i];
            if (nbd[i] != 0) {
                if (gi < 0.0) { if (nbd[i] >= 2) gi = dmax(x[i] - u[i], gi); }
                else { if (nbd[i] <= 2) gi = dmin(x[i] - l[i], gi); }
            }
            s = dmax(s, fabs(gi));
        }
        return s;
    };
    auto build_B = [&]() {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) B[i][j] = (i == j) ? theta : 0.0;
        for (int q = 0; q < col; ++q) {
            const int p = (head + q) % M;
            double Bs[NMAX], sBs = 0.0, ys = 0.0;
            for (int i = 0; i < n; ++i) {
                double v = 0.0;
                for (int j = 0; j < n; ++j) v += B[i][j] * W.S[p][j];
                Bs[i] = v;
            }
            for (int i = 0; i < n; ++i) { sBs += W.S[p][i] * Bs[i]; ys += W.Y[p][i] * W.S[p][i]; }
            // (two reciprocals per replayed pair, not 2 n^2 divisions: the replay is the optimiser's critical path
            // between two evaluations - up to ten pairs per iteration; fdiv: <= 1 ulp, a fifth of an IEEE division)
            const double rys = fdiv(1.0, ys), rsBs = fdiv(1.0, sBs);
            for (int i = 0; i < n; ++i) {
                const double yi = W.Y[p][i] * rys, bi = Bs[i] * rsBs;
                for (int j = 0; j < n; ++j) B[i][j] += yi * W.Y[p][j] - bi * Bs[j];
            }
        }
    };

    fg(x, f, g);
    nfev = 1;
    double sbgnrm = projgr();
    if (sbgnrm <= pgtol) { R = {f, true, nfev, 0, 0}; return R; }

#if defined(DSQ_TREND_PHASES) && defined(__HIP_DEVICE_COMPILE__)
    long long lbd_t_ = clock64();
#endif
    bool unbounded = true;
    for (int i = 0; i < n; ++i)
        if (nbd[i] != 0) unbounded = false;
    for (;;) {
        LBD_PH(5)
        // ------------------------------------------------------------ interior iteration
        // When no bound takes part - the problem has none, or (two variables with lower bounds: the trend's case) no
        // variable sits on its bound against the gradient, the minimiser of the model along -g comes before the first
        // breakpoint, and the subspace minimiser lies strictly inside - the generalized Cauchy point fixes nothing and the
        // subspace minimisation over all variables returns x - B^-1 g whatever the Cauchy point was.  H = B^-1 is the
        // same pairs replayed through the inverse BFGS update on I / theta: no division per pair, no factorisation, no
        // elimination - a third of the dependent operations of the code below, which is the latency of the trend kernel
        // between two data passes.  Same iterate up to rounding; any doubt sends the iteration through the general code.
        bool interior = false;
        if (unbounded || (NMAX == 2 && n == 2 && nbd[0] == 1 && nbd[1] == 1)) {
            double H[NMAX][NMAX];
            const double ith = fdiv(1.0, theta);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) H[i][j] = (i == j) ? ith : 0.0;
            for (int q = 0; q < col; ++q) {
                const int p = (head + q) % M;
                const double rho = W.RHO[p];
                double Hy[NMAX], yHy = 0.0;
                for (int i = 0; i < n; ++i) {
                    double v = 0.0;
                    for (int j = 0; j < n; ++j) v += H[i][j] * W.Y[p][j];
                    Hy[i] = v;
                }
                for (int i = 0; i < n; ++i) yHy += W.Y[p][i] * Hy[i];
                const double c = rho * (rho * yHy + 1.0);
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j)
                        H[i][j] += c * (W.S[p][i] * W.S[p][j]) - rho * (W.S[p][i] * Hy[j] + Hy[i] * W.S[p][j]);
            }
            bool ok = true;
            for (int i = 0; i < n; ++i) {
                double v = 0.0;
                for (int j = 0; j < n; ++j) v += H[i][j] * g[j];
                z[i] = x[i] - v;
            }
            if (!unbounded) {  // n == 2, lower bounds
                double t_min = INFINITY;
                for (int i = 0; i < 2; ++i) {
                    const double tl = x[i] - l[i];
                    if (tl <= 0.0 && g[i] >= 0.0) ok = false;            // on its bound, pushed outwards: stays fixed
                    if (g[i] > 0.0) t_min = dmin(t_min, fdiv(tl, g[i]));  // breakpoint of the projected path
                    if (!(z[i] > l[i])) ok = false;                       // the projection would act
                }
                // minimiser of the model along -g: g'g / g'Bg with B = H^-1 (2 x 2: adjugate / determinant)
                const double det = H[0][0] * H[1][1] - H[0][1] * H[1][0];
                const double gBg = fdiv(H[1][1] * g[0] * g[0] - (H[0][1] + H[1][0]) * g[0] * g[1] + H[0][0] * g[1] * g[1], det);
                const double dtm = fdiv(g[0] * g[0] + g[1] * g[1], gBg);
                if (!(det > 0.0) || !(gBg > 0.0) || !(dtm < t_min)) ok = false;
            }
            for (int i = 0; i < n; ++i)
                if (!(z[i] == z[i])) ok = false;
            interior = ok;
        }
        if (!interior) {
        build_B();
        LBD_PH(0)
        // ------------------------------------------------------------ generalized Cauchy point
        for (int i = 0; i < n; ++i) z[i] = x[i];
        double tb[NMAX];
        bool bnded = true;
        int nbreak = 0, nfreec = 0;
        for (int i = 0; i < n; ++i) {
            const double neggi = -g[i];
            double tl = 0.0, tu = 0.0;
            if (nbd[i] != 0 && nbd[i] <= 2) tl = x[i] - l[i];
            if (nbd[i] >= 2) tu = u[i] - x[i];
            const bool xlower = nbd[i] != 0 && nbd[i] <= 2 && tl <= 0.0;
            const bool xupper = nbd[i] >= 2 && tu <= 0.0;
            iwhere[i] = 0;
            if (nbd[i] == 0) iwhere[i] = -1;
            else if (xlower) { if (neggi <= 0.0) iwhere[i] = 1; }
            else if (xupper) { if (neggi >= 0.0) iwhere[i] = 2; }
            else if (fabs(neggi) <= 0.0) iwhere[i] = -3;
            tb[i] = -1.0;  // no breakpoint
            if (iwhere[i] != 0 && iwhere[i] != -1) {
                d[i] = 0.0;
            } else {
                d[i] = neggi;
                if (nbd[i] != 0 && nbd[i] <= 2 && neggi < 0.0) { tb[i] = fdiv(tl, -neggi); nbreak += 1; }
                else if (nbd[i] >= 2 && neggi > 0.0) { tb[i] = fdiv(tu, neggi); nbreak += 1; }
                else { nfreec += 1; if (fabs(neggi) > 0.0) bnded = false; }
            }
        }
        if (sbgnrm > 0.0 && (nbreak > 0 || nfreec > 0)) {
            double f1 = 0.0;
            for (int i = 0; i < n; ++i) f1 -= d[i] * d[i];
            auto quad = [&](const double* a, const double* b) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) {
                    double v = 0.0;
                    for (int j = 0; j < n; ++j) v += B[i][j] * b[j];
                    s += a[i] * v;
                }
                return s;
            };
            double f2 = quad(d, d);
            const double f2_org = f2;
            double dtm = fdiv(-f1, f2), tsum = 0.0, tj = 0.0;
            bool used[NMAX];
            for (int i = 0; i < n; ++i) used[i] = false;
            int nleft = nbreak;
            bool all_fixed = false;
            while (nleft > 0) {
                int ibp = -1;
                for (int i = 0; i < n; ++i)
                    if (tb[i] >= 0.0 && !used[i] && (ibp < 0 || tb[i] < tb[ibp])) ibp = i;
                const double tj0 = tj;
                tj = tb[ibp];
                const double dt = tj - tj0;
                if (dtm < dt) break;
                tsum += dt;
                nleft -= 1;
                used[ibp] = true;
                const double dibp = d[ibp];
                d[ibp] = 0.0;
                if (dibp > 0.0) { z[ibp] = u[ibp]; iwhere[ibp] = 2; }
                else { z[ibp] = l[ibp]; iwhere[ibp] = 1; }
                if (nleft == 0 && nbreak == n) { dtm = dt; all_fixed = true; break; }
                // derivatives of the model along the new segment: zc = current point - x
                double zc[NMAX];
                for (int i = 0; i < n; ++i) zc[i] = used[i] ? (z[i] - x[i]) : tsum * d[i];
                f1 = 0.0;
                for (int i = 0; i < n; ++i) f1 += g[i] * d[i];
                f1 += quad(d, zc);
                f2 = quad(d, d);
                f2 = dmax(epsmch * f2_org, f2);
                if (nleft > 0) { dtm = fdiv(-f1, f2); }
                else if (bnded) { f1 = 0.0; f2 = 0.0; dtm = 0.0; }
                else { dtm = fdiv(-f1, f2); }
            }
            if (!all_fixed) {
                if (dtm <= 0.0) dtm = 0.0;
                tsum += dtm;
                for (int i = 0; i < n; ++i)
                    if (!used[i]) z[i] = x[i] + tsum * d[i];
            }
        }
        LBD_PH(1)
        // ------------------------------------------------------------ subspace minimisation
        int nfree = 0, idx[NMAX];
        for (int i = 0; i < n; ++i)
            if (iwhere[i] <= 0) idx[nfree++] = i;
        if (nfree > 0 && col > 0) {
            double rr_[NMAX], A[NMAX][NMAX], ds[NMAX], xp[NMAX];
            for (int a = 0; a < nfree; ++a) {
                const int i = idx[a];
                double v = g[i];
                for (int j = 0; j < n; ++j) v += B[i][j] * (z[j] - x[j]);
                rr_[a] = -v;
                for (int b = 0; b < nfree; ++b) A[a][b] = B[i][idx[b]];
            }
            // Gaussian elimination (SPD, nfree <= NMAX)
            for (int a = 0; a < nfree; ++a) {
                const double piv = A[a][a];
                for (int b = a + 1; b < nfree; ++b) {
                    const double m_ = fdiv(A[b][a], piv);
                    for (int c = a; c < nfree; ++c) A[b][c] -= m_ * A[a][c];
                    rr_[b] -= m_ * rr_[a];
                }
            }
            for (int a = nfree - 1; a >= 0; --a) {
                double v = rr_[a];
                for (int b = a + 1; b < nfree; ++b) v -= A[a][b] * ds[b];
                ds[a] = fdiv(v, A[a][a]);
            }
            int iword = 0;
            for (int i = 0; i < n; ++i) xp[i] = z[i];
            for (int a = 0; a < nfree; ++a) {
                const int k = idx[a];
                const double dk = ds[a];
                double xk = z[k];
                if (nbd[k] != 0) {
                    if (nbd[k] == 1) { z[k] = dmax(l[k], xk + dk); if (z[k] == l[k]) iword = 1; }
                    else if (nbd[k] == 2) { xk = dmax(l[k], xk + dk); z[k] = dmin(u[k], xk); if (z[k] == l[k] || z[k] == u[k]) iword = 1; }
                    else { z[k] = dmin(u[k], xk + dk); if (z[k] == u[k]) iword = 1; }
                } else {
                    z[k] = xk + dk;
                }
            }
            if (iword != 0) {
                double dd_p = 0.0;
                for (int i = 0; i < n; ++i) dd_p += (z[i] - x[i]) * g[i];
                if (dd_p > 0.0) {
                    for (int i = 0; i < n; ++i) z[i] = xp[i];
                    double alpha = 1.0, temp1 = alpha;
                    int ibd = -1;
                    for (int a = 0; a < nfree; ++a) {
                        const int k = idx[a];
                        const double dk = ds[a];
                        if (nbd[k] != 0) {
                            if (dk < 0.0 && nbd[k] <= 2) {
                                const double t2 = l[k] - z[k];
                                if (t2 >= 0.0) temp1 = 0.0; else if (dk * alpha < t2) temp1 = fdiv(t2, dk);
                            } else if (dk > 0.0 && nbd[k] >= 2) {
                                const double t2 = u[k] - z[k];
                                if (t2 <= 0.0) temp1 = 0.0; else if (dk * alpha > t2) temp1 = fdiv(t2, dk);
                            }
                            if (temp1 < alpha) { alpha = temp1; ibd = a; }
                        }
                    }
                    if (alpha < 1.0 && ibd >= 0) {
                        const double dk = ds[ibd];
                        const int k = idx[ibd];
                        if (dk > 0.0) { z[k] = u[k]; ds[ibd] = 0.0; }
                        else if (dk < 0.0) { z[k] = l[k]; ds[ibd] = 0.0; }
                    }
                    for (int a = 0; a < nfree; ++a) z[idx[a]] += alpha * ds[a];
                }
            }
        }
        }  // (!interior)
        LBD_PH(2)
        // ------------------------------------------------------------ line search
        for (int i = 0; i < n; ++i) d[i] = z[i] - x[i];
        dtd = 0.0;
        for (int i = 0; i < n; ++i) dtd += d[i] * d[i];
        double stpmx = 1e10;
        if (cnstnd) {
            if (iter == 0) stpmx = 1.0;
            else
                for (int i = 0; i < n; ++i) {
                    const double a1 = d[i];
                    if (nbd[i] != 0) {
                        if (a1 < 0.0 && nbd[i] <= 2) {
                            const double a2 = l[i] - x[i];
                            if (a2 >= 0.0) stpmx = 0.0; else if (a1 * stpmx < a2) stpmx = fdiv(a2, a1);
                        } else if (a1 > 0.0 && nbd[i] >= 2) {
                            const double a2 = u[i] - x[i];
                            if (a2 <= 0.0) stpmx = 0.0; else if (a1 * stpmx > a2) stpmx = fdiv(a2, a1);
                        }
                    }
                }
        }
        stp = (iter == 0 && !boxed) ? dmin(1.0 / sqrt(dtd), stpmx) : 1.0;
        for (int i = 0; i < n; ++i) { t[i] = x[i]; r[i] = g[i]; }
        fold = f;
        int ifun = 0;
        bool lsfail = false;
        gd = 0.0;
        for (int i = 0; i < n; ++i) gd += g[i] * d[i];
        gdold = gd;
        Dcsrch ls;
        if (gd >= 0.0) lsfail = true;
        else {
            if (ls.start(f, gd, stp, stpmx) == Dcsrch::ERR) lsfail = true;
            while (!lsfail) {
                ifun += 1;
                if (ifun - 1 >= maxls) { lsfail = true; break; }
                if (stp == 1.0) { for (int i = 0; i < n; ++i) x[i] = z[i]; }
                else
                    for (int i = 0; i < n; ++i) {
                        x[i] = stp * d[i] + t[i];
                        if (nbd[i] == 1 || nbd[i] == 2) x[i] = dmax(x[i], l[i]);
                        if (nbd[i] == 2 || nbd[i] == 3) x[i] = dmin(x[i], u[i]);
                    }
                LBD_PH(3)
                fg(x, f, g);
                LBD_PH(4)
                nfev += 1;
                gd = 0.0;
                for (int i = 0; i < n; ++i) gd += g[i] * d[i];
                if (ls.step(f, gd, stp) != Dcsrch::FG) break;
            }
        }
        if (lsfail) {
            for (int i = 0; i < n; ++i) { x[i] = t[i]; g[i] = r[i]; }
            f = fold;
            if (col == 0) { R = {f, false, nfev, iter, 3}; return R; }
            col = 0; head = 0; theta = 1.0;
            continue;
        }
        iter += 1;
        sbgnrm = projgr();
        if (iter >= maxiter || nfev > maxfun) { R = {f, false, nfev, iter, 4}; return R; }
        if (sbgnrm <= pgtol) { R = {f, true, nfev, iter, 1}; return R; }
        {
            const double ddum = dmax(fabs(fold), dmax(fabs(f), 1.0));
            if ((fold - f) <= tol * ddum) { R = {f, true, nfev, iter, 2}; return R; }
        }
        double rr = 0.0;
        for (int i = 0; i < n; ++i) { r[i] = g[i] - r[i]; rr += r[i] * r[i]; }
        double dr, ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else { dr = (gd - gdold) * stp; for (int i = 0; i < n; ++i) d[i] *= stp; ddum = -gdold * stp; }
        if (dr <= epsmch * ddum) continue;  // skip the update
        int slot;
        if (col < M) { slot = (head + col) % M; col += 1; }
        else { slot = head; head = (head + 1) % M; }
        double ys = 0.0;
        for (int i = 0; i < n; ++i) { W.S[slot][i] = d[i]; W.Y[slot][i] = r[i]; ys += r[i] * d[i]; }
        W.RHO[slot] = fdiv(1.0, ys);
        theta = fdiv(rr, dr);
    }
}

}  // namespace dsq
