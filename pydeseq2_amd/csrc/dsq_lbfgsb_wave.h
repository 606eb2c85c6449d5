// dsq_lbfgsb_wave.h — UNCONSTRAINED L-BFGS-B for 5 ... 12 variables with the quasi-Newton matrix in the registers of a
// wavefront: up to 8 variables lane (i, j) = (lane >> 3, lane & 7) owns entry (i, j) of an 8 x 8 inverse matrix H_k; from 9
// to 16 lane (i, jg) = (lane >> 2, lane & 3) owns the four entries (i, 4 jg ... 4 jg + 3) of a 16 x 16 one; from 17 to 32
// lane (i, jg) = (lane >> 1, lane & 1) owns sixteen entries of a row of a 32 x 32 one.
//
// Why.  The apeGLM objective (utils.py:990-1207, minimize(..., method="L-BFGS-B") without bounds) is cheap - one pass
// over the samples per evaluation - and scipy's optimiser between two evaluations is not: in its compact representation
// (dsq_lbfgsb.h) an iteration is ~3500 dependent operations on 2m x 2m matrices; even spread over the lanes output by
// output (dsq_lbfgsb_par.h) k_shrink<8> spent 26 ms on 60 000 genes, 30 us per iteration, of which the evaluations are a
// tenth.  Without bounds the algorithm collapses: no variable ever sits on a bound, the generalized Cauchy point has no
// breakpoints, and the subspace minimisation over ALL variables returns  x - B_k^-1 g  whatever the Cauchy point was.
// B_k is the limited-memory BFGS matrix of the last m = 10 pairs on B_0 = theta I (Byrd, Nocedal, Schnabel: the compact
// representation IS that matrix), so H_k = B_k^-1 is the same pairs replayed through the inverse update
//     H <- H - rho (s (Hy)^T + (Hy) s^T) + rho (rho y^T H y + 1) s s^T,      rho = 1 / y^T s,   H_0 = I / theta,
// which needs no factorisation and no solve: per pair two 8-lane sums for H y (by rows, and - H is symmetric - by
// columns), one for y^T H y, and one fused update of the lane's entry; the direction is one more 8-lane sum.  The 8-lane
// sums are three DPP / lane-swap steps each.  Everything else - line search (MINPACK-2 dcsrch, as scipy), the stopping
// rules, theta, the curvature test that skips an update, the refresh after a failed search - is dsq_lbfgsb_dense.h's code
// for the case nbd = 0, on 8-vectors in the wave's LDS.
// Same algorithm, same iterates up to rounding as scipy (like dsq_lbfgsb_dense.h, which the p <= 4 designs use):
// tests/test_gpu_summary.py holds it to the unmodified utils.nbinomGLM's outputs (convergence flags included).
#pragma once
#include "dsq_lbfgsb.h"
#include "dsq_wave.h"

namespace dsq {

template <int R>
struct LbfgsbWaveWorkT {  // wave-private LDS (R = 8: 1.7 KB; the compact form's workspace is 11.6 KB at p = 8)
    double S[10][R], Y[10][R], RHO[10];
    double x[R], g[R], t[R], r[R], z[R], d[R];
};
typedef LbfgsbWaveWorkT<8> LbfgsbWaveWork;

#if defined(__HIP_DEVICE_COMPILE__)
namespace wv8 {
// sum over the 8 lanes that share lane >> 3 (a row of the matrix): xor 1, xor 2, then the other quad of the 8
__device__ __forceinline__ double rowsum(double v) {
    v += detail::dpp_d<detail::kXor1>(v);
    v += detail::dpp_d<detail::kXor2>(v);
    v += detail::dpp_d<0x141>(v);  // row_half_mirror: lane k of every 8 <- lane 7 - k (its quad's sum is complete)
    return v;
}
// sum over the 8 lanes that share lane & 7 (a column): xor 8, xor 16, xor 32
__device__ __forceinline__ double colsum(double v) {
    double a, c;
    v += detail::dpp_d<detail::kRor8>(v);
    detail::swap_d<false>(v, a, c); v = a + c;
    detail::swap_d<true>(v, a, c); v = a + c;
    return v;
}
}  // namespace wv8
namespace wv16 {
// lane = 4 i + jg: the four lanes of a quad hold row i
__device__ __forceinline__ double rowsum(double v) {  // over the quad
    v += detail::dpp_d<detail::kXor1>(v);
    v += detail::dpp_d<detail::kXor2>(v);
    return v;
}
// over the 16 lanes that share lane & 3 (a column group): + 8, + 4 inside a 16-lane row, then xor 16, xor 32
__device__ __forceinline__ double colsum(double v) {
    double a, c;
    v += detail::dpp_d<detail::kRor8>(v);
    v += detail::dpp_d<detail::kRor4>(v);
    detail::swap_d<false>(v, a, c); v = a + c;
    detail::swap_d<true>(v, a, c); v = a + c;
    return v;
}
}  // namespace wv16
namespace wv32 {
// lane = 2 i + jg: two lanes hold row i
__device__ __forceinline__ double rowsum(double v) { return v + detail::dpp_d<detail::kXor1>(v); }
// over the 32 lanes that share lane & 1: + 8, + 4 inside a 16-lane row (in that order: the second relies on the period
// the first leaves), xor 2, then xor 16, xor 32
__device__ __forceinline__ double colsum(double v) {
    double a, c;
    v += detail::dpp_d<detail::kRor8>(v);
    v += detail::dpp_d<detail::kRor4>(v);
    v += detail::dpp_d<detail::kXor2>(v);
    detail::swap_d<false>(v, a, c); v = a + c;
    detail::swap_d<true>(v, a, c); v = a + c;
    return v;
}
}  // namespace wv32
#endif

// x0 in W.x[0 .. P-1]; result in W.x.  FG: void(const double* x, double& f, double* g) (all lanes call it; g[0 .. P-1])
template <int P, int R, class FG>
DSQ_HD LbfgsbResult lbfgsb_wave(FG&& fg, LbfgsbWaveWorkT<R>& W, double factr = 1e7, double pgtol = 1e-5, int maxls = 20,
                                int maxiter = 15000, int maxfun = 15000) {
    static_assert((R == 8 || R == 16 || R == 32) && P >= 1 && P <= R,
                  "an 8 x 8 matrix with one entry per lane, 16 x 16 with four, 32 x 32 with sixteen");
    LbfgsbResult R_;
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int M = 10;
    constexpr int E = R * R / 64;  // entries per lane: 1, 4, 16
    const int lane = threadIdx.x & 63;
    const int i = R == 8 ? lane >> 3 : (R == 16 ? lane >> 2 : lane >> 1);                   // row of this lane's entries
    const int j0 = R == 8 ? (lane & 7) : (R == 16 ? 4 * (lane & 3) : 16 * (lane & 1));      // their first column
    const double epsmch = kEps, tol = factr * epsmch;
    int col = 0, head = 0, iter = 0, nfev = 0;
    double theta = 1.0, f = 0.0, fold = 0.0, gd = 0.0, gdold = 0.0, stp = 0.0;
    if (lane < R) {
        if (lane >= P) W.x[lane] = 0.0;
        W.g[lane] = 0.0; W.t[lane] = 0.0; W.r[lane] = 0.0; W.z[lane] = 0.0; W.d[lane] = 0.0;
    }
    DeviceWave::sync();
    auto projgr = [&]() {  // no bounds: the projected gradient is the gradient
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) s = dmax(s, fabs(W.g[k]));
        return s;
    };
    auto dot = [&](const double* a, const double* b) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) s += a[k] * b[k];
        return s;
    };
    auto rowsum = [&](double v) {
        if constexpr (R == 8) return wv8::rowsum(v);
        else if constexpr (R == 16) return wv16::rowsum(v);
        else return wv32::rowsum(v);
    };
    auto colsum = [&](double v) {
        if constexpr (R == 8) return wv8::colsum(v);
        else if constexpr (R == 16) return wv16::colsum(v);
        else return wv32::colsum(v);
    };
    fg(W.x, f, W.g);
    DeviceWave::sync();
    nfev = 1;
    double sbgnrm = projgr();
    if (sbgnrm <= pgtol) { R_ = {f, true, nfev, 0, 0}; return R_; }

    for (;;) {
        // ------------------------------------------------------------ H_k: the pairs replayed on I / theta
        double h[E];
        {
            const double ith = fdiv(1.0, theta);
#pragma unroll
            for (int c = 0; c < E; ++c) h[c] = (i == j0 + c) ? ith : 0.0;
        }
        for (int q = 0; q < col; ++q) {
            const int p = (head + q) % M;
            const double si = W.S[p][i], yi = W.Y[p][i], rho = W.RHO[p];
            double sj[E], yj[E], hy_j[E], t = 0.0;
#pragma unroll
            for (int c = 0; c < E; ++c) { sj[c] = W.S[p][j0 + c]; yj[c] = W.Y[p][j0 + c]; t += h[c] * yj[c]; }
            const double hy_i = rowsum(t);  // (H y)_i
#pragma unroll
            for (int c = 0; c < E; ++c) hy_j[c] = colsum(h[c] * yi);  // (H y)_j: H is symmetric
            const double yhy = colsum(yi * hy_i);
            const double cc = rho * yhy + 1.0;
#pragma unroll
            for (int c = 0; c < E; ++c) h[c] += rho * (cc * (si * sj[c]) - (si * hy_j[c] + hy_i * sj[c]));
        }
        // ------------------------------------------------------------ z = x - H g (Cauchy point + subspace minimisation)
        {
            double t = 0.0;
#pragma unroll
            for (int c = 0; c < E; ++c) t += h[c] * W.g[j0 + c];
            const double hg_i = rowsum(t);
            if (j0 == 0) { W.d[i] = -hg_i; W.z[i] = W.x[i] - hg_i; }
        }
        DeviceWave::sync();
        // ------------------------------------------------------------ line search (dsq_lbfgsb_dense.h, nbd = 0)
        const double dtd = dot(W.d, W.d);
        const double stpmx = 1e10;
        stp = (iter == 0) ? dmin(1.0 / sqrt(dtd), stpmx) : 1.0;
        if (lane < R) { W.t[lane] = W.x[lane]; W.r[lane] = W.g[lane]; }
        DeviceWave::sync();
        fold = f;
        int ifun = 0;
        bool lsfail = false;
        gd = dot(W.g, W.d);
        gdold = gd;
        Dcsrch ls;
        if (gd >= 0.0) lsfail = true;
        else {
            if (ls.start(f, gd, stp, stpmx) == Dcsrch::ERR) lsfail = true;
            while (!lsfail) {
                ifun += 1;
                if (ifun - 1 >= maxls) { lsfail = true; break; }
                if (lane < R) W.x[lane] = (stp == 1.0) ? W.z[lane] : stp * W.d[lane] + W.t[lane];
                DeviceWave::sync();
                fg(W.x, f, W.g);
                DeviceWave::sync();
                nfev += 1;
                gd = dot(W.g, W.d);
                if (ls.step(f, gd, stp) != Dcsrch::FG) break;
            }
        }
        if (lsfail) {
            if (lane < R) { W.x[lane] = W.t[lane]; W.g[lane] = W.r[lane]; }
            DeviceWave::sync();
            f = fold;
            if (col == 0) { R_ = {f, false, nfev, iter, 3}; return R_; }
            col = 0; head = 0; theta = 1.0;
            continue;
        }
        iter += 1;
        sbgnrm = projgr();
        if (iter >= maxiter || nfev > maxfun) { R_ = {f, false, nfev, iter, 4}; return R_; }
        if (sbgnrm <= pgtol) { R_ = {f, true, nfev, iter, 1}; return R_; }
        {
            const double ddum = dmax(fabs(fold), dmax(fabs(f), 1.0));
            if ((fold - f) <= tol * ddum) { R_ = {f, true, nfev, iter, 2}; return R_; }
        }
        if (lane < R) W.r[lane] = W.g[lane] - W.r[lane];
        DeviceWave::sync();
        const double rr = dot(W.r, W.r);
        double dr, ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else { dr = (gd - gdold) * stp; ddum = -gdold * stp; }
        if (dr <= epsmch * ddum) continue;  // skip the update
        int slot;
        if (col < M) { slot = (head + col) % M; col += 1; }
        else { slot = head; head = (head + 1) % M; }
        if (lane < R) {
            W.S[slot][lane] = (stp == 1.0) ? W.d[lane] : W.d[lane] * stp;
            W.Y[slot][lane] = W.r[lane];
        }
        if (lane == 0) W.RHO[slot] = fdiv(1.0, dr);
        theta = fdiv(rr, dr);
        DeviceWave::sync();
    }
#else
    (void)fg; (void)W; (void)factr; (void)pgtol; (void)maxls; (void)maxiter; (void)maxfun;
    R_ = {0.0, false, 0, 0, 3};
    return R_;
#endif
}

template <int P, class FG>
DSQ_HD LbfgsbResult lbfgsb_wave8(FG&& fg, LbfgsbWaveWork& W, double factr = 1e7, double pgtol = 1e-5) {
    return lbfgsb_wave<P, 8>(fg, W, factr, pgtol);
}

}  // namespace dsq
