// dsq_stats.h — the cheaper per-gene stages around the two optimisers.
//
//   gene_logmean      preprocessing.py:31-56   (deseq2_norm_fit)
//   mom_gene          utils.py:814-885 + dds.py:1149-1162 (rough / moments dispersions)
//   lin_mu_gene       utils.py:682-715         (fit_lin_mu)
//   wald_gene         utils.py:718-811         (wald_test)
//   cooks_gene        utils.py:567-679, 914-960 + dds.py:986-1040, 1066-1110
//   trimmed_base_mean utils.py:567-599 as used by dds.py:1332-1352 (outlier replacement)
// All follow the one-gene-per-wave layout of dsq_wave.h.
#pragma once
#include <cfloat>
#include <cstring>

#include "dsq_lgamma_int.h"
#include "dsq_linalg.h"
#include "dsq_wave.h"

namespace dsq {

// log of a positive count: 256-entry table (correctly rounded) for the bulk of RNA-seq counts, the lean
// log beyond it; the library log costs ~230 instructions on gfx950 and made the log-mean pass ALU bound
DSQ_HD double log_count(int c) { return c < 256 ? kLogInt[c] : flog((double)c); }

// ---------------------------------------------------------------- size factors, pass A
// logmean = mean_n log(y_n)  (-inf as soon as one count is zero); nonzero = any(y > 0)
// log_tab: the kernel's LDS copy of kLogInt (null: the table itself).  The pass is two dependent reads per sample
// (count, then its logarithm): unrolled so that four of each are in flight, the table read an LDS hit.
template <class Wv>
DSQ_HD void gene_logmean(const int32_t* y, int N, double& logmean, int& nonzero, const double* log_tab = nullptr) {
    double s = 0.0;
    int has_zero = 0, any_pos = 0;
    const auto tab = DSQ_AS_LDS(double, log_tab);
#pragma unroll 4
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const int v = y[n];
        if (v > 0) {
            s += (v < 256) ? (log_tab != nullptr ? tab[v] : kLogInt[v]) : flog((double)v);
            any_pos = 1;
        } else {
            has_zero = 1;
        }
    }
    s = Wv::sum(s);
    has_zero = Wv::sumi(has_zero);
    nonzero = Wv::sumi(any_pos) > 0 ? 1 : 0;
    logmean = has_zero ? -INFINITY : s / (double)N;
}

// ---------------------------------------------------------------- method of moments
struct MomOut {
    double normed_mean, rough, moments, mom;
};

// pinvXt rows of (X^T X)^-1 X^T so that beta_ols = pinvX @ normed  (sklearn
// LinearRegression(fit_intercept=False).fit(X, normed), utils.py:846-848)
template <class Wv, int P>
DSQ_HD MomOut mom_gene(const int32_t* y, const double* sf, const double* Xt, const double* pinvXt,
                       int ldx, int N, double s_mean_inv, double min_disp, double max_disp) {
    double s = 0.0, b[P];
#pragma unroll
    for (int j = 0; j < P; ++j) b[j] = 0.0;
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double v = (double)y[n] / sf[n];
        s += v;
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] += pinvXt[j * ldx + n] * v;
    }
    s = Wv::sum(s);
    Wv::template sum_n<P>(b);
    const double mean = s / (double)N;
    double ss = 0.0, rr = 0.0;
    const double dof = (double)(N - P);
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double v = (double)y[n] / sf[n];
        const double d = v - mean;
        ss += d * d;
        double yh = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * b[j];
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
    if (m != m) m = 0.0;                       // np.nan_to_num
    else if (m == INFINITY) m = DBL_MAX;
    else if (m == -INFINITY) m = -DBL_MAX;
    o.moments = m;
    o.mom = dmin(dmax(dmin(o.rough, o.moments), min_disp), max_disp);
    return o;
}

// MoM + linear-model mu_hat in one go (designs whose #cells == p, dds.py:747-756): both need the OLS
// coefficients of the normalised counts, so the fused routine sweeps the gene's row twice instead of four
// times.  Same arithmetic, same order as mom_gene / lin_mu_gene (bit-identical outputs).
template <class Wv, int P>
DSQ_HD MomOut mom_lin_mu_gene(const int32_t* y, const double* sf, const double* Xt, const double* pinvXt,
                              int ldx, int N, double s_mean_inv, double min_disp, double max_disp,
                              double min_mu, double* mu_out, double* coef_out = nullptr) {
    double s = 0.0, b[P];
#pragma unroll
    for (int j = 0; j < P; ++j) b[j] = 0.0;
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double v = (double)y[n] / sf[n];
        s += v;
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] += pinvXt[j * ldx + n] * v;
    }
    s = Wv::sum(s);
    Wv::template sum_n<P>(b);
    const double mean = s / (double)N;
    double ss = 0.0, rr = 0.0;
    const double dof = (double)(N - P);
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double sfn = sf[n];
        const double v = (double)y[n] / sfn;
        const double d = v - mean;
        ss += d * d;
        double yh = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * b[j];
        if (mu_out != nullptr) mu_out[n] = dmax(sfn * yh, min_mu);
        yh = dmax(yh, 1.0);
        rr += ((v - yh) * (v - yh) - yh) / (dof * yh * yh);
    }
    ss = Wv::sum(ss);
    rr = Wv::sum(rr);
    if (coef_out != nullptr && Wv::lane() == 0) {  // the dispersion kernel rebuilds mu_hat from these (k_alpha, ex.coef)
#pragma unroll
        for (int j = 0; j < P; ++j) coef_out[j] = b[j];
    }
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

// NG genes per wavefront (device kernels k_mom4 / k_mom_lin_mu4): the size factors and the rows of pinvXt / Xt are
// read once per sample and applied to NG count rows - the single-gene loops read 8 + 16 p bytes of shared vectors per
// 4 bytes of counts and ran at 1 TB/s of HBM.  Per gene the same operations in the same per-lane order, and the
// multi-value reduction is bit-identical to the single sums: same results as mom_lin_mu_gene up to the last bit of the
// normalised counts (the quotients y / sf are formed from one reciprocal per sample here, see below).
template <class Wv, int P, int NG>
DSQ_HD void mom_lin_mu_block(const int32_t* y0, int ldn, int n_valid, const double* sf, const double* Xt,
                             const double* pinvXt, int ldx, int N, double s_mean_inv, double min_disp,
                             double max_disp, double min_mu, double* mu0, double* coef0, MomOut (&out)[NG]) {
    const int32_t* yr[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) yr[k] = y0 + (size_t)(k < n_valid ? k : n_valid - 1) * ldn;
    double s[NG], b[NG * P];
#pragma unroll
    for (int k = 0; k < NG; ++k) s[k] = 0.0;
#pragma unroll
    for (int i = 0; i < NG * P; ++i) b[i] = 0.0;
    // y / sf for NG genes and two passes: ONE division per sample (the reciprocal) and, per quotient, a product with one
    // residual correction - the correctly rounded quotient but for rare double roundings (<= 1 ulp) - instead of 2 NG
    // IEEE divisions of ~30 dependent instructions each, which made this HBM-sized kernel ALU-bound (0.235 ms for 240 MB)
    auto quot = [](double yv, double sfn, double rs) {
        const double q = yv * rs;
        return fma(fma(-q, sfn, yv), rs, q);
    };
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double sfn = sf[n];
        const double rs = 1.0 / sfn;
        double pv[P];
#pragma unroll
        for (int j = 0; j < P; ++j) pv[j] = pinvXt[j * ldx + n];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const double v = quot((double)yr[k][n], sfn, rs);
            s[k] += v;
#pragma unroll
            for (int j = 0; j < P; ++j) b[k * P + j] += pv[j] * v;
        }
    }
    Wv::template sum_n<NG>(s);
    Wv::template sum_n<NG * P>(b);
    double mean[NG], acc[2 * NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) { mean[k] = s[k] / (double)N; acc[k] = 0.0; acc[NG + k] = 0.0; }
    const double dof = (double)(N - P);
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double sfn = sf[n];
        const double rs = 1.0 / sfn;
        double xv[P];
#pragma unroll
        for (int j = 0; j < P; ++j) xv[j] = Xt[j * ldx + n];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const double v = quot((double)yr[k][n], sfn, rs);
            const double d = v - mean[k];
            acc[k] += d * d;
            double yh = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) yh += xv[j] * b[k * P + j];
            if (mu0 != nullptr && k < n_valid) mu0[(size_t)k * ldn + n] = dmax(sfn * yh, min_mu);
            yh = dmax(yh, 1.0);
            acc[NG + k] += ((v - yh) * (v - yh) - yh) * frcp(dof * yh * yh);  // (yh >= 1: a normal positive number)
        }
    }
    Wv::template sum_n<2 * NG>(acc);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        if (coef0 != nullptr && Wv::lane() == 0 && k < n_valid) {
#pragma unroll
            for (int j = 0; j < P; ++j) coef0[(size_t)k * P + j] = b[k * P + j];
        }
        MomOut o;
        o.normed_mean = mean[k];
        o.rough = dmax(acc[NG + k], 0.0);
        const double var = acc[k] / (double)(N - 1);
        double m = (var - s_mean_inv * mean[k]) / (mean[k] * mean[k]);
        if (m != m) m = 0.0;
        else if (m == INFINITY) m = DBL_MAX;
        else if (m == -INFINITY) m = -DBL_MAX;
        o.moments = m;
        o.mom = dmin(dmax(dmin(o.rough, o.moments), min_disp), max_disp);
        out[k] = o;
    }
}

// ---------------------------------------------------------------- linear-model mu_hat
template <class Wv, int P>
DSQ_HD void lin_mu_gene(const int32_t* y, const double* sf, const double* Xt, const double* pinvXt,
                        int ldx, int N, double min_mu, double* mu_out) {
    double b[P];
#pragma unroll
    for (int j = 0; j < P; ++j) b[j] = 0.0;
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double v = (double)y[n] / sf[n];
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] += pinvXt[j * ldx + n] * v;
    }
    Wv::template sum_n<P>(b);
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        double yh = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * b[j];
        mu_out[n] = dmax(sf[n] * yh, min_mu);
    }
}

// ---------------------------------------------------------------- Wald test
enum WaldAlt { ALT_NONE = 0, ALT_GREATER_ABS = 1, ALT_LESS_ABS = 2, ALT_GREATER = 3, ALT_LESS = 4 };

struct WaldOut {
    double p, stat, se;
};

// Wald statistic from M = X^T W X (packed, WITHOUT ridge; W = mu/(1 + mu disp) at the UNclamped mu, ds.py:320-324)
template <int P>
DSQ_HD WaldOut wald_from_M(const double (&M)[Tri<P>::N], const double (&beta)[P], const double* ridge /*[P*P]*/,
                           const double* contrast /*[P]*/, double lfc_null, int alt) {
    constexpr int T = Tri<P>::N;
    double Hm[T], c[P], Hc[P], MHc[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        c[i] = contrast[i];
#pragma unroll
        for (int j = 0; j <= i; ++j) Hm[tri(i, j)] = M[tri(i, j)] + ridge[i * P + j];
    }
    chol<P>(Hm);
#pragma unroll
    for (int j = 0; j < P; ++j) Hc[j] = c[j];
    chol_solve<P>(Hm, Hc);                   // H c
    sym_matvec<P>(M, Hc, MHc);
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) q += Hc[j] * MHc[j];
    WaldOut o;
    o.se = sqrt(q);
    double stat = 0.0, pval;
    if (alt == ALT_NONE) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) t += c[j] * (beta[j] - lfc_null);
        stat = t / o.se;
        pval = 2.0 * norm_sf(fabs(stat));
    } else if (alt == ALT_GREATER) {
#pragma unroll
        for (int j = 0; j < P; ++j) stat += c[j] * np_fmax((beta[j] - lfc_null) / o.se, 0.0);
        pval = norm_sf(stat);
    } else if (alt == ALT_LESS) {
#pragma unroll
        for (int j = 0; j < P; ++j) stat += c[j] * np_fmin((beta[j] - lfc_null) / o.se, 0.0);
        pval = norm_sf(fabs(stat));
    } else if (alt == ALT_GREATER_ABS) {
#pragma unroll
        for (int j = 0; j < P; ++j)
            stat += c[j] * (dsign(beta[j]) * np_fmax((fabs(beta[j]) - lfc_null) / o.se, 0.0));
        pval = 2.0 * norm_sf(fabs(stat));
    } else {  // lessAbs: greater(-|null|) vs less(|null|)
        const double an = fabs(lfc_null);
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            sa += c[j] * np_fmax((beta[j] + an) / o.se, 0.0);
            sb += c[j] * np_fmin((beta[j] - an) / o.se, 0.0);
        }
        const double pa = norm_sf(sa), pb = norm_sf(fabs(sb));
        stat = (fabs(sb) < fabs(sa)) ? sb : sa;   // min(stat_above, stat_below, key=abs)
        pval = (pb > pa) ? pb : pa;              // max(pval_above, pval_below)
    }
    o.stat = stat;
    o.p = pval;
    return o;
}

// mu == nullptr: mu_n = sf_n exp(x_n . beta) is recomputed (what ds.py:320-324 builds on the
// host); otherwise the caller's mu row is used (Inference.wald_test contract).
template <class Wv, int P>
DSQ_HD WaldOut wald_gene(const double* mu, const double* sf, const double* Xt, int ldx, int N,
                         double disp, const double (&beta)[P], const double* ridge /*[P*P]*/,
                         const double* contrast /*[P]*/, double lfc_null, int alt) {
    constexpr int T = Tri<P>::N;
    double M[T];
#pragma unroll
    for (int k = 0; k < T; ++k) M[k] = 0.0;
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        double x[P];
        double eta = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) { x[j] = Xt[j * ldx + n]; eta += x[j] * beta[j]; }
        const double m = (mu != nullptr) ? mu[n] : sf[n] * exp(eta);
        const double w = m / (1.0 + m * disp);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const double xw = x[i] * w;
#pragma unroll
            for (int j = 0; j <= i; ++j) M[tri(i, j)] += xw * x[j];
        }
    }
    Wv::template sum_n<T>(M);
    return wald_from_M<P>(M, beta, ridge, contrast, lfc_null, alt);
}

// ---------------------------------------------------------------- Cook's distances
struct CellPlan {
    // samples grouped by design cell; only cells with >= 3 replicates are listed.
    // If n_cells == 0 the whole sample set is one pseudo-cell (utils.py:949-952, trim 1/8,
    // scale 1.51) and cell_index = 0..N-1.
    const int32_t* cell_offsets;  // [n_cells + 1]
    const int32_t* cell_index;    // [cell_offsets[n_cells]] sample ids
    int n_cells;
    int whole;                    // 1: no cell has >= 3 replicates
};

DSQ_HD int trim_class(int n) { return n >= 24 ? 2 : (n >= 4 ? 1 : 0); }  // >=23.5 / >=3.5

// sum of sorted buf[lo:hi) (buf ascending, length n)
template <class Wv>
DSQ_HD double range_sum(const double* buf, int lo, int hi) {
    double s = 0.0;
    for (int k = lo + Wv::lane(); k < hi; k += Wv::W) s += buf[k];
    return Wv::sum(s);
}

// ---- trimmed sums by selection instead of sorting
// order-preserving 64-bit key of a double (NaN of positive sign sorts last, like numpy.sort)
DSQ_HD unsigned long long trim_key(double v) {
    unsigned long long b;
    memcpy(&b, &v, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

constexpr int kTrimBins = 256;  // 8-bit digits; hist holds 2 * kTrimBins counters (two order statistics)

// sum of the sorted values buf[nt : n - nt) without sorting: radix-select the two boundary order
// statistics (ranks nt and n-nt-1) jointly, 8 bits per pass, starting at the first byte in which the
// values differ at all and stopping as soon as both buckets hold a single value; then one sweep adds
// up what lies strictly between them plus the boundary ties that fall inside the range.
// hist: 2 * kTrimBins counters private to the wave (LDS on the device).
template <class Wv>
DSQ_HD double trimmed_sum_select(const double* buf, int n, int nt, unsigned int* hist) {
    if (nt <= 0) {
        double s = 0.0;
        for (int k = Wv::lane(); k < n; k += Wv::W) s += buf[k];
        return Wv::sum(s);
    }
    double vmin = INFINITY, vmax = -INFINITY;
    for (int k = Wv::lane(); k < n; k += Wv::W) {
        const double v = buf[k];
        vmin = v < vmin ? v : vmin;
        vmax = v > vmax ? v : vmax;
    }
    vmin = -Wv::max(-vmin);
    vmax = Wv::max(vmax);
    const unsigned long long kmin = trim_key(vmin), kmax = trim_key(vmax);
    if (kmin == kmax) return (double)(n - 2 * nt) * vmin;
    int hb = 63;
    while (!(((kmin ^ kmax) >> hb) & 1ull)) --hb;
    int shift = (hb >> 3) << 3;  // byte that holds the first differing bit
    // prefix = key bits above the current byte (shared by all values at the start)
    unsigned long long pre[2];
    pre[0] = pre[1] = (shift == 56) ? 0ull : (kmin >> (shift + 8));
    int rank[2] = {nt, n - nt - 1};
    int cnt[2] = {n, n};
    constexpr int BPL = kTrimBins / (Wv::W < kTrimBins ? Wv::W : kTrimBins);  // bins per lane
    for (;;) {
        for (int b = Wv::lane(); b < 2 * kTrimBins; b += Wv::W) hist[b] = 0u;
        Wv::sync();
        for (int k = Wv::lane(); k < n; k += Wv::W) {
            const unsigned long long key = trim_key(buf[k]);
            const unsigned long long hi = (shift == 56) ? 0ull : (key >> (shift + 8));
            const int dg = (int)((key >> shift) & 255ull);
            if (hi == pre[0]) Wv::hist_add(hist + dg);
            if (hi == pre[1]) Wv::hist_add(hist + kTrimBins + dg);
        }
        Wv::sync();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned int* h = hist + t * kTrimBins;
            const int b0 = Wv::lane() * BPL;
            int c[BPL], tot = 0;
#pragma unroll
            for (int q = 0; q < BPL; ++q) { c[q] = (b0 + q < kTrimBins) ? (int)h[b0 + q] : 0; tot += c[q]; }
            int cum = Wv::excl_scan_i(tot);
            int fbin = 0, fcum = 0, fcnt = 0;
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
                const bool hit = rank[t] >= cum && rank[t] < cum + c[q];
                if (hit) { fbin = b0 + q; fcum = cum; fcnt = c[q]; }
                cum += c[q];
            }
            fbin = Wv::sumi(fbin); fcum = Wv::sumi(fcum); fcnt = Wv::sumi(fcnt);  // exactly one lane hits
            pre[t] = (pre[t] << 8) | (unsigned long long)fbin;
            rank[t] -= fcum;
            cnt[t] = fcnt;
        }
        if (shift == 0 || (cnt[0] == 1 && cnt[1] == 1)) break;
        shift -= 8;
    }
    // the two boundary values: any element whose key starts with the selected prefix
    double lo = -INFINITY, hi = -INFINITY;
    for (int k = Wv::lane(); k < n; k += Wv::W) {
        const double v = buf[k];
        const unsigned long long kk = trim_key(v) >> shift;
        if (kk == pre[0]) lo = v;
        if (kk == pre[1]) hi = v;
    }
    lo = Wv::max(lo);
    hi = Wv::max(hi);
    if (!(lo < hi)) return (double)(n - 2 * nt) * lo;
    double s = 0.0;
    int below_lo = 0, eq_lo = 0, below_hi = 0;
    for (int k = Wv::lane(); k < n; k += Wv::W) {
        const double v = buf[k];
        s += (v > lo && v < hi) ? v : 0.0;
        below_lo += v < lo ? 1 : 0;
        eq_lo += v == lo ? 1 : 0;
        below_hi += v < hi ? 1 : 0;
    }
    s = Wv::sum(s);
    below_lo = Wv::sumi(below_lo); eq_lo = Wv::sumi(eq_lo); below_hi = Wv::sumi(below_hi);
    return s + lo * (double)(below_lo + eq_lo - nt) + hi * (double)((n - nt) - below_hi);
}

// ---- trimmed sums of LARGE cells in one histogram pass
// The values of a cell are dropped into kBuckets order-preserving buckets (linear in the bit pattern of the double
// between the smallest and the largest value, i.e. logarithmic in the value), each bucket keeping its count and its
// SUM.  A prefix scan of the counts finds the two buckets that hold the boundary ranks; every bucket strictly between
// them contributes its sum, and only the (few) elements of the two boundary buckets are ranked individually.  Three
// light passes over the cell instead of the ~50 compare-exchange passes of a bitonic sort of 512 values (or the 5+
// passes of the radix selection above).  Samples with a ZERO count are kept out (markers < 0 in the buffer): they are
// the one large group of ties real count data has, and the caller adds their contribution in closed form.
#ifndef DSQ_BUCKETS
#define DSQ_BUCKETS 512
#endif
#ifndef DSQ_BUCKET_GATHER
#define DSQ_BUCKET_GATHER 128
#endif
constexpr int kBuckets = DSQ_BUCKETS;
constexpr int kBucketGather = DSQ_BUCKET_GATHER;  // elements a boundary bucket may hold; beyond: the caller takes the selection path
constexpr int kTrimBucketMin = 129; // cells from this size on take the bucket path (smaller ones sort in a few stages)
struct BucketWork {                 // wave-private LDS
    double sum[kBuckets];           // the 2 * kTrimBins counters of trimmed_sum_select, the fallback (the bucket path itself
                                    // only counts: bucket_rank_sum)
    unsigned int cnt[kBuckets];
    double edge[2][kBucketGather];
    unsigned int n_edge[2];
    unsigned int pad[2];
};
static_assert(kBuckets * sizeof(double) >= 2 * kTrimBins * sizeof(unsigned int), "the selection path borrows BucketWork::sum");

DSQ_HD unsigned long long pos_key(double v) {  // order-preserving for v >= +0
    unsigned long long b;
    memcpy(&b, &v, 8);
    return b;
}

// body(buf[k]) for this lane's k = lane, lane + W, ... < n, the values fetched U at a time BEFORE the bodies run: with an
// accessor that reads global memory (NormedValues) a wavefront keeps U loads in flight instead of one per trip through
// LDS atomics (k_robust_disp_lean was 76 % SQ_WAIT_ANY at one).  Same order of the calls per lane as the plain loop.
#ifndef DSQ_BATCH_U
#define DSQ_BATCH_U 4
#endif
template <int U, class Buf>
DSQ_HD void fetch_batch(const Buf& buf, int k0, int stride, int n, double (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = k0 + u * stride;
        v[u] = k < n ? buf[k] : 0.0;
    }
}
struct NormedValues;  // (below; its fetch_batch overload loads without branches)
template <int U>
DSQ_HD void fetch_batch(const NormedValues& V, int k0, int stride, int n, double (&v)[U]);

template <class Wv, int U = DSQ_BATCH_U, class Buf, class F>
DSQ_HD void for_each_batched_k(const Buf& buf, int n, F&& body) {  // body(k, buf[k])
    for (int k0 = Wv::lane(); k0 < n; k0 += Wv::W * U) {
        double v[U];
        fetch_batch<U>(buf, k0, Wv::W, n, v);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k0 + u * Wv::W < n) body(k0 + u * Wv::W, v[u]);
    }
}
template <class Wv, int U = DSQ_BATCH_U, class Buf, class F>
DSQ_HD void for_each_batched(const Buf& buf, int n, F&& body) {
    for_each_batched_k<Wv, U>(buf, n, [&](int, double v) { body(v); });
}

// out = sum of the elements of ranks j_lo .. j_hi (ascending, 0-based, inclusive) among the ACTIVE entries
// (buf[k] >= 0) of buf[0..n), of which there must be n_act.  false: not applicable here (a non-finite value, a
// boundary bucket with more than kBucketGather entries) - nothing but W has been written.
// range: the smallest and the largest active value when the caller already knows them (and that every active entry is
// finite) - saves the pass that finds them
// Buf: anything indexable by k in [0, n) - a buffer of doubles, or an accessor that recomputes the value (NormedValues)
template <class Wv, class Buf>
DSQ_HD bool bucket_rank_sum(const Buf& buf, int n, int n_act, int j_lo, int j_hi, BucketWork& W, double& out,
                            const double* range = nullptr) {
    out = 0.0;
    if (n_act <= 0 || j_hi < j_lo) return true;
    double vmin = INFINITY, vmax = -INFINITY;
    if (range != nullptr) {
        vmin = range[0]; vmax = range[1];
    } else {
        int seen = 0;
        for_each_batched<Wv>(buf, n, [&](double v) {
            if (v >= 0.0) {
                vmin = v < vmin ? v : vmin;
                vmax = v > vmax ? v : vmax;
                seen += 1;
            }
        });
        vmin = -Wv::max(-vmin);
        vmax = Wv::max(vmax);
        seen = Wv::sumi(seen);
        if (seen != n_act || !(vmax < INFINITY)) return false;  // NaN (fails v >= 0) or inf among the values
    }
    const unsigned long long kmin = pos_key(vmin), kmax = pos_key(vmax);
    if (kmin == kmax) { out = (double)(j_hi - j_lo + 1) * vmin; return true; }
    int shift = 0;
    while (((kmax - kmin) >> shift) >= (unsigned long long)kBuckets) ++shift;
    for (int b = Wv::lane(); b < kBuckets; b += Wv::W) W.cnt[b] = 0u;
    for (int t = Wv::lane(); t < 2; t += Wv::W) W.n_edge[t] = 0u;
    Wv::sync();
    for_each_batched<Wv>(buf, n, [&](double v) {
        if (v >= 0.0) {
            const int b = (int)((pos_key(v) - kmin) >> shift);
            Wv::hist_add(&W.cnt[b]);
        }
    });
    Wv::sync();
    // the buckets of the two boundary ranks and the sum of everything strictly between them
    constexpr int BPL = kBuckets / (Wv::W < kBuckets ? Wv::W : kBuckets);  // consecutive buckets per lane
    const int b0 = Wv::lane() * BPL;
    int tot = 0;
    for (int q = 0; q < BPL; ++q) tot += (int)W.cnt[b0 + q];
    const int cum0 = Wv::excl_scan_i(tot);
    int fb[2] = {0, 0}, fc[2] = {0, 0}, fn[2] = {0, 0};
    {
        int cum = cum0;
        for (int q = 0; q < BPL; ++q) {
            const int c = (int)W.cnt[b0 + q];
            if (j_lo >= cum && j_lo < cum + c) { fb[0] = b0 + q; fc[0] = cum; fn[0] = c; }
            if (j_hi >= cum && j_hi < cum + c) { fb[1] = b0 + q; fc[1] = cum; fn[1] = c; }
            cum += c;
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) { fb[t] = Wv::sumi(fb[t]); fc[t] = Wv::sumi(fc[t]); fn[t] = Wv::sumi(fn[t]); }  // one lane hits
    if (fn[0] > kBucketGather || fn[1] > kBucketGather) return false;
    // the entries of the boundary buckets, ranked one against the other; everything strictly between the two buckets is
    // added up in registers on the way (the histogram pass counts only: an LDS atomic on a double per value cost more
    // than the rest of that pass)
    const bool one = fb[0] == fb[1];
    double inside = 0.0;
    for_each_batched<Wv>(buf, n, [&](double v) {
        if (v >= 0.0) {
            const int b = (int)((pos_key(v) - kmin) >> shift);
            inside += (b > fb[0] && b < fb[1]) ? v : 0.0;
            if (b == fb[0]) W.edge[0][Wv::slot_add(&W.n_edge[0])] = v;
            else if (b == fb[1]) W.edge[1][Wv::slot_add(&W.n_edge[1])] = v;
        }
    });
    Wv::sync();
    double part = 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1 && one) break;
        const int m = fn[t];
        // in-bucket ranks to keep
        const int r_from = (t == 0) ? j_lo - fc[0] : 0;
        const int r_to = (t == 1) ? j_hi - fc[1] : (one ? j_hi - fc[0] : m - 1);
        for (int i = Wv::lane(); i < m; i += Wv::W) {
            const double e = W.edge[t][i];
            int r = 0;
            for (int j = 0; j < m; ++j) {
                const double o = W.edge[t][j];
                r += (o < e || (o == e && j < i)) ? 1 : 0;
            }
            part += (r >= r_from && r <= r_to) ? e : 0.0;
        }
    }
    out = Wv::sum(inside + part);
    Wv::sync();  // W is free again
    return true;
}

// The same sum by radix selection - the algorithm of trimmed_sum_select over an accessor, among the ACTIVE entries only:
// what bucket_rank_sum's callers fall back to where no buffer of the cell's values exists (robust_disp_gene_lean; cells of
// tens of thousands of samples put more than kBucketGather values into a boundary bucket as a rule).  Every active entry
// must be finite; range: their smallest and largest value.  hist: 2 * kTrimBins counters private to the wave.
template <class Wv, class Buf>
DSQ_HD double select_rank_sum(const Buf& buf, int n, int n_act, int j_lo, int j_hi, unsigned int* hist,
                              const double* range) {
    if (n_act <= 0 || j_hi < j_lo) return 0.0;
    const unsigned long long kmin = pos_key(range[0]), kmax = pos_key(range[1]);
    if (kmin == kmax) return (double)(j_hi - j_lo + 1) * range[0];
    int hb = 63;
    while (!(((kmin ^ kmax) >> hb) & 1ull)) --hb;
    int shift = (hb >> 3) << 3;  // byte that holds the first differing bit
    unsigned long long pre[2];
    pre[0] = pre[1] = (shift == 56) ? 0ull : (kmin >> (shift + 8));
    int rank[2] = {j_lo, j_hi};
    int cnt[2] = {n_act, n_act};
    constexpr int BPL = kTrimBins / (Wv::W < kTrimBins ? Wv::W : kTrimBins);  // bins per lane
    for (;;) {
        for (int b = Wv::lane(); b < 2 * kTrimBins; b += Wv::W) hist[b] = 0u;
        Wv::sync();
        const unsigned long long p0 = pre[0], p1 = pre[1];
        for_each_batched<Wv>(buf, n, [&](double v) {
            if (v >= 0.0) {
                const unsigned long long key = pos_key(v);
                const unsigned long long hi = (shift == 56) ? 0ull : (key >> (shift + 8));
                const int dg = (int)((key >> shift) & 255ull);
                if (hi == p0) Wv::hist_add(hist + dg);
                if (hi == p1) Wv::hist_add(hist + kTrimBins + dg);
            }
        });
        Wv::sync();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned int* h = hist + t * kTrimBins;
            const int b0 = Wv::lane() * BPL;
            int c[BPL], tot = 0;
#pragma unroll
            for (int q = 0; q < BPL; ++q) { c[q] = (b0 + q < kTrimBins) ? (int)h[b0 + q] : 0; tot += c[q]; }
            int cum = Wv::excl_scan_i(tot);
            int fbin = 0, fcum = 0, fcnt = 0;
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
                const bool hit = rank[t] >= cum && rank[t] < cum + c[q];
                if (hit) { fbin = b0 + q; fcum = cum; fcnt = c[q]; }
                cum += c[q];
            }
            fbin = Wv::sumi(fbin); fcum = Wv::sumi(fcum); fcnt = Wv::sumi(fcnt);  // exactly one lane hits
            pre[t] = (pre[t] << 8) | (unsigned long long)fbin;
            rank[t] -= fcum;
            cnt[t] = fcnt;
        }
        if (shift == 0 || (cnt[0] == 1 && cnt[1] == 1)) break;
        shift -= 8;
    }
    // the two boundary values: any active element whose key starts with the selected prefix
    double lo = -INFINITY, hi = -INFINITY;
    {
        const unsigned long long p0 = pre[0], p1 = pre[1];
        for_each_batched<Wv>(buf, n, [&](double v) {
            if (v >= 0.0) {
                const unsigned long long kk = pos_key(v) >> shift;
                if (kk == p0) lo = v;
                if (kk == p1) hi = v;
            }
        });
    }
    lo = Wv::max(lo);
    hi = Wv::max(hi);
    if (!(lo < hi)) return (double)(j_hi - j_lo + 1) * lo;
    double sm = 0.0;
    int below_lo = 0, eq_lo = 0, below_hi = 0;
    for_each_batched<Wv>(buf, n, [&](double v) {
        if (v >= 0.0) {
            sm += (v > lo && v < hi) ? v : 0.0;
            below_lo += v < lo ? 1 : 0;
            eq_lo += v == lo ? 1 : 0;
            below_hi += v < hi ? 1 : 0;
        }
    });
    sm = Wv::sum(sm);
    below_lo = Wv::sumi(below_lo); eq_lo = Wv::sumi(eq_lo); below_hi = Wv::sumi(below_hi);
    Wv::sync();  // hist is free again
    return sm + lo * (double)(below_lo + eq_lo - j_lo) + hi * (double)((j_hi + 1) - below_hi);
}

struct CooksOut {
    double robust_disp;
    int any_gt_all;      // any sample with cooks > cutoff                     (dds.py:1325-1326)
    int any_gt_use;      // any(cooks[use_for_max] > cutoff)                   (dds.py:1093)
    int any_gt_use_nr;   // same with replaceable samples zeroed               (dds.py:1089, 1458)
    int few_above;       // (#samples with y > y[argmax cooks]) < 3           (dds.py:1097-1101)
};

// Trimmed statistics of a design cell: cells of fewer than kTrimBucketMin samples are SORTED in the wave's LDS
// segment (bitonic network: a few stages; 6x faster than selection at 30 cells x 17), larger ones take the bucket
// path above, the radix selection being their fallback.
//
// scratch: >= max cell doubles (the next power of two for cells that are sorted), hist: a BucketWork when a cell has
// kTrimBucketMin samples or more (both wave-private LDS on the device).
// ---- trimmed variances of SMALL cells, several cells per pass
// A design with many small cells (c4: 30 cells of 16-17 samples) leaves most of a wavefront idle when the cells are
// sorted one after the other (a 32-element bitonic stage has 16 compare-exchanges for 64 lanes).  Here kSegBatch / L
// cells - L = the power of two that holds the largest cell - sit side by side in the wave's buffer as segments of L and
// are sorted by ONE network (every stage: kSegBatch / 2 compare-exchanges, one per lane); the trimmed sums are
// per-segment accumulators in LDS.  Same arithmetic per cell as the one-by-one path, up to the order of the additions
// inside a trimmed sum.
constexpr int kSegBatch = 128;   // elements per pass
constexpr int kSegMaxCell = 64;  // cells up to this size take the batched path

template <class Wv>
DSQ_HD void seg_bitonic_stage(double* buf, int L, int k, int j, bool merge_only) {
    const int half = L >> 1;
    for (int p = Wv::lane(); p < kSegBatch / 2; p += Wv::W) {
        const int seg = p / half, ii = p % half;
        const int lo_l = ((ii & ~(j - 1)) << 1) | (ii & (j - 1));
        const int lo = seg * L + lo_l, hi = lo | j;
        const bool up = merge_only || ((lo_l & k) == 0);
        const double a = buf[lo], b = buf[hi];
        const bool gt = (a > b) || (a != a && b == b);  // NaNs sort last (numpy.sort)
        if (gt == up) { buf[lo] = b; buf[hi] = a; }
    }
    Wv::sync();
}

// max over the cells c0 .. c0 + kSegBatch / L - 1 of scale * trimmed mean of squared errors (NaN wins, as the caller's max)
// segsum: kSegBatch / L doubles (LDS)
template <class Wv>
DSQ_HD double seg_trimmed_variances(const int32_t* y, const double* sf, const CellPlan& C, int c0, int L, double* buf,
                                    double* segsum, double vmax) {
    const double ratios[3] = {1.0 / 3.0, 1.0 / 4.0, 1.0 / 8.0};
    const double scales[3] = {2.04, 1.86, 1.51};
    const int nseg = kSegBatch / L;
    // this lane's elements e = lane, lane + W, ...: segment, position, and the trimming ranks of the segment's cell
    auto cell_of_elem = [&](int e, int& k, int& n, int& nt, int& cls, int& beg) {
        const int c = c0 + e / L;
        k = e % L;
        beg = 0; n = 0; nt = 0; cls = 0;
        if (c < C.n_cells) {
            beg = C.cell_offsets[c];
            n = C.cell_offsets[c + 1] - beg;
            cls = trim_class(n);
            nt = (int)floor((double)n * ratios[cls]);
        }
    };
    for (int e = Wv::lane(); e < kSegBatch; e += Wv::W) {
        int k, n, nt, cls, beg;
        cell_of_elem(e, k, n, nt, cls, beg);
        double v = INFINITY;
        if (k < n) {
            const int sidx = C.cell_index[beg + k];
            v = (double)y[sidx] / sf[sidx];
        }
        buf[e] = v;
    }
    for (int q = Wv::lane(); q < nseg; q += Wv::W) segsum[q] = 0.0;
    Wv::sync();
    for (int k = 2; k <= L; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) seg_bitonic_stage<Wv>(buf, L, k, j, false);
    for (int e = Wv::lane(); e < kSegBatch; e += Wv::W) {
        int k, n, nt, cls, beg;
        cell_of_elem(e, k, n, nt, cls, beg);
        if (k >= nt && k < n - nt) Wv::cell_add(&segsum[e / L], buf[e]);
    }
    Wv::sync();
    // squared errors around the segment's trimmed mean (ascending -> decreasing-then-increasing: one merge sorts them)
    for (int e = Wv::lane(); e < kSegBatch; e += Wv::W) {
        int k, n, nt, cls, beg;
        cell_of_elem(e, k, n, nt, cls, beg);
        if (k < n) {
            const double d = buf[e] - segsum[e / L] / (double)(n - 2 * nt);
            buf[e] = d * d;
        }
    }
    Wv::sync();
    for (int q = Wv::lane(); q < nseg; q += Wv::W) segsum[q] = 0.0;
    Wv::sync();
    for (int j = L >> 1; j > 0; j >>= 1) seg_bitonic_stage<Wv>(buf, L, L, j, true);
    for (int e = Wv::lane(); e < kSegBatch; e += Wv::W) {
        int k, n, nt, cls, beg;
        cell_of_elem(e, k, n, nt, cls, beg);
        if (k >= nt && k < n - nt) Wv::cell_add(&segsum[e / L], buf[e]);
    }
    Wv::sync();
    for (int q = Wv::lane(); q < nseg; q += Wv::W) {
        if (c0 + q < C.n_cells) {
            const int n = C.cell_offsets[c0 + q + 1] - C.cell_offsets[c0 + q];
            const int cls = trim_class(n);
            const int nt = (int)floor((double)n * ratios[cls]);
            const double tv = scales[cls] * (segsum[q] / (double)(n - 2 * nt));
            vmax = (tv > vmax || tv != tv) ? tv : vmax;
        }
    }
    Wv::sync();  // buf and segsum are free again
    return vmax;
}

// The k-th normalised count of a design cell (y / sf, as y * (1 / sf)), or its squared error against `tm`; -1 for a
// zero count (inactive for bucket_rank_sum) - recomputed from the gene's row on every access instead of being kept in
// a wave-private LDS buffer: the row is read from L1 / L2 three more times, the LDS footprint of a wavefront drops from
// next_pow2(N) doubles (64 KB at N = 5000: two wavefronts per CU) to the bucket table (8 KB)
struct NormedValues {
    const int32_t* y;
    const double* sf;
    const int32_t* idx;  // the cell's sample indices, or null: samples 0 .. n-1
    double tm;
    bool squared;
    DSQ_HD double operator[](int k) const {
        const int s = idx != nullptr ? idx[k] : k;
        const int yi = y[s];
        if (yi == 0) return -1.0;
        const double v = (double)yi * frcp_g(sf[s]);
        if (!squared) return v;
        const double d = v - tm;
        return d * d;
    }
};

// U values of a NormedValues at once, without a branch between the loads: the index loads go out together, then the
// count and size-factor loads (the one-value accessor is a chain of three dependent loads with a branch on the count
// in the middle; the compiler keeps that chain per value).  Positions past n re-read position k0 and are ignored by
// for_each_batched.
template <int U>
DSQ_HD void fetch_batch(const NormedValues& V, int k0, int stride, int n, double (&v)[U]) {
    int s[U];
    if (V.idx != nullptr) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * stride;
            s[u] = V.idx[k < n ? k : k0];
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * stride;
            s[u] = k < n ? k : k0;
        }
    }
    int yi[U];
    double f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        yi[u] = V.y[s[u]];
        f[u] = V.sf[s[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        double x = (double)yi[u] * frcp_g(f[u]);
        if (V.squared) {
            const double d = x - V.tm;
            x = d * d;
        }
        v[u] = yi[u] == 0 ? -1.0 : x;
    }
}

// flags[n]: bit0 use_for_max (cell >= 3 replicates), bit1 replaceable (cell >= min_replicates)
// Robust dispersion of utils.robust_method_of_moments_disp (utils.py:914-960): per design cell the trimmed
// variance of the normalised counts around their trimmed mean, the largest cell variance vs the overall mean.
// Depends on the counts, the size factors and the design cells only - not on any fit.
// BIG: the design has a cell of kTrimBucketMin samples or more (the bucket path is compiled in; without it the kernel
// needs half the registers, which the small-cell designs turn into occupancy)
// seg_len > 0: every cell has at most seg_len <= kSegMaxCell samples (a power of two) and scratch holds kSegBatch values
// followed by kSegBatch / seg_len sums: the cells go through seg_trimmed_variances, several per pass
template <class Wv, bool BIG = true, class Sorter>
DSQ_HD double robust_disp_gene(const int32_t* y, const double* sf, const CellPlan& C, int N, double* scratch,
                               unsigned int* hist, Sorter&& sorter, int seg_len = 0) {
    const double ratios[3] = {1.0 / 3.0, 1.0 / 4.0, 1.0 / 8.0};
    const double scales[3] = {2.04, 1.86, 1.51};
    double vmax = -INFINITY;
    const int ncell = C.whole ? 1 : C.n_cells;
    double cell_total = 0.0;  // per-lane sum of the normalised counts of the cells that took the bucket path ...
    int cells_summed = 0;     // ... and how many samples those cells hold
    const bool batched = !BIG && seg_len > 0 && !C.whole;
    if (batched) {
        for (int c0 = 0; c0 < ncell; c0 += kSegBatch / seg_len) {
            const double v = seg_trimmed_variances<Wv>(y, sf, C, c0, seg_len, scratch, scratch + kSegBatch, vmax);
            vmax = v;
        }
        // (per-lane partial maxima: combine; a NaN anywhere wins)
        const double any_nan = Wv::max(vmax != vmax ? 1.0 : 0.0);
        vmax = Wv::max(vmax != vmax ? -INFINITY : vmax);
        if (any_nan > 0.0) vmax = NAN;
    }
    for (int c = batched ? ncell : 0; c < ncell; ++c) {
        const int beg = C.whole ? 0 : C.cell_offsets[c];
        const int end = C.whole ? N : C.cell_offsets[c + 1];
        const int n = end - beg;
        const int cls = C.whole ? 2 : trim_class(n);
        const int nt = (int)floor((double)n * ratios[cls]);
        double tm, ts;
        bool done = false;
        if (BIG && n >= kTrimBucketMin) {
            // large cell: one histogram pass per trimmed sum (bucket_rank_sum).  The samples with a zero count - the
            // smallest values, all equal - stay out of the buckets: they add nothing to the first sum, and their squared
            // error (0 - tm)^2 enters the second one as a block of `zeros` equal values at a known rank.
            // (the normalised count as y * (1 / sf): a reciprocal with two Newton steps instead of an IEEE division - an ulp
            // of difference in values that only enter trimmed sums; the same pass finds the range of the non-zero values
            // and adds up the cell for the overall mean of utils.py:954)
            BucketWork& W = *(BucketWork*)hist;
            int zeros = 0, bad = 0;
            double lo1 = INFINITY, hi1 = -INFINITY;
            const NormedValues V{y, sf, C.whole ? nullptr : C.cell_index + beg, 0.0, false};
            for_each_batched_k<Wv>(V, n, [&](int k, double v) {  // (v = -1: a zero count)
                scratch[k] = v;
                zeros += v < 0.0 ? 1 : 0;
                if (!(v < 0.0)) {
                    bad |= (v >= 0.0 && v < INFINITY) ? 0 : 1;
                    lo1 = v < lo1 ? v : lo1;
                    hi1 = v > hi1 ? v : hi1;
                    cell_total += v;
                }
            });
            zeros = Wv::sumi(zeros);
            bad = Wv::sumi(bad);
            double range[2] = {-Wv::max(-lo1), Wv::max(hi1)};
            cells_summed += n;
            Wv::sync();
            const int r_lo = nt, r_hi = n - nt - 1, n_act = n - zeros;
            double s1 = 0.0, s2 = 0.0;
            bool ok = bucket_rank_sum<Wv>(scratch, n, n_act, r_lo > zeros ? r_lo - zeros : 0, r_hi - zeros, W, s1,
                                          bad == 0 ? range : nullptr);
            double tm2 = 0.0;
            if (ok) {
                tm = s1 / (double)(n - 2 * nt);
                const double d0 = 0.0 - tm;
                tm2 = d0 * d0;
                int below = 0;  // values whose squared error sorts before the block of the zero counts
                double lo2 = INFINITY, hi2 = -INFINITY;
                bad = 0;
                for (int k = Wv::lane(); k < n; k += Wv::W) {
                    const double v = scratch[k];
                    if (v >= 0.0) {
                        const double d = v - tm;
                        const double q = d * d;
                        scratch[k] = q;
                        below += q < tm2 ? 1 : 0;
                        bad |= (q >= 0.0 && q < INFINITY) ? 0 : 1;
                        lo2 = q < lo2 ? q : lo2;
                        hi2 = q > hi2 ? q : hi2;
                    }
                }
                below = Wv::sumi(below);
                bad = Wv::sumi(bad);
                range[0] = -Wv::max(-lo2); range[1] = Wv::max(hi2);
                Wv::sync();
                // ranks among the non-zero samples that fall into [r_lo, r_hi] once the block sits at [below, below + zeros)
                const int j_lo = r_lo < below ? r_lo : (r_lo - zeros > below ? r_lo - zeros : below);
                const int j_hi = r_hi < below ? r_hi : (r_hi < below + zeros ? below - 1 : r_hi - zeros);
                const int b_lo = r_lo > below ? r_lo : below, b_hi = r_hi < below + zeros - 1 ? r_hi : below + zeros - 1;
                ok = bucket_rank_sum<Wv>(scratch, n, n_act, j_lo, j_hi, W, s2, bad == 0 ? range : nullptr);
                if (ok) {
                    ts = s2 + (b_hi >= b_lo ? (double)(b_hi - b_lo + 1) * tm2 : 0.0);
                    done = true;
                } else {  // selection on the squared errors, the zero counts back in
                    for (int k = Wv::lane(); k < n; k += Wv::W)
                        if (scratch[k] < 0.0) scratch[k] = tm2;
                    Wv::sync();
                    ts = trimmed_sum_select<Wv>(scratch, n, nt, (unsigned int*)W.sum);
                    done = true;
                }
            }
        }
        if (!done) {
        // trimmed mean of normalised counts
        for (int k = Wv::lane(); k < n; k += Wv::W) {
            const int sidx = C.whole ? k : C.cell_index[beg + k];
            scratch[k] = (double)y[sidx] / sf[sidx];
        }
        const bool by_sort = n < kTrimBucketMin;
        if (by_sort) {
            sorter(scratch, n);
            tm = range_sum<Wv>(scratch, nt, n - nt) / (double)(n - 2 * nt);
        } else {
            Wv::sync();
            tm = trimmed_sum_select<Wv>(scratch, n, nt, hist) / (double)(n - 2 * nt);
        }
        // trimmed mean of squared errors, values transformed in place.  After a sort the buffer is
        // ascending, so (v - tm)^2 is a decreasing-then-increasing (bitonic) sequence: one bitonic MERGE
        // (log2 L stages) sorts it instead of a second full sort (log2 L (log2 L + 1) / 2 stages).
        for (int k = Wv::lane(); k < n; k += Wv::W) {
            const double d = scratch[k] - tm;
            scratch[k] = d * d;
        }
        if (by_sort) {
            sorter.merge(scratch, n);
            ts = range_sum<Wv>(scratch, nt, n - nt);
        } else {
            Wv::sync();
            ts = trimmed_sum_select<Wv>(scratch, n, nt, hist);
        }
        }
        const double tv = scales[cls] * (ts / (double)(n - 2 * nt));
        vmax = (tv > vmax || tv != tv) ? tv : vmax;
    }
    // mean of normalised counts over ALL samples (utils.py:954): already summed when the bucket path saw every sample
    double m;
    if (BIG && cells_summed == N) {
        m = Wv::sum(cell_total) / (double)N;
    } else {
        double s = 0.0;
        for (int n = Wv::lane(); n < N; n += Wv::W) s += (double)y[n] / sf[n];
        m = Wv::sum(s) / (double)N;
    }
    double ar = (vmax - m) / (m * m);
    ar = (ar > 0.04) ? ar : 0.04;  // np.maximum(alpha, 0.04) (NaN -> stays NaN in numpy; see below)
    if (vmax != vmax) ar = vmax;
    return ar;
}

// robust_disp_gene without a per-wave buffer of the cell's values (NormedValues): the bucket path for cells of at least
// kTrimBucketMin samples, radix selection over the same accessor for smaller ones and where a boundary bucket holds too
// many values - cells of any size (round 6; it used to hand such genes back to the buffered kernel, whose LDS ends at
// ~13 000 samples per cell).  failed = true: a value was not finite - the caller hands the gene to robust_disp_gene, which
// orders NaNs as numpy.sort does.
template <class Wv>
DSQ_HD double robust_disp_gene_lean(const int32_t* y, const double* sf, const CellPlan& C, int N, BucketWork& W,
                                    bool& failed) {
    const double ratios[3] = {1.0 / 3.0, 1.0 / 4.0, 1.0 / 8.0};
    const double scales[3] = {2.04, 1.86, 1.51};
    failed = false;
    double vmax = -INFINITY, cell_total = 0.0;
    const int ncell = C.whole ? 1 : C.n_cells;
    int cells_summed = 0;
    for (int c = 0; c < ncell; ++c) {
        const int beg = C.whole ? 0 : C.cell_offsets[c];
        const int end = C.whole ? N : C.cell_offsets[c + 1];
        const int n = end - beg;
        const int cls = C.whole ? 2 : trim_class(n);
        const int nt = (int)floor((double)n * ratios[cls]);
        NormedValues V{y, sf, C.whole ? nullptr : C.cell_index + beg, 0.0, false};
        int zeros = 0, bad = 0;
        double lo1 = INFINITY, hi1 = -INFINITY;
        for_each_batched<Wv>(V, n, [&](double v) {
            zeros += v < 0.0 ? 1 : 0;
            if (!(v < 0.0)) {
                bad |= (v >= 0.0 && v < INFINITY) ? 0 : 1;
                lo1 = v < lo1 ? v : lo1;
                hi1 = v > hi1 ? v : hi1;
                cell_total += v;
            }
        });
        zeros = Wv::sumi(zeros);
        bad = Wv::sumi(bad);
        double range[2] = {-Wv::max(-lo1), Wv::max(hi1)};
        cells_summed += n;
        if (bad != 0) { failed = true; return NAN; }
        const int r_lo = nt, r_hi = n - nt - 1, n_act = n - zeros;
        double s1 = 0.0, s2 = 0.0;
        // (small cells, and cells whose boundary bucket holds more than kBucketGather values: radix selection over the
        // same accessor - no buffer either way, so a cell may have any number of samples)
        const bool by_bucket = n >= kTrimBucketMin;
        if (!by_bucket || !bucket_rank_sum<Wv>(V, n, n_act, r_lo > zeros ? r_lo - zeros : 0, r_hi - zeros, W, s1, range))
            s1 = select_rank_sum<Wv>(V, n, n_act, r_lo > zeros ? r_lo - zeros : 0, r_hi - zeros, (unsigned int*)W.sum,
                                     range);
        const double tm = s1 / (double)(n - 2 * nt);
        const double d0 = 0.0 - tm;
        const double tm2 = d0 * d0;
        V.tm = tm;
        V.squared = true;
        int below = 0;  // values whose squared error sorts before the block of the zero counts
        double lo2 = INFINITY, hi2 = -INFINITY;
        for_each_batched<Wv>(V, n, [&](double q) {
            if (q >= 0.0) {
                below += q < tm2 ? 1 : 0;
                bad |= (q < INFINITY) ? 0 : 1;
                lo2 = q < lo2 ? q : lo2;
                hi2 = q > hi2 ? q : hi2;
            }
        });
        below = Wv::sumi(below);
        bad = Wv::sumi(bad);
        range[0] = -Wv::max(-lo2); range[1] = Wv::max(hi2);
        if (bad != 0) { failed = true; return NAN; }
        // ranks among the non-zero samples that fall into [r_lo, r_hi] once the block sits at [below, below + zeros)
        const int j_lo = r_lo < below ? r_lo : (r_lo - zeros > below ? r_lo - zeros : below);
        const int j_hi = r_hi < below ? r_hi : (r_hi < below + zeros ? below - 1 : r_hi - zeros);
        const int b_lo = r_lo > below ? r_lo : below, b_hi = r_hi < below + zeros - 1 ? r_hi : below + zeros - 1;
        if (!by_bucket || !bucket_rank_sum<Wv>(V, n, n_act, j_lo, j_hi, W, s2, range))
            s2 = select_rank_sum<Wv>(V, n, n_act, j_lo, j_hi, (unsigned int*)W.sum, range);
        const double ts = s2 + (b_hi >= b_lo ? (double)(b_hi - b_lo + 1) * tm2 : 0.0);
        const double tv = scales[cls] * (ts / (double)(n - 2 * nt));
        vmax = (tv > vmax || tv != tv) ? tv : vmax;
    }
    double m;
    if (cells_summed == N) {
        m = Wv::sum(cell_total) / (double)N;
    } else {
        double s = 0.0;
        for (int n = Wv::lane(); n < N; n += Wv::W) s += (double)y[n] / sf[n];
        m = Wv::sum(s) / (double)N;
    }
    double ar = (vmax - m) / (m * m);
    ar = (ar > 0.04) ? ar : 0.04;
    if (vmax != vmax) ar = vmax;
    return ar;
}

// Per-sample accumulator of the Cook's bookkeeping (dds.py:1034-1040, 1066-1110, 1325-1326): feed every
// sample's (y, mu, hat) once, finish() reduces over the wave.  Used by cooks_gene (mu / hat rows from memory)
// and by the epilogue of the LFC fit (mu / hat straight from the IRLS registers).
template <class Wv>
struct CooksAcc {
    double ar, cutoff, invP;
    int g_all = 0, g_use = 0, g_use_nr = 0;
    double best = -INFINITY;
    double best_y = 0.0;  // the count of the sample that holds `best` (finish_counted: no second look at the row)
    int best_idx = 0x7fffffff;
    bool best_nan = false;
    DSQ_HD CooksAcc(double robust_disp, double cutoff_, int P) : ar(robust_disp), cutoff(cutoff_), invP(1.0 / (double)P) {}
    DSQ_HD double add(int n, double yv, double mv, double h, int fl) {
        // (y - mu)^2 / V / p * h / (1 - h)^2 (dds.py:1034-1040) with reciprocals instead of three IEEE divisions
        // (~30 instructions each; the per-sample epilogue of the LFC fit was a third of that kernel)
        const double V = (mv * mv) * ar + mv;
        const double r = yv - mv;
        const double omh = 1.0 - h;
        const double ck = ((r * r) * invP) * h * frcp_g(V * (omh * omh));  // (one reciprocal: round 6)
        const bool gt = ck > cutoff;
        g_all |= gt ? 1 : 0;
        g_use |= (gt && (fl & 1)) ? 1 : 0;
        g_use_nr |= (gt && (fl & 1) && !(fl & 2)) ? 1 : 0;
        // np.argmax: first NaN wins, else first maximum - whatever order the samples arrive in (the mixed-design kernels
        // walk them sorted by design cell): ties go to the smaller sample index.  Selects, no branches (round 6: the
        // four-way branch of this update was a sixth of the mixed-design LFC launch).
        const bool isn = (ck != ck);
        const bool take = isn ? (!best_nan || n < best_idx) : (!best_nan && (ck > best || (ck == best && n < best_idx)));
        best = (take && !isn) ? ck : best;
        best_idx = take ? n : best_idx;
        best_y = take ? yv : best_y;
        best_nan = best_nan || isn;
        return ck;
    }
    // wave reduction of the flags and of the argmax; returns the winning sample's index and count
    DSQ_HD void reduce(CooksOut& o, int& bi, double& yref) {
        o.robust_disp = ar;
        o.any_gt_all = Wv::sumi(g_all) > 0;
        o.any_gt_use = Wv::sumi(g_use) > 0;
        o.any_gt_use_nr = Wv::sumi(g_use_nr) > 0;
        // wave argmax: (nan first, then value desc, then index asc)
        const int any_nan = Wv::sumi(best_nan ? 1 : 0);
        int cand;
        if (any_nan > 0) {
            cand = best_nan ? best_idx : 0x7fffffff;
        } else {
            const double wmax = Wv::max(best);
            cand = (best == wmax) ? best_idx : 0x7fffffff;
        }
        const double ci = -Wv::max(-(double)cand);  // min index over lanes
        bi = (int)ci;
        yref = Wv::max((cand == bi && best_idx == bi) ? best_y : -INFINITY);  // (one lane holds sample bi)
    }
    DSQ_HD CooksOut finish(const int32_t* y, int N) {
        CooksOut o;
        int bi;
        double yref_d;
        reduce(o, bi, yref_d);
        int above = 0;
        if (bi >= 0 && bi < N) {
            const int yref = y[bi];
            for (int n = Wv::lane(); n < N; n += Wv::W) above += (y[n] > yref) ? 1 : 0;
        }
        o.few_above = Wv::sumi(above) < 3;
        return o;
    }
    // the same with the caller counting the samples above the winner's count (it holds the gene's counts closer than the
    // global row: count_above(yref) -> this lane's share of #{n: y_n > yref})
    template <class F>
    DSQ_HD CooksOut finish_counted(int N, F&& count_above) {
        CooksOut o;
        int bi;
        double yref;
        reduce(o, bi, yref);
        int above = 0;
        if (bi >= 0 && bi < N) above = count_above((int)yref);
        o.few_above = Wv::sumi(above) < 3;
        return o;
    }
};

// scratch: >= max cell doubles (next power of two when the cell is sorted), hist: 2 * kTrimBins
// counters (both wave-private LDS on the device).
// flags[n]: bit0 use_for_max (cell >= 3 replicates), bit1 replaceable (cell >= min_replicates)
template <class Wv, bool BIG = true, class Sorter>
DSQ_HD CooksOut cooks_gene(const int32_t* y, const double* sf, const double* mu, const double* H,
                           const CellPlan& C, const uint8_t* flags, int N, int P, double cutoff,
                           double* scratch, unsigned int* hist, Sorter&& sorter, double* cooks_out) {
    const double ar = robust_disp_gene<Wv, BIG>(y, sf, C, N, scratch, hist, sorter);
    CooksAcc<Wv> acc(ar, cutoff, P);
    for (int n = Wv::lane(); n < N; n += Wv::W) {
        const double ck = acc.add(n, (double)y[n], mu[n], H[n], flags[n]);
        if (cooks_out != nullptr) cooks_out[n] = ck;
    }
    return acc.finish(y, N);
}

// trimmed mean (trim 0.2) of the normalised counts over all samples (dds.py:1332-1340)
// bw != nullptr (and N >= kTrimBucketMin): the one-pass bucket sum instead of the sort (bucket_rank_sum; the zero counts
// are the smallest values and add nothing)
template <class Wv, class Sorter>
DSQ_HD double trimmed_base_mean(const int32_t* y, const double* sf, int N, double trim,
                                double* scratch, Sorter&& sorter, BucketWork* bw = nullptr) {
    const int nt = (int)floor((double)N * trim);
    if (bw != nullptr && N >= kTrimBucketMin) {
        int zeros = 0;
        for (int k = Wv::lane(); k < N; k += Wv::W) {
            const int yi = y[k];
            scratch[k] = yi == 0 ? -1.0 : (double)yi / sf[k];
            zeros += yi == 0 ? 1 : 0;
        }
        zeros = Wv::sumi(zeros);
        Wv::sync();
        double s;
        if (bucket_rank_sum<Wv>(scratch, N, N - zeros, nt > zeros ? nt - zeros : 0, N - nt - 1 - zeros, *bw, s))
            return s / (double)(N - 2 * nt);
    }
    for (int k = Wv::lane(); k < N; k += Wv::W) scratch[k] = (double)y[k] / sf[k];
    sorter(scratch, N);
    return range_sum<Wv>(scratch, nt, N - nt) / (double)(N - 2 * nt);
}

// The same without a buffer of the row's values (rows beyond what a wavefront's LDS holds: k_replace_lean): the
// normalised counts are recomputed on every pass - by IEEE division, as above, because the result is truncated to an
// integer count - and the trimmed sum comes from the bucket pass or, where that is not applicable, from the radix
// selection over the same accessor.  failed: a normalised count was not finite (the result is NaN then).
struct ExactNormedValues {
    const int32_t* y;
    const double* sf;
    DSQ_HD double operator[](int k) const {
        const int yi = y[k];
        return yi == 0 ? -1.0 : (double)yi / sf[k];
    }
};
template <class Wv>
DSQ_HD double trimmed_base_mean_lean(const int32_t* y, const double* sf, int N, double trim, BucketWork& W, bool& failed) {
    const int nt = (int)floor((double)N * trim);
    const ExactNormedValues V{y, sf};
    int zeros = 0, bad = 0;
    double lo = INFINITY, hi = -INFINITY;
    for_each_batched<Wv>(V, N, [&](double v) {
        zeros += v < 0.0 ? 1 : 0;
        if (!(v < 0.0)) {
            bad |= (v >= 0.0 && v < INFINITY) ? 0 : 1;
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    });
    zeros = Wv::sumi(zeros);
    bad = Wv::sumi(bad);
    failed = bad != 0;
    if (failed) return NAN;
    const double range[2] = {-Wv::max(-lo), Wv::max(hi)};
    const int n_act = N - zeros, j_lo = nt > zeros ? nt - zeros : 0, j_hi = N - nt - 1 - zeros;
    double s = 0.0;
    if (N < kTrimBucketMin || !bucket_rank_sum<Wv>(V, N, n_act, j_lo, j_hi, W, s, range))
        s = select_rank_sum<Wv>(V, N, n_act, j_lo, j_hi, (unsigned int*)W.sum, range);
    return s / (double)(N - 2 * nt);
}

}  // namespace dsq
