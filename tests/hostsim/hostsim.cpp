// hostsim.cpp — TEST-ONLY host instantiation of the per-gene device templates.
//
// The per-gene math of the HIP engine (pydeseq2_amd/csrc/dsq_*.h) is written against a
// Wave policy.  Here it is instantiated with HostWave (one lane walks all samples) and
// compiled with g++, so the exact same source that runs on gfx950 can be unit-tested
// against the oracle in a container that has no GPU.  This library is built by
// tests/hostsim/build.py into tests/hostsim/_hostsim.so, is loaded only by tests, and is
// never imported by the pydeseq2_amd package (whose ops fail loudly without the HIP .so).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "dsq_alpha.h"
#include "dsq_bfgs.h"
#include "dsq_dispatch.h"
#include "dsq_irls.h"
#include "dsq_lbfgsb.h"
#include "dsq_lbfgsb_dense.h"
#include "dsq_stats.h"
#include "dsq_shrink.h"
#include "dsq_trend.h"
#include "dsq_wide.h"

using namespace dsq;

namespace {
struct HostSorter {
    int operator()(double* buf, int n) const {
        std::sort(buf, buf + n);
        return n;
    }
    void merge(double* buf, int n) const { std::sort(buf, buf + n); }
};
}  // namespace

// What the wave-private workspaces (LDS on the device: whatever the previous workgroup left there) hold when a routine
// starts.  0 was what every entry point here used - and what hid, on the host, a read of an unwritten L-BFGS-B work entry that
// made a rescued gene's iterates vary from launch to launch on the device (round 6): the tests now run the routines on
// workspaces filled with 0x00, 0xFF (NaNs) and 0x5A and ask for identical results.
static int g_hs_fill = 0;

extern "C" {

void hs_set_workspace_fill(int byte) { g_hs_fill = byte & 0xFF; }

void hs_lgamma_digamma(const double* x, int n, double* lg, double* dg) {
    for (int i = 0; i < n; ++i) lgamma_digamma<true>(x[i], lg[i], dg[i]);
}

void hs_flog(const double* x, int n, double* lg, double* l1p, double* rc) {
    for (int i = 0; i < n; ++i) { lg[i] = flog(x[i]); l1p[i] = flog1p(x[i]); rc[i] = frcp(x[i]); }
}

void hs_fexp(const double* x, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = fexp_t(x[i]);
}

void hs_norm_sf(const double* x, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = norm_sf(x[i]);
}

typedef void (*fg_cb)(double x, double* f, double* g);
void hs_lbfgsb1d(fg_cb cb, double x0, double l, double u, double* x, double* f, int* success,
                 int* nfev, int* nit, int* status) {
    auto fg = [&](double xx, double& ff, double& gg) { cb(xx, &ff, &gg); };
    Lbfgsb1dResult r = lbfgsb_1d(fg, x0, l, u);
    *x = r.x; *f = r.f; *success = r.success; *nfev = r.nfev; *nit = r.nit; *status = r.status;
}

// gene-major inputs: y[G][ldn] int32, mu[G][ldn]; Xt[P][ldx]
int hs_alpha_mle(const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx, int N,
                 int G, int P_, const double* alpha_hat, double min_disp, double max_disp,
                 double prior_var, int cr_reg, int prior_reg, double* alpha, uint8_t* conv,
                 int32_t* nfev) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    Lbfgsb1d mach;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        AlphaOut o = fit_alpha_gene<HostWave, P, true>(y + (size_t)g * ldn, mu + (size_t)g * ldn, Xt, ldx,
                                                 N, alpha_hat[g], min_disp, max_disp, prior_var,
                                                 cr_reg != 0, prior_reg != 0, mach);
        alpha[g] = o.alpha; conv[g] = (uint8_t)o.converged;
        if (nfev) nfev[g] = o.nfev;
    })
    return 0;
}

// optimizer="BFGS" variants (utils.py:546-554, 389-399)
int hs_alpha_mle_bfgs(const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx, int N,
                      int G, int P_, const double* alpha_hat, double min_disp, double max_disp,
                      double prior_var, int cr_reg, int prior_reg, double* alpha, uint8_t* conv, int32_t* nfev) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        AlphaOut o = fit_alpha_gene_bfgs<HostWave, P, true>(y + (size_t)g * ldn, mu + (size_t)g * ldn, Xt, ldx, N,
                                                            alpha_hat[g], min_disp, max_disp, prior_var, cr_reg != 0,
                                                            prior_reg != 0, 1);
        alpha[g] = o.alpha; conv[g] = (uint8_t)o.converged;
        if (nfev) nfev[g] = o.nfev;
    })
    return 0;
}

int hs_irls_bfgs(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt,
                 int ldx, int N, int G, int P_, const double* disp, double min_mu, double beta_tol,
                 double min_beta, double max_beta, int maxiter, int full_rank, double* beta, double* mu, double* H,
                 uint8_t* conv, int32_t* iters, uint8_t* fallback) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        IrlsArgs A;
        A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
        A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta;
        A.max_beta = max_beta; A.maxiter = maxiter; A.full_rank = full_rank != 0;
        double b[P];
        double* mo = mu ? mu + (size_t)g * ldn : nullptr;
        double* ho = H ? H + (size_t)g * ldn : nullptr;
        IrlsOut o = irls_gene<HostWave, P>(A, b, mo, ho);
        if (o.fallback) {
            static IrlsRescueWork<P> Wk;
            std::memset(&Wk, g_hs_fill, sizeof(Wk));
            o = irls_rescue_gene<HostWave, P>(A, Wk, b, mo, ho, nullptr, 1);
        }
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)o.converged;
        if (iters) iters[g] = o.iters;
        if (fallback) fallback[g] = (uint8_t)o.fallback;
    })
    return 0;
}

int hs_grid_alpha(const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx, int N,
                  int G, int P_, double min_disp, double max_disp, double* log_alpha) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        AlphaArgs A;
        A.y = y + (size_t)g * ldn; A.mu = mu + (size_t)g * ldn; A.Xt = Xt; A.ldx = ldx; A.N = N;
        A.la_hat = 0; A.prior_var = 1;
        A.cst = alpha_const<HostWave>(A.y, A.mu, N);
        log_alpha[g] = grid_fit_alpha<HostWave, P>(A, log(min_disp), log(max_disp));
    })
    return 0;
}

int hs_irls(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt,
            int ldx, int N, int G, int P_, const double* disp, double min_mu, double beta_tol,
            double min_beta, double max_beta, int maxiter, int full_rank, double* beta /*[G][P]*/,
            double* mu /*[G][ldn]*/, double* H /*[G][ldn]*/, uint8_t* conv, int32_t* iters,
            uint8_t* fallback) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        IrlsArgs A;
        A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
        A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta;
        A.max_beta = max_beta; A.maxiter = maxiter; A.full_rank = full_rank != 0;
        double b[P];
        double* mo = mu ? mu + (size_t)g * ldn : nullptr;
        double* ho = H ? H + (size_t)g * ldn : nullptr;
        IrlsOut o = irls_gene<HostWave, P>(A, b, mo, ho);
        if (o.fallback) {
            static IrlsRescueWork<P> Wk;
            std::memset(&Wk, g_hs_fill, sizeof(Wk));
            o = irls_rescue_gene<HostWave, P>(A, Wk, b, mo, ho);
        }
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)o.converged;
        if (iters) iters[g] = o.iters;
        if (fallback) fallback[g] = (uint8_t)o.fallback;
    })
    return 0;
}

// dispersion fit through the cell path (designs with few distinct rows)
int hs_alpha_mle_cell(const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx, int N, int G, int P_,
                      const double* alpha_hat, double min_disp, double max_disp, double prior_var, int cr_reg,
                      int prior_reg, const int32_t* cell_of, const double* Xc, const double* XX, int C,
                      double* alpha, uint8_t* conv) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    Lbfgsb1d mach;
    CellDesign D{cell_of, Xc, XX, C};
    DSQ_DISPATCH_P(P_, {
        static CellWork<P> Wk;
        std::memset(&Wk, g_hs_fill, sizeof(Wk));
        CellCtx cctx{D, (void*)&Wk};
        for (int g = 0; g < G; ++g) {
            AlphaOut o = fit_alpha_gene<HostWave, P, false, false, true>(
                y + (size_t)g * ldn, mu + (size_t)g * ldn, Xt, ldx, N, alpha_hat[g], min_disp, max_disp, prior_var,
                cr_reg != 0, prior_reg != 0, mach, nullptr, nullptr, 1, &cctx);
            alpha[g] = o.alpha; conv[g] = (uint8_t)o.converged;
        }
    })
    return 0;
}

// LFC fit with the fused epilogue (Cook's bookkeeping + Wald), general (C == 0) or cell path
int hs_lfc_fit(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt, int ldx, int N,
               int G, int P_, const double* disp, double min_mu, double beta_tol, int full_rank,
               const int32_t* cell_of, const double* Xc, const double* XX, int C,
               const double* robust_disp, const uint8_t* flags, double cutoff, double* cooks,
               uint8_t* any_all, uint8_t* any_use, uint8_t* any_use_nr, uint8_t* few_above,
               const double* ridge, const double* contrast, double lfc_null, int alt,
               double* beta, double* mu, double* H, uint8_t* conv, double* pv, double* st, double* se) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    CellDesign D{cell_of, Xc, XX, C};
    DSQ_DISPATCH_P(P_, {
        static CellWork<P> Wk;
        std::memset(&Wk, g_hs_fill, sizeof(Wk));
        for (int g = 0; g < G; ++g) {
            IrlsArgs A;
            A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
            A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = -30.0; A.max_beta = 30.0;
            A.maxiter = 250; A.full_rank = full_rank != 0;
            if (C > 0) { A.cells = &D; A.cell_ws = (void*)&Wk; }
            LfcEpilogue E;
            if (flags != nullptr) {
                E.flags = flags; E.robust_disp = robust_disp[g]; E.cutoff = cutoff;
                E.cooks_row = cooks ? cooks + (size_t)g * ldn : nullptr;
            }
            if (ridge != nullptr) { E.ridge = ridge; E.contrast = contrast; E.lfc_null = lfc_null; E.alt = alt; }
            double b[P];
            double* mo = mu ? mu + (size_t)g * ldn : nullptr;
            double* ho = H ? H + (size_t)g * ldn : nullptr;
            IrlsOut o;
            if (C > kSmallCells) o = irls_gene<HostWave, P, 1>(A, b, mo, ho, &E);
            else if (C > 0) {
                if constexpr (P <= 2) {
                    if (C <= 2) o = irls_gene<HostWave, P, 3>(A, b, mo, ho, &E);
                    else o = irls_gene<HostWave, P, 2>(A, b, mo, ho, &E);
                } else if constexpr (P <= 4) o = irls_gene<HostWave, P, 2>(A, b, mo, ho, &E);
                else return -2;
            } else o = irls_gene<HostWave, P, 0>(A, b, mo, ho, &E);
            if (o.fallback) {
                static IrlsRescueWork<P> Rk;
                std::memset(&Rk, g_hs_fill, sizeof(Rk));
                o = irls_rescue_gene<HostWave, P>(A, Rk, b, mo, ho, &E);
            }
            for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
            conv[g] = (uint8_t)o.converged;
            if (flags != nullptr) {
                any_all[g] = E.cooks.any_gt_all; any_use[g] = E.cooks.any_gt_use; any_use_nr[g] = E.cooks.any_gt_use_nr;
                few_above[g] = E.cooks.few_above;
            }
            if (ridge != nullptr) { pv[g] = E.wald.p; st[g] = E.wald.stat; se[g] = E.wald.se; }
        }
    })
    return 0;
}

// ---- run-time-P ("wide") path: any number of design columns up to kWideMaxP
int hs_alpha_mle_wide(const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx, int N, int G, int P_,
                      const double* alpha_hat, double min_disp, double max_disp, double prior_var, int cr_reg,
                      int prior_reg, const int32_t* cell_of, const double* Xc, const double* XX, int C,
                      double* alpha, uint8_t* conv) {
    if (P_ < 1 || P_ > kWideMaxP) return -1;
    std::vector<double> buf((size_t)wide_work_doubles(P_));
    WideWork W;
    W.bind(buf.data(), P_);
    Lbfgsb1d mach;
    CellDesign D{cell_of, Xc, XX, C};
    for (int g = 0; g < G; ++g) {
        AlphaOut o = fit_alpha_wide<HostWave>(y + (size_t)g * ldn, mu + (size_t)g * ldn, Xt, ldx, N, W,
                                              C > 0 ? &D : nullptr, alpha_hat[g], min_disp, max_disp, prior_var,
                                              cr_reg != 0, prior_reg != 0, mach, nullptr, nullptr);
        alpha[g] = o.alpha; conv[g] = (uint8_t)o.converged;
    }
    return 0;
}

int hs_lfc_fit_wide(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt, int ldx, int N,
                    int G, int P_, const double* disp, double min_mu, double beta_tol, int full_rank,
                    const int32_t* cell_of, const double* Xc, const double* XX, int C,
                    const double* robust_disp, const uint8_t* flags, double cutoff, double* cooks,
                    uint8_t* any_all, uint8_t* any_use, uint8_t* any_use_nr, uint8_t* few_above,
                    const double* ridge, const double* contrast, double lfc_null, int alt,
                    double* beta, double* mu, double* H, uint8_t* conv, double* pv, double* st, double* se) {
    if (P_ < 1 || P_ > kWideMaxP) return -1;
    std::vector<double> buf((size_t)wide_work_doubles(P_)), xlu(3 * kWideMaxP);
    std::vector<int> nbd(kWideMaxP);
    static LbfgsbWork<kWideMaxP> Lb;
    WideWork W;
    W.bind(buf.data(), P_);
    CellDesign D{cell_of, Xc, XX, C};
    for (int g = 0; g < G; ++g) {
        IrlsArgs A;
        A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
        A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = -30.0; A.max_beta = 30.0;
        A.maxiter = 250; A.full_rank = full_rank != 0;
        if (C > 0) A.cells = &D;
        LfcEpilogue E;
        if (flags != nullptr) {
            E.flags = flags; E.robust_disp = robust_disp[g]; E.cutoff = cutoff;
            E.cooks_row = cooks ? cooks + (size_t)g * ldn : nullptr;
        }
        if (ridge != nullptr) { E.ridge = ridge; E.contrast = contrast; E.lfc_null = lfc_null; E.alt = alt; }
        double* mo = mu ? mu + (size_t)g * ldn : nullptr;
        double* ho = H ? H + (size_t)g * ldn : nullptr;
        IrlsOut o = irls_gene_wide<HostWave>(A, W, mo, ho, &E);
        if (o.fallback) {
            std::memset(&Lb, g_hs_fill, sizeof(Lb));
            o = irls_rescue_wide<HostWave>(A, W, Lb, xlu.data(), nbd.data(), mo, ho, &E);
        }
        for (int j = 0; j < P_; ++j) beta[(size_t)g * P_ + j] = W.v(0)[j];
        conv[g] = (uint8_t)o.converged;
        if (flags != nullptr) {
            any_all[g] = E.cooks.any_gt_all; any_use[g] = E.cooks.any_gt_use; any_use_nr[g] = E.cooks.any_gt_use_nr;
            few_above[g] = E.cooks.few_above;
        }
        if (ridge != nullptr) { pv[g] = E.wald.p; st[g] = E.wald.stat; se[g] = E.wald.se; }
    }
    return 0;
}

int hs_mom_wide(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt, int ldx, int N,
                int G, int P_, double min_disp, double max_disp, double min_mu, double* normed_mean, double* rough,
                double* moments, double* mom, double* mu) {
    if (P_ < 1 || P_ > kWideMaxP) return -1;
    std::vector<double> buf((size_t)wide_work_doubles(P_));
    WideWork W;
    W.bind(buf.data(), P_);
    double smi = 0.0;
    for (int n = 0; n < N; ++n) smi += 1.0 / sf[n];
    smi /= N;
    for (int g = 0; g < G; ++g) {
        MomOut o = mom_wide<HostWave>(y + (size_t)g * ldn, sf, Xt, pinvXt, ldx, N, W, smi, min_disp, max_disp, min_mu,
                                      mu ? mu + (size_t)g * ldn : nullptr);
        normed_mean[g] = o.normed_mean; rough[g] = o.rough; moments[g] = o.moments; mom[g] = o.mom;
    }
    return 0;
}

int hs_shrink(const int32_t* y, int ldn, const double* offset, const double* Xt, int ldx, int N, int G, int P_,
              const double* size, double sigma0, double sigma, int shrink_index, double* beta /*[G][P]*/,
              double* invh /*[G][P][P]*/, uint8_t* conv, int optimizer) {
    if (P_ < 1 || P_ > 48) return -1;
    if (optimizer != 0) {  // "BFGS" / "Newton-CG" (k_shrink<P, true> on the device)
        if (P_ > DSQ_REG_MAX_P) return -1;
        DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
            ShrinkArgs A;
            A.y = y + (size_t)g * ldn; A.offset = offset; A.Xt = Xt; A.ldx = ldx; A.N = N;
            A.size = size[g]; A.sigma0 = sigma0; A.sigma = sigma; A.shrink_index = shrink_index;
            static ShrinkWorkAlt<P> Wa;
            std::memset(&Wa, g_hs_fill, sizeof(Wa));
            double b[P];
            conv[g] = (uint8_t)shrink_gene<HostWave, P>(A, Wa, b, invh + (size_t)g * P * P, nullptr, optimizer);
            for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        })
        return 0;
    }
    if (P_ > 32) {  // 33 ... 48 columns (round 6: k_shrink_wide<48, PB, true>)
        static ShrinkWorkWide<48> Ww;
        for (int g = 0; g < G; ++g) {
            ShrinkArgs A;
            A.y = y + (size_t)g * ldn; A.offset = offset; A.Xt = Xt; A.ldx = ldx; A.N = N;
            A.size = size[g]; A.sigma0 = sigma0; A.sigma = sigma; A.shrink_index = shrink_index;
            std::memset(&Ww, g_hs_fill, sizeof(Ww));
            if (P_ <= 40)
                conv[g] = (uint8_t)shrink_gene_wide<HostWave, 48, decltype(Ww), 40>(A, P_, Ww, beta + (size_t)g * P_, invh + (size_t)g * P_ * P_);
            else
                conv[g] = (uint8_t)shrink_gene_wide<HostWave, 48>(A, P_, Ww, beta + (size_t)g * P_, invh + (size_t)g * P_ * P_);
        }
        return 0;
    }
    if (P_ > DSQ_REG_MAX_P) {  // 13 ... 32 columns: the run-time-p templates (k_shrink_wide on the device)
        static ShrinkWorkWide<32> Ww;
        for (int g = 0; g < G; ++g) {
            ShrinkArgs A;
            A.y = y + (size_t)g * ldn; A.offset = offset; A.Xt = Xt; A.ldx = ldx; A.N = N;
            A.size = size[g]; A.sigma0 = sigma0; A.sigma = sigma; A.shrink_index = shrink_index;
            std::memset(&Ww, g_hs_fill, sizeof(Ww));
            conv[g] = (uint8_t)shrink_gene_wide<HostWave, 32>(A, P_, Ww, beta + (size_t)g * P_, invh + (size_t)g * P_ * P_);
        }
        return 0;
    }
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        ShrinkArgs A;
        A.y = y + (size_t)g * ldn; A.offset = offset; A.Xt = Xt; A.ldx = ldx; A.N = N;
        A.size = size[g]; A.sigma0 = sigma0; A.sigma = sigma; A.shrink_index = shrink_index;
        static ShrinkWork<P> Wk;
        std::memset(&Wk, g_hs_fill, sizeof(Wk));
        double b[P];
        conv[g] = (uint8_t)shrink_gene<HostWave, P>(A, Wk, b, invh + (size_t)g * P * P);
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
    })
    return 0;
}

int hs_logmeans(const int32_t* y, int ldn, int N, int G, double* logmeans, uint8_t* nonzero) {
    for (int g = 0; g < G; ++g) {
        int nz;
        gene_logmean<HostWave>(y + (size_t)g * ldn, N, logmeans[g], nz);
        nonzero[g] = (uint8_t)nz;
    }
    return 0;
}

int hs_mom(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt,
           int ldx, int N, int G, int P_, double min_disp, double max_disp, double* normed_mean,
           double* rough, double* moments, double* mom) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    double smi = 0;
    for (int n = 0; n < N; ++n) smi += 1.0 / sf[n];
    smi /= N;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        MomOut o = mom_gene<HostWave, P>(y + (size_t)g * ldn, sf, Xt, pinvXt, ldx, N, smi, min_disp,
                                         max_disp);
        normed_mean[g] = o.normed_mean; rough[g] = o.rough; moments[g] = o.moments; mom[g] = o.mom;
    })
    return 0;
}

int hs_lin_mu(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt,
              int ldx, int N, int G, int P_, double min_mu, double* mu) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g)
        lin_mu_gene<HostWave, P>(y + (size_t)g * ldn, sf, Xt, pinvXt, ldx, N, min_mu,
                                 mu + (size_t)g * ldn);)
    return 0;
}

int hs_mom_lin_mu(const int32_t* y, int ldn, const double* sf, const double* Xt, const double* pinvXt, int ldx,
                  int N, int G, int P_, double min_disp, double max_disp, double min_mu, double* normed_mean,
                  double* mom, double* mu) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    double smi = 0;
    for (int n = 0; n < N; ++n) smi += 1.0 / sf[n];
    smi /= N;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        MomOut o = mom_lin_mu_gene<HostWave, P>(y + (size_t)g * ldn, sf, Xt, pinvXt, ldx, N, smi, min_disp,
                                                max_disp, min_mu, mu + (size_t)g * ldn);
        normed_mean[g] = o.normed_mean; mom[g] = o.mom;
    })
    return 0;
}

int hs_wald(const double* mu, int ldn, const double* sf, const double* Xt, int ldx, int N, int G,
            int P_, const double* disp, const double* beta, const double* ridge,
            const double* contrast, double lfc_null, int alt, double* pval, double* stat,
            double* se) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    DSQ_DISPATCH_P(P_, for (int g = 0; g < G; ++g) {
        double b[P];
        for (int j = 0; j < P; ++j) b[j] = beta[(size_t)g * P + j];
        WaldOut o = wald_gene<HostWave, P>(mu ? mu + (size_t)g * ldn : nullptr, sf, Xt, ldx, N,
                                           disp[g], b, ridge, contrast, lfc_null, alt);
        pval[g] = o.p; stat[g] = o.stat; se[g] = o.se;
    })
    return 0;
}

double hs_trimmed_sum(const double* buf, int n, int nt) {
    std::vector<unsigned int> hist(2 * kTrimBins);
    return trimmed_sum_select<HostWave>(buf, n, nt, hist.data());
}

int hs_cooks(const int32_t* y, int ldn, const double* sf, const double* mu, const double* H,
             const int32_t* cell_offsets, const int32_t* cell_index, int n_cells, int whole,
             const uint8_t* flags, int N, int G, int P_, double cutoff, double* cooks,
             double* robust_disp, uint8_t* any_all, uint8_t* any_use, uint8_t* any_use_nr,
             uint8_t* few_above) {
    std::vector<double> scratch(N + 8);
    std::vector<unsigned int> hist(sizeof(BucketWork) / sizeof(unsigned int) + 2);  // (robust_disp_gene: a BucketWork)
    CellPlan C{cell_offsets, cell_index, n_cells, whole};
    for (int g = 0; g < G; ++g) {
        CooksOut o = cooks_gene<HostWave>(y + (size_t)g * ldn, sf, mu + (size_t)g * ldn,
                                          H + (size_t)g * ldn, C, flags, N, P_, cutoff,
                                          scratch.data(), hist.data(), HostSorter(),
                                          cooks ? cooks + (size_t)g * ldn : nullptr);
        robust_disp[g] = o.robust_disp; any_all[g] = o.any_gt_all; any_use[g] = o.any_gt_use;
        any_use_nr[g] = o.any_gt_use_nr; few_above[g] = o.few_above;
    }
    return 0;
}

// robust dispersions with the small cells sorted several per pass (seg_len > 0: k_robust_disp<.., false>'s path)
int hs_robust_disp_seg(const int32_t* y, int ldn, const double* sf, const int32_t* cell_offsets,
                       const int32_t* cell_index, int n_cells, int whole, int N, int G, int seg_len, double* out) {
    std::vector<double> scratch((size_t)(N + 8 > 256 ? N + 8 : 256));
    CellPlan C{cell_offsets, cell_index, n_cells, whole};
    for (int g = 0; g < G; ++g)
        out[g] = robust_disp_gene<HostWave, false>(y + (size_t)g * ldn, sf, C, N, scratch.data(), nullptr, HostSorter(),
                                                   seg_len);
    return 0;
}

int hs_trimmed_base_mean(const int32_t* y, int ldn, const double* sf, int N, int G, double trim,
                         double* out) {
    std::vector<double> scratch(N + 8);
    std::vector<BucketWork> bw(1);  // (as k_replace: the bucket path from kTrimBucketMin samples on)
    for (int g = 0; g < G; ++g)
        out[g] = trimmed_base_mean<HostWave>(y + (size_t)g * ldn, sf, N, trim, scratch.data(),
                                             HostSorter(), bw.data());
    return 0;
}

// the buffer-less routines of cells / rows of any length (k_robust_disp_lean, k_replace_lean)
int hs_robust_disp_lean(const int32_t* y, int ldn, const double* sf, const int32_t* cell_offsets,
                        const int32_t* cell_index, int n_cells, int whole, int N, int G, double* out, uint8_t* failed) {
    std::vector<BucketWork> bw(1);
    CellPlan C{cell_offsets, cell_index, n_cells, whole};
    for (int g = 0; g < G; ++g) {
        bool f = false;
        out[g] = robust_disp_gene_lean<HostWave>(y + (size_t)g * ldn, sf, C, N, bw[0], f);
        failed[g] = f ? 1 : 0;
    }
    return 0;
}

int hs_trimmed_base_mean_lean(const int32_t* y, int ldn, const double* sf, int N, int G, double trim, double* out) {
    std::vector<BucketWork> bw(1);
    for (int g = 0; g < G; ++g) {
        bool f = false;
        out[g] = trimmed_base_mean_lean<HostWave>(y + (size_t)g * ldn, sf, N, trim, bw[0], f);
    }
    return 0;
}

double hs_select_rank_sum(const double* buf, int n, int j_lo, int j_hi) {
    std::vector<unsigned int> hist(2 * kTrimBins);
    int n_act = 0;
    double range[2] = {INFINITY, -INFINITY};
    for (int k = 0; k < n; ++k)
        if (buf[k] >= 0.0) { ++n_act; range[0] = buf[k] < range[0] ? buf[k] : range[0]; range[1] = buf[k] > range[1] ? buf[k] : range[1]; }
    return select_rank_sum<HostWave>(buf, n, n_act, j_lo, j_hi, hist.data(), range);
}

// loss/gradient of one gene at log_alpha (debug / unit tests)
int hs_alpha_eval(const int32_t* y, const double* mu, const double* Xt, int ldx, int N, int P_,
                  double la, double la_hat, double prior_var, int cr_reg, int prior_reg, double* f,
                  double* g) {
    if (P_ < 1 || P_ > DSQ_REG_MAX_P) return -1;
    DSQ_DISPATCH_P(P_, {
        AlphaArgs A;
        A.y = y; A.mu = mu; A.Xt = Xt; A.ldx = ldx; A.N = N; A.la_hat = la_hat;
        A.prior_var = prior_var;
        A.cst = alpha_const<HostWave>(y, mu, N);
        alpha_eval<HostWave, P, true>(A, la, cr_reg != 0, prior_reg != 0, *f, *g);
    })
    return 0;
}

typedef void (*fgn_cb)(const double* x, double* f, double* g);
int hs_lbfgsb_nd(fgn_cb cb, int n, double* x, const double* l, const double* u, const int* nbd,
                 double* f, int* success, int* nfev, int* nit, int* status) {
    if (n < 1 || n > 16) return -1;
    static LbfgsbWork<16> W;
    std::memset(&W, g_hs_fill, sizeof(W));
    auto fg = [&](const double* xx, double& ff, double* gg) { cb(xx, &ff, gg); };
    LbfgsbResult r = lbfgsb_nd<16>(fg, n, x, l, u, nbd, W);
    *f = r.f; *success = r.success; *nfev = r.nfev; *nit = r.nit; *status = r.status;
    return 0;
}

int hs_trend_fit(const double* disp, const double* means, int n, double min_disp, double max_disp,
                 double* coeffs, int* ok, int* n_outer) {
    std::vector<uint8_t> keep(n + 1);
    static TrendWork W;
    std::memset(&W, g_hs_fill, sizeof(W));
    TrendOut o = trend_fit<HostWave>(disp, means, n, min_disp, max_disp, keep.data(), W);
    coeffs[0] = o.a0; coeffs[1] = o.a1; *ok = o.ok; *n_outer = o.n_outer;
    return 0;
}

int hs_bfgs(fgn_cb cb, int n, double* x, int* success, int* nfev, int* nit, int* status) {
    if (n < 1 || n > 16) return -1;
    static BfgsWork<16> W;
    std::memset(&W, g_hs_fill, sizeof(W));
    auto fg = [&](const double* xx, double& ff, double* gg) { cb(xx, &ff, gg); };
    const BfgsResult r = bfgs_min<16>(fg, n, x, W);
    *success = r.success; *nfev = r.nfev; *nit = r.nit; *status = r.status;
    return 0;
}

int hs_lbfgsb_dense(fgn_cb cb, int n, double* x, const double* l, const double* u, const int* nbd,
                    double* f, int* success, int* nfev, int* nit, int* status) {
    if (n < 1 || n > 4) return -1;
    static LbfgsbDenseWork<4> W;
    std::memset(&W, g_hs_fill, sizeof(W));
    auto fg = [&](const double* xx, double& ff, double* gg) { cb(xx, &ff, gg); };
    LbfgsbResult r = lbfgsb_dense<4>(fg, n, x, l, u, nbd, W);
    *f = r.f; *success = r.success; *nfev = r.nfev; *nit = r.nit; *status = r.status;
    return 0;
}

}  // extern "C"
