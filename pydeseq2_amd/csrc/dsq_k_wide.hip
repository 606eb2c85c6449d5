// dsq_k_wide.hip — kernels of the run-time-P path (dsq_wide.h): designs wider than the register path's 12 columns
// (up to kWideMaxP = 48), and — optionally — narrower designs without cell structure, whose p(p+1) register
// accumulators spill.  One gene per wavefront; the wave's p x p matrices, the staged design chunk and the small
// vectors live in a wave-private segment of dynamic LDS whose size follows P (1, 2 or 4 waves per workgroup);
// X^T W X is accumulated by v_mfma_f64_16x16x4_f64 (or from per-cell sums for cell designs).
#include "dsq_dispatch.h"
#include "dsq_launch.h"
#include "dsq_wide.h"

namespace dsq {

namespace {

struct WideGeom {
    int wpb;        // waves (genes) per workgroup
    size_t per_wave;  // bytes of LDS per wave
};
WideGeom wide_geom(int P, size_t extra_doubles = 0) {
    WideGeom g;
    g.per_wave = ((size_t)wide_work_doubles(P) + extra_doubles) * sizeof(double);
    g.wpb = g.per_wave * 4 <= 64 * 1024 ? 4 : (g.per_wave * 2 <= 64 * 1024 ? 2 : 1);
    return g;
}

template <class K>
void set_smem(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        (void)hipGetLastError();
    }
}

constexpr int kWideMaxWaves = 4;

}  // namespace

#define DSQ_WIDE_PROLOGUE(G_)                                                     \
    extern __shared__ __attribute__((aligned(16))) double wide_lds[];             \
    const int wv = threadIdx.x >> 6;                                              \
    const int g = blockIdx.x * (blockDim.x >> 6) + wv;                            \
    if (g >= (G_)) return;                                                        \
    WideWork W;                                                                   \
    W.bind(wide_lds + (size_t)wv * per_wave_doubles, P)

// ------------------------------------------------------------------ MoM (+ linear-model mu_hat, OLS coefficients)
__global__ __launch_bounds__(256) void k_mom_wide(const int32_t* __restrict__ y, int ldn, const double* __restrict__ sf,
                                                  const double* __restrict__ Xt, const double* __restrict__ pinvXt,
                                                  int ldx, int N, int G, int P, int per_wave_doubles,
                                                  const double* __restrict__ s_mean_inv, double min_disp,
                                                  double max_disp, double min_mu, double* __restrict__ normed_mean,
                                                  double* __restrict__ rough, double* __restrict__ moments,
                                                  double* __restrict__ mom, double* __restrict__ mu,
                                                  double* __restrict__ coef) {
    DSQ_WIDE_PROLOGUE(G);
    const MomOut o = mom_wide<DeviceWave>(y + (size_t)g * ldn, sf, Xt, pinvXt, ldx, N, W, s_mean_inv[0], min_disp,
                                          max_disp, min_mu, mu ? mu + (size_t)g * ldn : nullptr);
    if ((threadIdx.x & 63) == 0) {
        if (normed_mean) normed_mean[g] = o.normed_mean;
        if (rough) rough[g] = o.rough;
        if (moments) moments[g] = o.moments;
        if (mom) mom[g] = o.mom;
    }
    if (coef != nullptr)
        for (int j = threadIdx.x & 63; j < P; j += 64) coef[(size_t)g * P + j] = W.v(0)[j];
}

hipError_t launch_wide_mom(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                           const double* pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                           double min_mu, double* normed_mean, double* rough, double* moments, double* mom,
                           double* mu, double* coef, const double* d_s_mean_inv) {
    if (G <= 0) return hipSuccess;
    const WideGeom ge = wide_geom(P);
    set_smem(k_mom_wide, ge.per_wave * ge.wpb);
    hipLaunchKernelGGL(k_mom_wide, dim3((G + ge.wpb - 1) / ge.wpb), dim3(64 * ge.wpb), ge.per_wave * ge.wpb, st, y,
                       ldn, sf, Xt, pinvXt, ldx, N, G, P, (int)(ge.per_wave / 8), d_s_mean_inv, min_disp, max_disp,
                       min_mu, normed_mean, rough, moments, mom, mu, coef);
    return hipGetLastError();
}

// rough dispersions from already-normalised counts (Inference.fit_rough_dispersions)
__global__ __launch_bounds__(256) void k_rough_normed_wide(const double* __restrict__ normed, int ldn,
                                                           const double* __restrict__ Xt,
                                                           const double* __restrict__ pinvXt, int ldx, int N, int G,
                                                           int P, int per_wave_doubles, double* __restrict__ out) {
    DSQ_WIDE_PROLOGUE(G);
    const double* v = normed + (size_t)g * ldn;
    for (int j = 0; j < P; ++j) {
        double b = 0.0;
        for (int n = DeviceWave::lane(); n < N; n += 64) b += pinvXt[j * ldx + n] * v[n];
        b = DeviceWave::sum(b);
        if ((threadIdx.x & 63) == 0) W.v(0)[j] = b;
    }
    DeviceWave::sync();
    double rr = 0.0;
    const double dof = (double)(N - P);
    for (int n = DeviceWave::lane(); n < N; n += 64) {
        double yh = 0.0;
        for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * W.v(0)[j];
        yh = dmax(yh, 1.0);
        rr += ((v[n] - yh) * (v[n] - yh) - yh) / (dof * yh * yh);
    }
    rr = DeviceWave::sum(rr);
    if ((threadIdx.x & 63) == 0) out[g] = dmax(rr, 0.0);
}

hipError_t launch_wide_rough_normed(hipStream_t st, const double* normed, int ldn, const double* Xt,
                                    const double* pinvXt, int ldx, int N, int G, int P, double* out) {
    if (G <= 0) return hipSuccess;
    const WideGeom ge = wide_geom(P);
    set_smem(k_rough_normed_wide, ge.per_wave * ge.wpb);
    hipLaunchKernelGGL(k_rough_normed_wide, dim3((G + ge.wpb - 1) / ge.wpb), dim3(64 * ge.wpb), ge.per_wave * ge.wpb,
                       st, normed, ldn, Xt, pinvXt, ldx, N, G, P, (int)(ge.per_wave / 8), out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ dispersion fit
__global__ __launch_bounds__(256) void k_alpha_wide(const int32_t* __restrict__ y, const double* __restrict__ mu,
                                                    int ldn, const double* __restrict__ Xt, int ldx, int N, int G,
                                                    int P, int per_wave_doubles, const double* __restrict__ alpha_hat,
                                                    double min_disp, double max_disp, double prior_var, int cr_reg,
                                                    int prior_reg, double* __restrict__ alpha,
                                                    uint8_t* __restrict__ conv, int32_t* __restrict__ nfev,
                                                    double* __restrict__ nll_const, int const_mode, CellDesign cells,
                                                    const int32_t* __restrict__ list) {
    __shared__ Lbfgsb1d machine[kWideMaxWaves];
    extern __shared__ __attribute__((aligned(16))) double wide_lds[];
    log_tab_fill();  // the count memo takes its logarithms through the LDS table (flog_t, dsq_math.h)
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const int k = blockIdx.x * (blockDim.x >> 6) + wv;
    if (k >= G) return;
    const int g = list != nullptr ? list[k] : k;  // list: the genes of a grid-search-only pass
    WideWork W;
    W.bind(wide_lds + (size_t)wv * per_wave_doubles, P);
    const AlphaOut o = fit_alpha_wide<DeviceWave>(
        y + (size_t)g * ldn, mu + (size_t)g * ldn, Xt, ldx, N, W, cells.C > 0 ? &cells : nullptr, alpha_hat[g],
        min_disp, max_disp, prior_var, cr_reg != 0, prior_reg != 0, machine[wv],
        const_mode == DSQ_CONST_LOAD ? nll_const + g : nullptr, const_mode == DSQ_CONST_STORE ? nll_const + g : nullptr);
    if ((threadIdx.x & 63) == 0) {
        alpha[g] = o.alpha;
        conv[g] = (uint8_t)o.converged;
        if (nfev != nullptr) nfev[g] = o.nfev;
    }
}

// grid_fit_alpha alone for listed genes: alpha[list[k]] = exp(best grid point)
__global__ __launch_bounds__(256) void k_alpha_grid_wide(const int32_t* __restrict__ y, const double* __restrict__ mu,
                                                         int ldn, const double* __restrict__ Xt, int ldx, int N,
                                                         int P, int per_wave_doubles, double min_disp, double max_disp,
                                                         double* __restrict__ alpha, const int32_t* __restrict__ list,
                                                         int n_list) {
    extern __shared__ __attribute__((aligned(16))) double wide_lds[];
    log_tab_fill();  // the count memo takes its logarithms through the LDS table (flog_t, dsq_math.h)
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const int k = blockIdx.x * (blockDim.x >> 6) + wv;
    if (k >= n_list) return;
    const int g = list[k];
    WideWork W;
    W.bind(wide_lds + (size_t)wv * per_wave_doubles, P);
    WideAlphaArgs A;
    A.y = y + (size_t)g * ldn; A.mu = mu + (size_t)g * ldn; A.Xt = Xt; A.ldx = ldx; A.N = N; A.cells = nullptr;
    A.la_hat = 0.0; A.prior_var = 1.0;
    A.cst = alpha_const<DeviceWave>(A.y, A.mu, N);
    double lohi[2] = {log(min_disp), log(max_disp)};
    double best_la = 0.0;
    for (int level = 0; level < 2; ++level) {
        double best = 0.0;
        int kbest = 0;
        bool best_nan = false;
        for (int i = 0; i < 100; ++i) {
            double f, gu;
            alpha_eval_wide<DeviceWave, false>(A, W, linspace_at(lohi[0], lohi[1], 100, i), true, false, f, gu);
            const bool isn = (f != f);
            if (i == 0 || (!best_nan && (isn || f < best))) { best = f; kbest = i; best_nan = isn; }
        }
        const double c = linspace_at(lohi[0], lohi[1], 100, kbest);
        const double delta = linspace_at(lohi[0], lohi[1], 100, 1) - linspace_at(lohi[0], lohi[1], 100, 0);
        best_la = c;
        lohi[0] = c - delta; lohi[1] = c + delta;
    }
    if ((threadIdx.x & 63) == 0) alpha[g] = exp(best_la);
}

hipError_t launch_wide_alpha(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx,
                             int N, int G, int P, const double* alpha_hat, double min_disp, double max_disp,
                             double prior_var, int cr_reg, int prior_reg, double* alpha, uint8_t* conv,
                             int32_t* nfev, double* nll_const, int const_mode, const CellDesign* cells) {
    if (G <= 0) return hipSuccess;
    if (nll_const == nullptr) const_mode = DSQ_CONST_COMPUTE;
    CellDesign cd{};
    if (cells != nullptr) cd = *cells;
    const WideGeom ge = wide_geom(P);
    set_smem(k_alpha_wide, ge.per_wave * ge.wpb);
    hipLaunchKernelGGL(k_alpha_wide, dim3((G + ge.wpb - 1) / ge.wpb), dim3(64 * ge.wpb), ge.per_wave * ge.wpb, st, y,
                       mu, ldn, Xt, ldx, N, G, P, (int)(ge.per_wave / 8), alpha_hat, min_disp, max_disp, prior_var,
                       cr_reg, prior_reg, alpha, conv, nfev, nll_const, const_mode, cd, (const int32_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_wide_alpha_grid(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                                  int ldx, int N, int P, double min_disp, double max_disp, double* alpha,
                                  const int32_t* list, int n_list) {
    if (n_list <= 0) return hipSuccess;
    const WideGeom ge = wide_geom(P);
    set_smem(k_alpha_grid_wide, ge.per_wave * ge.wpb);
    hipLaunchKernelGGL(k_alpha_grid_wide, dim3((n_list + ge.wpb - 1) / ge.wpb), dim3(64 * ge.wpb),
                       ge.per_wave * ge.wpb, st, y, mu, ldn, Xt, ldx, N, P, (int)(ge.per_wave / 8), min_disp, max_disp,
                       alpha, list, n_list);
    return hipGetLastError();
}

// ------------------------------------------------------------------ IRLS (+ fused epilogue), rescue, layers, Wald
__device__ __forceinline__ void wide_epilogue_begin(LfcEpilogue& E, const IrlsExtras& ex, int g, int ldn) {
    if (ex.flags != nullptr) {
        E.flags = ex.flags; E.robust_disp = ex.robust_disp[g]; E.cutoff = ex.cutoff;
        E.cooks_row = ex.cooks ? ex.cooks + (size_t)g * ldn : nullptr;
    }
    if (ex.ridge != nullptr) { E.ridge = ex.ridge; E.contrast = ex.contrast; E.lfc_null = ex.lfc_null; E.alt = ex.alt; }
}
__device__ __forceinline__ void wide_epilogue_store(const LfcEpilogue& E, const IrlsExtras& ex, int g) {
    if (ex.flags != nullptr) {
        ex.any_all[g] = (uint8_t)E.cooks.any_gt_all;
        ex.any_use[g] = (uint8_t)E.cooks.any_gt_use;
        ex.any_use_nr[g] = (uint8_t)E.cooks.any_gt_use_nr;
        ex.few_above[g] = (uint8_t)E.cooks.few_above;
    }
    if (ex.ridge != nullptr) { ex.pvals[g] = E.wald.p; ex.stats[g] = E.wald.stat; ex.se[g] = E.wald.se; }
}

__global__ __launch_bounds__(256) void k_irls_wide(const int32_t* __restrict__ y, int ldn, const double* __restrict__ sf,
                                                   const double* __restrict__ lsf, const double* __restrict__ Xt,
                                                   const double* __restrict__ pinvXt, int ldx, int N, int G, int P,
                                                   int per_wave_doubles, int full_rank, const double* __restrict__ disp,
                                                   double min_mu, double beta_tol, double min_beta, double max_beta,
                                                   int maxiter, double* __restrict__ beta, double* __restrict__ mu,
                                                   double* __restrict__ hat, uint8_t* __restrict__ conv,
                                                   int32_t* __restrict__ iters, int32_t* __restrict__ fb_count,
                                                   int32_t* __restrict__ fb_list, IrlsExtras ex) {
    DSQ_WIDE_PROLOGUE(G);
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = lsf; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta; A.max_beta = max_beta;
    A.maxiter = maxiter; A.full_rank = full_rank != 0;
    if (ex.cells.C > 0) A.cells = &ex.cells;
    LfcEpilogue E;
    wide_epilogue_begin(E, ex, g, ldn);
    const IrlsOut o = irls_gene_wide<DeviceWave>(A, W, mu ? mu + (size_t)g * ldn : nullptr,
                                                 hat ? hat + (size_t)g * ldn : nullptr, &E);
    if (!o.fallback)
        for (int j = threadIdx.x & 63; j < P; j += 64) beta[(size_t)g * P + j] = W.v(0)[j];
    if ((threadIdx.x & 63) == 0) {
        conv[g] = (uint8_t)o.converged;
        if (iters != nullptr) iters[g] = o.iters;
        if (o.fallback) fb_list[atomicAdd(fb_count, 1)] = g;
        else wide_epilogue_store(E, ex, g);
    }
}

__global__ __launch_bounds__(64) void k_irls_rescue_wide(const int32_t* __restrict__ y, int ldn,
                                                         const double* __restrict__ sf, const double* __restrict__ lsf,
                                                         const double* __restrict__ Xt,
                                                         const double* __restrict__ pinvXt, int ldx, int N, int P,
                                                         int full_rank, const double* __restrict__ disp, double min_mu,
                                                         double beta_tol, double min_beta, double max_beta, int maxiter,
                                                         double* __restrict__ beta, double* __restrict__ mu,
                                                         double* __restrict__ hat, uint8_t* __restrict__ conv,
                                                         int32_t* __restrict__ iters,
                                                         const int32_t* __restrict__ fb_list, int n_fb, IrlsExtras ex) {
    extern __shared__ __attribute__((aligned(16))) double wide_lds[];
    __shared__ LbfgsbWork<kWideMaxP> Lb;
    __shared__ double xlu[3 * kWideMaxP];
    __shared__ int nbd[kWideMaxP];
    const int k = blockIdx.x;
    if (k >= n_fb) return;
    const int g = fb_list[k];
    WideWork W;
    W.bind(wide_lds, P);
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = lsf; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta; A.max_beta = max_beta;
    A.maxiter = maxiter; A.full_rank = full_rank != 0;
    // beta_init of the gene (the first kernel's W is gone): recompute as irls_gene_wide does
    for (int j = 0; j < P; ++j) {
        double b0 = 0.0;
        for (int n = DeviceWave::lane(); n < N; n += 64) {
            const double yv = (double)A.y[n];
            if (A.full_rank) b0 += pinvXt[j * ldx + n] * log(yv / sf[n] + 0.1);
            else if (j == 0) b0 += log(yv / sf[n]);
        }
        b0 = DeviceWave::sum(b0);
        if (!A.full_rank) b0 = j == 0 ? b0 / (double)N : 0.0;
        if ((threadIdx.x & 63) == 0) W.v(2)[j] = b0;
    }
    DeviceWave::sync();
    LfcEpilogue E;
    wide_epilogue_begin(E, ex, g, ldn);
    if (ex.cooks_ld != 0 && E.cooks_row != nullptr) E.cooks_row = ex.cooks_tmp + (size_t)k * ldn;  // (see k_irls_rescue)
    const IrlsOut o = irls_rescue_wide<DeviceWave>(A, W, Lb, xlu, nbd, mu ? mu + (size_t)g * ldn : nullptr,
                                                   hat ? hat + (size_t)g * ldn : nullptr, &E);
    for (int j = threadIdx.x & 63; j < P; j += 64) beta[(size_t)g * P + j] = W.v(0)[j];
    if ((threadIdx.x & 63) == 0) {
        conv[g] = (uint8_t)o.converged;
        if (iters != nullptr) iters[g] = o.iters;
        wide_epilogue_store(E, ex, g);
    }
}

hipError_t launch_wide_irls(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* lsf,
                            const double* Xt, const double* pinvXt, int ldx, int N, int G, int P, int full_rank,
                            const double* disp, double min_mu, double beta_tol, double min_beta, double max_beta,
                            int maxiter, double* beta, double* mu, double* hat, uint8_t* conv, int32_t* iters,
                            int32_t* fb_count, int32_t* fb_list, const IrlsExtras* extras) {
    if (G <= 0) return hipSuccess;
    IrlsExtras ex{};
    if (extras != nullptr) ex = *extras;
    const WideGeom ge = wide_geom(P);
    set_smem(k_irls_wide, ge.per_wave * ge.wpb);
    hipLaunchKernelGGL(k_irls_wide, dim3((G + ge.wpb - 1) / ge.wpb), dim3(64 * ge.wpb), ge.per_wave * ge.wpb, st, y,
                       ldn, sf, lsf, Xt, pinvXt, ldx, N, G, P, (int)(ge.per_wave / 8), full_rank, disp, min_mu,
                       beta_tol, min_beta, max_beta, maxiter, beta, mu, hat, conv, iters, fb_count, fb_list, ex);
    return hipGetLastError();
}

hipError_t launch_wide_irls_rescue(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* lsf,
                                   const double* Xt, const double* pinvXt, int ldx, int N, int P, int full_rank,
                                   const double* disp, double min_mu, double beta_tol, double min_beta,
                                   double max_beta, int maxiter, double* beta, double* mu, double* hat, uint8_t* conv,
                                   int32_t* iters, const int32_t* fb_list, int n_fb, const IrlsExtras* extras) {
    if (n_fb <= 0) return hipSuccess;
    IrlsExtras ex{};
    if (extras != nullptr) ex = *extras;
    ex.cells = CellDesign{};  // the rescue of a diverged gene runs the general evaluation
    const size_t smem = (size_t)wide_work_doubles(P) * sizeof(double);
    set_smem(k_irls_rescue_wide, smem);
    hipLaunchKernelGGL(k_irls_rescue_wide, dim3(n_fb), dim3(64), smem, st, y, ldn, sf, lsf, Xt, pinvXt, ldx, N, P,
                       full_rank, disp, min_mu, beta_tol, min_beta, max_beta, maxiter, beta, mu, hat, conv, iters,
                       fb_list, n_fb, ex);
    return hipGetLastError();
}

// layers (mu, hat) of a finished fit from beta
__global__ __launch_bounds__(256) void k_irls_layers_wide(const int32_t* __restrict__ y, int ldn,
                                                          const double* __restrict__ sf, const double* __restrict__ Xt,
                                                          int ldx, int N, int G, int P, int per_wave_doubles,
                                                          const double* __restrict__ disp,
                                                          const double* __restrict__ beta, double min_mu,
                                                          double* __restrict__ mu, double* __restrict__ hat) {
    DSQ_WIDE_PROLOGUE(G);
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = nullptr; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = 0.0; A.min_beta = 0.0; A.max_beta = 0.0; A.maxiter = 0;
    A.full_rank = false;
    for (int j = threadIdx.x & 63; j < P; j += 64) W.v(0)[j] = beta[(size_t)g * P + j];
    DeviceWave::sync();
    double S;
    irls_sweep_wide<DeviceWave>(A, W, 1.0 / A.disp, S);
    irls_finish_wide<DeviceWave>(A, W, mu ? mu + (size_t)g * ldn : nullptr, hat ? hat + (size_t)g * ldn : nullptr,
                                 nullptr);
}

hipError_t launch_wide_irls_layers(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                                   int ldx, int N, int G, int P, const double* disp, const double* beta, double min_mu,
                                   double* mu, double* hat) {
    if (G <= 0) return hipSuccess;
    const WideGeom ge = wide_geom(P);
    set_smem(k_irls_layers_wide, ge.per_wave * ge.wpb);
    hipLaunchKernelGGL(k_irls_layers_wide, dim3((G + ge.wpb - 1) / ge.wpb), dim3(64 * ge.wpb), ge.per_wave * ge.wpb,
                       st, y, ldn, sf, Xt, ldx, N, G, P, (int)(ge.per_wave / 8), disp, beta, min_mu, mu, hat);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_wald_wide(const double* __restrict__ mu, int ldn, const double* __restrict__ sf,
                                                   const double* __restrict__ Xt, int ldx, int N, int G, int P,
                                                   int per_wave_doubles, const double* __restrict__ disp,
                                                   const double* __restrict__ beta, const double* __restrict__ ridge,
                                                   const double* __restrict__ contrast, double lfc_null, int alt,
                                                   double* __restrict__ pvals, double* __restrict__ stats,
                                                   double* __restrict__ se) {
    DSQ_WIDE_PROLOGUE(G);
    for (int j = threadIdx.x & 63; j < P; j += 64) W.v(0)[j] = beta[(size_t)g * P + j];
    DeviceWave::sync();
    const WaldOut o = wald_gene_wide<DeviceWave>(mu ? mu + (size_t)g * ldn : nullptr, sf, Xt, ldx, N, disp[g], W, ridge,
                                                 contrast, lfc_null, alt);
    if ((threadIdx.x & 63) == 0) {
        pvals[g] = o.p;
        stats[g] = o.stat;
        se[g] = o.se;
    }
}

hipError_t launch_wide_wald(hipStream_t st, const double* mu, int ldn, const double* sf, const double* Xt, int ldx,
                            int N, int G, int P, const double* disp, const double* beta, const double* d_ridge,
                            const double* d_contrast, double lfc_null, int alt, double* pvals, double* stats,
                            double* se) {
    if (G <= 0) return hipSuccess;
    const WideGeom ge = wide_geom(P);
    set_smem(k_wald_wide, ge.per_wave * ge.wpb);
    hipLaunchKernelGGL(k_wald_wide, dim3((G + ge.wpb - 1) / ge.wpb), dim3(64 * ge.wpb), ge.per_wave * ge.wpb, st, mu,
                       ldn, sf, Xt, ldx, N, G, P, (int)(ge.per_wave / 8), disp, beta, d_ridge, d_contrast, lfc_null, alt,
                       pvals, stats, se);
    return hipGetLastError();
}

// the width from which the register / cell kernels hand over to this path (DSQ_WIDE_MIN_P: measurements)
int wide_min_p() {
    static const int v = [] {
        const char* e = getenv("DSQ_WIDE_MIN_P");
        const int x = e ? atoi(e) : DSQ_REG_MAX_P + 1;
        return x < 1 ? 1 : (x > DSQ_REG_MAX_P + 1 ? DSQ_REG_MAX_P + 1 : x);
    }();
    return v;
}

bool wide_with_cells() {
    static const bool v = getenv("DSQ_WIDE_CELLS") != nullptr;
    return v;
}

}  // namespace dsq
