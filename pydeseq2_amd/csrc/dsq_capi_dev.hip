// dsq_capi_dev.hip — the stages of the path on device-resident buffers (dsq_dev_*, dsq_mix_*; include/deseq_hip.h).
#include "dsq_capi_internal.h"

extern "C" {

// ------------------------------------------------------------------ device-resident stages
int dsq_dev_counts_to_gene_major(dsq_ctx* ctx, const void* d_src, int count_type, int layout, int N, int G,
                                 int32_t* d_dst, int ldn, int* h_bad) {
    DSQ_CHECK_ARG(ldn >= N, "ldn < N");
    int* d_bad = (int*)ctx->d_scratch;
    DSQ_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
    DSQ_HIP(dsq::launch_transpose_counts(ctx->stream, d_src, count_type, layout, N, G, d_dst, ldn, d_bad));
    if (h_bad) {
        DSQ_HIP(hipMemcpyAsync(h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return DSQ_OK;
}

int dsq_dev_f64_to_gene_major(dsq_ctx* ctx, const double* d_src, int layout, int N, int G, double* d_dst,
                              int ldn) {
    DSQ_HIP(dsq::launch_transpose_f64(ctx->stream, d_src, layout, N, G, d_dst, ldn));
    return DSQ_OK;
}

int dsq_dev_logmeans(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, double* d_logmeans,
                     uint8_t* d_nonzero) {
    DSQ_HIP(dsq::launch_logmeans(ctx->stream, d_y, ldn, N, G, d_logmeans, d_nonzero));
    return DSQ_OK;
}

int dsq_dev_logmeans_poscounts(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, double* d_logmeans,
                               uint8_t* d_usable) {
    DSQ_HIP(dsq::launch_logmeans_pos(ctx->stream, d_y, ldn, N, G, d_logmeans, d_usable));
    return DSQ_OK;
}

size_t dsq_size_factors_work_doubles(int N, int G) { return dsq::size_factors_work_doubles(N, G); }

int dsq_dev_size_factors(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                         const double* d_logmeans, const uint8_t* d_gene_mask, double* d_work,
                         double* d_size_factors) {
    DSQ_HIP(dsq::launch_size_factors(ctx->stream, d_counts_sm, count_type, N, G, d_logmeans, d_gene_mask,
                                     d_work, d_size_factors));
    return DSQ_OK;
}

int dsq_dev_size_factors_new(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                             const double* d_logmeans, const uint8_t* d_gene_mask, double* d_work,
                             double* d_size_factors) {
    DSQ_HIP(dsq::launch_size_factors(ctx->stream, d_counts_sm, count_type, N, G, d_logmeans, d_gene_mask,
                                     d_work, d_size_factors, 1));
    return DSQ_OK;
}

int dsq_dev_mom(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                double* d_normed_mean, double* d_rough, double* d_moments, double* d_mom) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(N != P, "The number of samples and the number of design variables are equal, i.e., "
                          "there are no replicates to estimate the dispersion.");
    DSQ_HIP(dsq::launch_mom(ctx->stream, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, min_disp, max_disp,
                            d_normed_mean, d_rough, d_moments, d_mom, ctx->d_scratch + 8));
    return DSQ_OK;
}

int dsq_dev_mom_lin_mu(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                       const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                       double min_mu, double* d_normed_mean, double* d_mom, double* d_mu) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(N != P, "The number of samples and the number of design variables are equal, i.e., "
                          "there are no replicates to estimate the dispersion.");
    DSQ_HIP(dsq::launch_mom_lin_mu(ctx->stream, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, min_disp, max_disp,
                                   min_mu, d_normed_mean, d_mom, d_mu, ctx->d_scratch + 8));
    return DSQ_OK;
}

int dsq_dev_mom_lin_coef(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                         const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                         double min_mu, double* d_normed_mean, double* d_mom, double* d_mu, double* d_coef) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(N != P, "The number of samples and the number of design variables are equal, i.e., "
                          "there are no replicates to estimate the dispersion.");
    DSQ_HIP(dsq::launch_mom_lin_mu(ctx->stream, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, min_disp, max_disp,
                                   min_mu, d_normed_mean, d_mom, d_mu, ctx->d_scratch + 8, d_coef));
    return DSQ_OK;
}

int dsq_dev_mom_raw(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_ones, const double* d_sf,
                    const double* d_Xt, const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp,
                    double max_disp, double* d_normed_mean, double* d_mom) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(N != P, "The number of samples and the number of design variables are equal, i.e., "
                          "there are no replicates to estimate the dispersion.");
    DSQ_HIP(dsq::launch_mom(ctx->stream, d_y, ldn, d_ones, d_Xt, d_pinvXt, ldx, N, G, P, min_disp, max_disp,
                            d_normed_mean, nullptr, nullptr, d_mom, ctx->d_scratch + 8, d_sf));
    return DSQ_OK;
}

int dsq_dev_nll_const(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, const double* d_disp, double* d_cst) {
    DSQ_HIP(dsq::launch_nll_const(ctx->stream, d_y, ldn, N, G, d_disp, d_cst));
    return DSQ_OK;
}

int dsq_dev_nll_scaled(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, int N, int G,
                       const double* d_disp, const double* d_scale, const double* d_cst, double* d_nll) {
    DSQ_HIP(dsq::launch_nll_scaled(ctx->stream, d_y, d_mu, ldn, N, G, d_disp, d_scale, d_cst, d_nll));
    return DSQ_OK;
}

int dsq_dev_lin_mu(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                   const double* d_pinvXt, int ldx, int N, int G, int P, double min_mu, double* d_mu) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_HIP(dsq::launch_lin_mu(ctx->stream, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, min_mu, d_mu));
    return DSQ_OK;
}

int dsq_dev_alpha_mle(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt,
                      int ldx, int N, int G, int P, const double* d_alpha_hat, double min_disp,
                      double max_disp, double prior_disp_var, int cr_reg, int prior_reg, double* d_alpha,
                      uint8_t* d_converged, int32_t* d_nfev, double* d_nll_const, int const_mode) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(const_mode >= DSQ_CONST_COMPUTE && const_mode <= DSQ_CONST_LOAD, "const_mode out of range");
    return run_alpha(ctx, d_y, d_mu, ldn, d_Xt, ldx, N, G, P, d_alpha_hat, min_disp, max_disp, prior_disp_var,
                     cr_reg, prior_reg, d_alpha, d_converged, d_nfev, d_nll_const, const_mode);
}


int dsq_set_alpha_hook(dsq_ctx* ctx, dsq_hook_fn fn, void* arg) {
    ctx->alpha_hook = fn;
    ctx->alpha_hook_arg = arg;
    return DSQ_OK;
}

int dsq_irls_order_hint(dsq_ctx* ctx, const int32_t* d_iters, int G) {
    ctx->d_irls_hint = d_iters;
    ctx->irls_hint_genes = d_iters != nullptr ? G : 0;
    return DSQ_OK;
}


int dsq_dev_irls(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                 const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank, const double* d_disp,
                 double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter,
                 double* d_beta, double* d_mu, double* d_hat, uint8_t* d_converged, int32_t* d_iters) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    return run_irls(ctx, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, full_rank, d_disp, min_mu, beta_tol, min_beta,
                    max_beta, maxiter, d_beta, d_mu, d_hat, d_converged, d_iters, nullptr);
}


int dsq_dev_alpha_mle4(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx,
                       int N, int G, int P, const double* d_alpha_hat, double min_disp, double max_disp,
                       double prior_disp_var, int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_converged,
                       int32_t* d_nfev, double* d_nll_const, int const_mode, const dsq_cells* cells,
                       const double* d_coef, const double* d_sf, double min_mu, const int32_t* d_rows, int n_rows,
                       const int32_t* d_waves, int n_waves, const double* d_cell_mu, const dsq_mix* mix,
                       const double* d_beta) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(const_mode >= DSQ_CONST_COMPUTE && const_mode <= DSQ_CONST_LOAD, "const_mode out of range");
    DSQ_CHECK_ARG(d_mu != nullptr || (d_coef != nullptr && d_sf != nullptr) ||
                      (d_cell_mu != nullptr && d_sf != nullptr && cells != nullptr && cells->n_cells > 0) ||
                      (mix != nullptr && d_beta != nullptr && d_sf != nullptr),
                  "mu_hat is needed as a matrix, as (coef, sf), as (cell_mu, sf, cells) or as (mix, beta, sf)");
    DSQ_CHECK_ARG(mix == nullptr || (mix->d.P == P && mix->d.N == N && d_rows != nullptr),
                  "mix: built for another design, or the gene lists are missing");
    DSQ_CHECK_ARG(d_beta == nullptr || (mix != nullptr && d_mu == nullptr && d_coef == nullptr && d_cell_mu == nullptr &&
                                        n_waves == 0),
                  "beta: with mix only, alone, and every gene on the mixed-design kernel");
    DSQ_CHECK_ARG(d_cell_mu == nullptr || (d_mu == nullptr && d_coef == nullptr && !dsq::alpha_is_wide(P, cells->n_cells)),
                  "cell_mu: alone, on the register kernels");
    DSQ_CHECK_ARG(cells == nullptr || cells->n_cells <= dsq::kMaxCells, "too many design cells for the cell path");
    DSQ_CHECK_ARG(d_rows == nullptr || (n_rows >= 0 && n_waves >= 0 && n_rows + n_waves == G &&
                                        (n_waves == 0 || d_waves != nullptr)),
                  "the two gene lists must partition the G genes of the call");
    dsq::AlphaExtras ex{};
    ex.cells = to_cells(cells);
    if (d_mu == nullptr) { ex.coef = d_coef; ex.cell_mu = d_cell_mu; ex.sf = d_sf; ex.min_mu = min_mu; }
    if (d_rows != nullptr) { ex.rows = d_rows; ex.n_rows = n_rows; ex.waves = d_waves; ex.n_waves = n_waves; }
    if (mix != nullptr) { ex.mix = &mix->d; ex.mix_beta = d_beta; ex.sf = d_sf; }
    return run_alpha(ctx, d_y, d_mu, ldn, d_Xt, ldx, N, G, P, d_alpha_hat, min_disp, max_disp, prior_disp_var, cr_reg,
                     prior_reg, d_alpha, d_converged, d_nfev, d_nll_const, const_mode, &ex);
}

int dsq_dev_alpha_mle3(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx,
                       int N, int G, int P, const double* d_alpha_hat, double min_disp, double max_disp,
                       double prior_disp_var, int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_converged,
                       int32_t* d_nfev, double* d_nll_const, int const_mode, const dsq_cells* cells,
                       const double* d_coef, const double* d_sf, double min_mu, const int32_t* d_rows, int n_rows,
                       const int32_t* d_waves, int n_waves, const double* d_cell_mu) {
    return dsq_dev_alpha_mle4(ctx, d_y, d_mu, ldn, d_Xt, ldx, N, G, P, d_alpha_hat, min_disp, max_disp, prior_disp_var,
                              cr_reg, prior_reg, d_alpha, d_converged, d_nfev, d_nll_const, const_mode, cells, d_coef,
                              d_sf, min_mu, d_rows, n_rows, d_waves, n_waves, d_cell_mu, nullptr, nullptr);
}

int dsq_dev_alpha_mle2(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx,
                       int N, int G, int P, const double* d_alpha_hat, double min_disp, double max_disp,
                       double prior_disp_var, int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_converged,
                       int32_t* d_nfev, double* d_nll_const, int const_mode, const dsq_cells* cells,
                       const double* d_coef, const double* d_sf, double min_mu) {
    return dsq_dev_alpha_mle3(ctx, d_y, d_mu, ldn, d_Xt, ldx, N, G, P, d_alpha_hat, min_disp, max_disp, prior_disp_var,
                              cr_reg, prior_reg, d_alpha, d_converged, d_nfev, d_nll_const, const_mode, cells, d_coef,
                              d_sf, min_mu, nullptr, 0, nullptr, 0, nullptr);
}

// ------------------------------------------------------------------ mixed designs (csrc/dsq_mix.h)
// Analysis of a design matrix (row-major N x P), once per design: which columns are continuous covariates, the cells
// of the remaining (categorical) columns, the slot order of the samples.  *out = NULL (and DSQ_OK): not a mixed
// design the kernels take - the caller stays on the general path.
int dsq_mix_create(dsq_ctx* ctx, const double* design, int N, int P, dsq_mix** out) {
    DSQ_CHECK_ARG(out != nullptr && design != nullptr, "null argument");
    *out = nullptr;
    DSQ_HIP(hipSetDevice(ctx->device));  // (the descriptor's block must live on this context's GPU)
    if (P < 1 || P > dsq::kMixMaxP || N < 2 || N > 65535 || !dsq::alpha_mix_enabled()) return DSQ_OK;
    const bool force = getenv("DSQ_MIX_FORCE") != nullptr;  // tests: also designs whose padding exceeds the waste limit
    // columns by decreasing number of distinct values
    std::vector<int> nd((size_t)P), order((size_t)P);
    for (int j = 0; j < P; ++j) {
        std::vector<double> col((size_t)N);
        for (int n = 0; n < N; ++n) col[(size_t)n] = design[(size_t)n * P + j];
        std::sort(col.begin(), col.end());
        nd[(size_t)j] = (int)(std::unique(col.begin(), col.end()) - col.begin());
        order[(size_t)j] = j;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nd[(size_t)a] > nd[(size_t)b]; });
    std::vector<char> cont((size_t)P, 0);
    std::vector<int> idx((size_t)N), cell((size_t)N);
    int C = 0;
    auto less_cat = [&](int a, int b) {  // lexicographic on the categorical columns, then by sample index
        for (int j = 0; j < P; ++j) {
            if (cont[(size_t)j]) continue;
            const double va = design[(size_t)a * P + j], vb = design[(size_t)b * P + j];
            if (va != vb) return va < vb;
        }
        return a < b;
    };
    auto same_cat = [&](int a, int b) {
        for (int j = 0; j < P; ++j)
            if (!cont[(size_t)j] && design[(size_t)a * P + j] != design[(size_t)b * P + j]) return false;
        return true;
    };
    auto find_cells = [&]() {
        for (int n = 0; n < N; ++n) idx[(size_t)n] = n;
        std::sort(idx.begin(), idx.end(), less_cat);
        C = 0;
        for (int k = 0; k < N; ++k) {
            if (k > 0 && !same_cat(idx[(size_t)k - 1], idx[(size_t)k])) ++C;
            cell[(size_t)idx[(size_t)k]] = C;
        }
        ++C;
    };
    int Q = 0;
    find_cells();
    while (C > dsq::kMixMaxCells && Q < dsq::kMixMaxQ && Q < P) {
        cont[(size_t)order[(size_t)Q]] = 1;
        ++Q;
        find_cells();
    }
    if (Q == 0 || C > dsq::kMixMaxCells) return DSQ_OK;  // purely categorical (the cell kernels), or too many covariates
    // slot order: cells one after the other (idx is sorted by cell, then sample), each padded to whole loop iterations
    // of the kernels (kMixU trips of 64 slots)
    std::vector<int> count((size_t)C, 0);
    for (int n = 0; n < N; ++n) ++count[(size_t)cell[(size_t)n]];
    const int blk = 64 * dsq::kMixU;
    int Ns = 0;
    for (int c = 0; c < C; ++c) Ns += (count[(size_t)c] + blk - 1) / blk * blk;
    Ns = (Ns + 255) & ~255;  // whole blocks of four trips (the staging passes walk four at a time); the tail is padding
    if (!force && Ns > N + N / 2 + 256) return DSQ_OK;  // mostly padding (small cells): the general kernels do less work
    dsq::MixDesign M{};
    M.Ns = Ns; M.C = C; M.Q = Q; M.P = P; M.N = N;
    {
        int q = 0;
        for (int j = 0; j < P; ++j) {
            M.colq[j] = cont[(size_t)j] ? q : -1;
            if (cont[(size_t)j]) M.zcol[q++] = j;
        }
    }
    if (!dsq::alpha_mix_fits(M)) return DSQ_OK;  // rows too long for the kernel's LDS staging
    std::vector<int32_t> perm((size_t)Ns, -1), slot_of((size_t)N, 0);
    std::vector<uint8_t> trip_cell((size_t)(Ns / 64), (uint8_t)(C - 1));
    std::vector<double> Zs((size_t)Q * Ns, 0.0), Xc((size_t)C * P, 0.0), Ginv;
    {
        int s = 0, k = 0;
        for (int c = 0; c < C; ++c) {
            const int s0 = s;
            for (int i = 0; i < count[(size_t)c]; ++i, ++k, ++s) {
                const int n = idx[(size_t)k];
                perm[(size_t)s] = n;
                slot_of[(size_t)n] = s;
                for (int q = 0; q < Q; ++q) Zs[(size_t)q * Ns + s] = design[(size_t)n * P + M.zcol[q]];
                if (i == 0)
                    for (int j = 0; j < P; ++j) Xc[(size_t)c * P + j] = cont[(size_t)j] ? 0.0 : design[(size_t)n * P + j];
            }
            s = s0 + (count[(size_t)c] + blk - 1) / blk * blk;
            for (int t = s0 / 64; t < s / 64; ++t) trip_cell[(size_t)t] = (uint8_t)c;
        }
    }
    {   // (X^T X)^-1 by Cholesky in extended precision (start values of the IRLS kernel); skipped when rank deficient
        std::vector<long double> A((size_t)P * P, 0.0L), Li((size_t)P * P, 0.0L);
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < P; ++i)
                for (int j = 0; j <= i; ++j) A[(size_t)i * P + j] += (long double)design[(size_t)n * P + i] * design[(size_t)n * P + j];
        bool ok = true;
        long double dmax_ = 0.0L;
        for (int i = 0; i < P; ++i) dmax_ = std::max(dmax_, A[(size_t)i * P + i]);
        for (int j = 0; j < P && ok; ++j) {
            long double d = A[(size_t)j * P + j];
            for (int k = 0; k < j; ++k) d -= A[(size_t)j * P + k] * A[(size_t)j * P + k];
            if (!(d > dmax_ * 1e-13L)) { ok = false; break; }
            d = std::sqrt(d);
            A[(size_t)j * P + j] = d;
            for (int i = j + 1; i < P; ++i) {
                long double v = A[(size_t)i * P + j];
                for (int k = 0; k < j; ++k) v -= A[(size_t)i * P + k] * A[(size_t)j * P + k];
                A[(size_t)i * P + j] = v / d;
            }
        }
        if (ok) {
            for (int j = 0; j < P; ++j) {  // L^-1, column by column
                Li[(size_t)j * P + j] = 1.0L / A[(size_t)j * P + j];
                for (int i = j + 1; i < P; ++i) {
                    long double v = 0.0L;
                    for (int k = j; k < i; ++k) v -= A[(size_t)i * P + k] * Li[(size_t)k * P + j];
                    Li[(size_t)i * P + j] = v / A[(size_t)i * P + i];
                }
            }
            Ginv.assign((size_t)P * P, 0.0);
            for (int i = 0; i < P; ++i)
                for (int j = 0; j < P; ++j) {
                    long double v = 0.0L;
                    for (int k = std::max(i, j); k < P; ++k) v += Li[(size_t)k * P + i] * Li[(size_t)k * P + j];
                    Ginv[(size_t)i * P + j] = (double)v;
                }
        }
    }
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_perm = up((size_t)Ns * 4), b_tc = up((size_t)Ns / 64), b_z = up((size_t)Q * Ns * 8),
                 b_xc = up((size_t)C * P * 8), b_g = up((size_t)P * P * 8), b_so = up((size_t)N * 4);
    dsq_mix* m = new dsq_mix();
    m->device = ctx->device;
    hipError_t e = hipMalloc(&m->d_block, b_perm + b_tc + b_z + b_xc + b_g + b_so);
    if (e != hipSuccess) { delete m; return fail(ctx, DSQ_ERR_HIP, std::string("dsq_mix_create: ") + hipGetErrorString(e)); }
    char* p = (char*)m->d_block;
    auto put = [&](const void* src, size_t bytes, size_t slot) {
        char* dst = p;
        if (e == hipSuccess && bytes) e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
        p += slot;
        return dst;
    };
    M.perm = (const int32_t*)put(perm.data(), (size_t)Ns * 4, b_perm);
    M.trip_cell = (const uint8_t*)put(trip_cell.data(), (size_t)Ns / 64, b_tc);
    M.Zs = (const double*)put(Zs.data(), (size_t)Q * Ns * 8, b_z);
    M.Xc = (const double*)put(Xc.data(), (size_t)C * P * 8, b_xc);
    const char* g = put(Ginv.empty() ? nullptr : Ginv.data(), Ginv.empty() ? 0 : (size_t)P * P * 8, b_g);
    M.Ginv = Ginv.empty() ? nullptr : (const double*)g;
    M.slot_of = (const int32_t*)put(slot_of.data(), (size_t)N * 4, b_so);
    if (e != hipSuccess) {
        (void)hipFree(m->d_block);
        delete m;
        return fail(ctx, DSQ_ERR_HIP, std::string("dsq_mix_create: ") + hipGetErrorString(e));
    }
    m->d = M;
    *out = m;
    return DSQ_OK;
}

void dsq_mix_destroy(dsq_mix* mix) {
    if (mix == nullptr) return;
    (void)hipSetDevice(mix->device);
    if (mix->d_block) (void)hipFree(mix->d_block);
    delete mix;
}

int dsq_mix_slots(const dsq_mix* mix, int32_t* h_slot_of) {
    if (mix == nullptr || h_slot_of == nullptr) return DSQ_ERR_ARG;
    return hipMemcpy(h_slot_of, mix->d.slot_of, (size_t)mix->d.N * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess
               ? DSQ_OK
               : DSQ_ERR_HIP;
}

// Slot-ordered copies for the mixed-design kernels (dsq_mix.h): written once per count matrix / per fit by the caller and
// handed to the NEXT dispersion or IRLS fit of this context (one-shot; any of the three may be NULL - the fit then builds
// what it lacks itself, per call).
int dsq_mix_bind(dsq_ctx* ctx, const uint16_t* d_ys, const uint8_t* d_big, const double* d_mu_slots) {
    ctx->bind_ys = d_ys;
    ctx->bind_big = d_big;
    ctx->bind_mu = d_mu_slots;
    ctx->bind_G = 0;
    return DSQ_OK;
}

int dsq_mix_bind2(dsq_ctx* ctx, const uint16_t* d_ys, const uint8_t* d_big, const double* d_mu_slots, int G) {
    const int rc = dsq_mix_bind(ctx, d_ys, d_big, d_mu_slots);
    ctx->bind_G = (rc == DSQ_OK && G > 0) ? G : 0;
    return rc;
}
int dsq_dev_mix_counts_to_slots(dsq_ctx* ctx, const int32_t* d_y, int ldn, int G, const dsq_mix* mix, uint16_t* d_ys,
                                uint8_t* d_big) {
    DSQ_CHECK_ARG(mix != nullptr && d_y != nullptr && d_ys != nullptr && d_big != nullptr, "null argument");
    DSQ_HIP(dsq::launch_mix_counts_to_slots(ctx->stream, d_y, ldn, mix->d, G, d_ys, d_big));
    return DSQ_OK;
}
int dsq_dev_mix_mu_slots(dsq_ctx* ctx, const dsq_mix* mix, const double* d_beta, const double* d_sf, int G,
                         double* d_mu_slots) {
    DSQ_CHECK_ARG(mix != nullptr && d_beta != nullptr && d_sf != nullptr && d_mu_slots != nullptr, "null argument");
    DSQ_HIP(dsq::launch_mix_mu_slots(ctx->stream, d_beta, d_sf, mix->d, G, d_mu_slots));
    return DSQ_OK;
}

int dsq_mix_takes_irls(const dsq_mix* mix, int full_rank) {
    return (mix != nullptr && dsq::irls_takes_mix(&mix->d, full_rank)) ? 1 : 0;
}

int dsq_mix_launch_count(void) { return dsq::alpha_mix_launches(); }

int dsq_mix_info(const dsq_mix* mix, int* n_slots, int* n_cells, int* n_continuous) {
    if (mix == nullptr) return DSQ_ERR_ARG;
    if (n_slots) *n_slots = mix->d.Ns;
    if (n_cells) *n_cells = mix->d.C;
    if (n_continuous) *n_continuous = mix->d.Q;
    return DSQ_OK;
}

int dsq_alpha_needs_mu(int N, int P, int n_cells) { return dsq::alpha_needs_mu(N, P, n_cells) ? 1 : 0; }

int dsq_alpha_rows_eligible(int N, int P, int n_cells) {
    if (dsq::alpha_rows_eligible(N, P, n_cells, true, 1)) return 1;  // <= 4 cells == columns: per-cell sums in registers
    return dsq::alpha_rowsc_tail(N, P, n_cells) > 0 ? 2 : 0;         // up to 32 cells: per-cell tables in LDS
}

int dsq_dev_cell_mu(dsq_ctx* ctx, const double* d_beta, const dsq_cells* cells, int G, int P, double* d_cell_mu) {
    DSQ_CHECK_ARG(P >= 1 && P <= 12 && cells != nullptr && cells->n_cells > 0, "cells / P out of range");
    DSQ_HIP(dsq::launch_cell_mu(ctx->stream, d_beta, cells->d_Xc, cells->n_cells, G, P, d_cell_mu));
    return DSQ_OK;
}

int dsq_dev_alpha_row_split(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, int32_t* d_flags) {
    DSQ_HIP(dsq::launch_count_big(ctx->stream, d_y, ldn, N, G, d_flags));
    return DSQ_OK;
}

int dsq_dev_robust_disp2(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const int32_t* d_cell_offsets,
                         const int32_t* d_cell_index, int n_cells, int whole, int max_cell, int min_cell, int N, int G,
                         double* d_robust_disp) {
    if (whole) min_cell = N;  // (cells of any size: beyond a wavefront's LDS the buffer-less kernel takes the design)
    {   // (runs on the side stream from inside another call: attribute an error left behind by an earlier launch to it)
        const hipError_t pend = hipGetLastError();
        if (pend != hipSuccess)
            return fail(ctx, DSQ_ERR_HIP, std::string("HIP error pending before dsq_dev_robust_disp2: ") + hipGetErrorString(pend));
    }
    if ((size_t)G + 1 > ctx->redo_cap) {
        if (ctx->d_redo) (void)hipFree(ctx->d_redo);
        ctx->d_redo = nullptr; ctx->redo_cap = 0;
        DSQ_HIP(hipMalloc((void**)&ctx->d_redo, ((size_t)G + 1 + (size_t)G / 4) * sizeof(int32_t)));
        ctx->redo_cap = (size_t)G + 1 + (size_t)G / 4;
    }
    DSQ_HIP(dsq::launch_robust_disp(ctx->stream, d_y, ldn, d_sf, d_cell_offsets, d_cell_index, n_cells, whole, max_cell,
                                    N, G, d_robust_disp, min_cell, ctx->d_redo));
    return DSQ_OK;
}

int dsq_dev_robust_disp(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const int32_t* d_cell_offsets,
                        const int32_t* d_cell_index, int n_cells, int whole, int max_cell, int N, int G,
                        double* d_robust_disp) {
    return dsq_dev_robust_disp2(ctx, d_y, ldn, d_sf, d_cell_offsets, d_cell_index, n_cells, whole, max_cell, 0, N, G,
                                d_robust_disp);
}

int dsq_dev_irls_layers(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt, int ldx,
                        int N, int G, int P, const double* d_disp, const double* d_beta, double min_mu, double* d_mu,
                        double* d_hat) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_HIP(dsq::launch_irls_layers(ctx->stream, d_y, ldn, d_sf, d_Xt, ldx, N, G, P, d_disp, d_beta, min_mu, d_mu,
                                    d_hat));
    return DSQ_OK;
}


int dsq_dev_lfc_fit2(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                     const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank, const double* d_disp,
                     double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter, double* d_beta,
                     double* d_mu, double* d_hat, uint8_t* d_converged, int32_t* d_iters, const dsq_cells* cells,
                     const double* d_robust_disp, const uint8_t* d_flags, double cutoff, double* d_cooks,
                     uint8_t* d_any_all, uint8_t* d_any_use, uint8_t* d_any_use_nr, uint8_t* d_few_above,
                     const double* h_ridge, const double* h_contrast, double lfc_null, int alt, double* d_pvals,
                     double* d_stats, double* d_se, const dsq_mix* mix, int cooks_ld) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(mix == nullptr || (mix->d.P == P && mix->d.N == N), "mix: built for another design");
    DSQ_CHECK_ARG(cooks_ld == 0 || (mix != nullptr && cooks_ld >= mix->d.Ns && dsq::irls_takes_mix(&mix->d, full_rank)),
                  "cooks_ld: a slot-ordered Cook's layer needs a mixed design the kernel takes and a pitch >= its slots");
    DSQ_CHECK_ARG(cells == nullptr || cells->n_cells <= dsq::kMaxCells, "too many design cells for the cell path");
    DSQ_CHECK_ARG(d_flags == nullptr || (d_robust_disp && d_any_all && d_any_use && d_any_use_nr && d_few_above),
                  "the fused Cook's bookkeeping needs the robust dispersions and the four flag vectors");
    DSQ_CHECK_ARG(h_ridge == nullptr || (h_contrast && d_pvals && d_stats && d_se && alt >= 0 && alt <= 4),
                  "the fused Wald test needs contrast, outputs and a valid alternative");
    if (G <= 0) return DSQ_OK;
    dsq::IrlsExtras ex{};
    ex.cells = to_cells(cells);
    if (mix != nullptr) ex.mix = &mix->d;
    ex.cooks_ld = cooks_ld;
    if (d_flags != nullptr) {
        ex.robust_disp = d_robust_disp; ex.flags = d_flags; ex.cutoff = cutoff; ex.cooks = d_cooks;
        ex.any_all = d_any_all; ex.any_use = d_any_use; ex.any_use_nr = d_any_use_nr; ex.few_above = d_few_above;
    }
    if (h_ridge != nullptr) {
        double* d_ridge = ctx->d_scratch + 1664;  // (behind the trend kernels' partials and outputs)
        double* d_contrast = d_ridge + P * P;  // (behind the matrix: both travel in one copy)
        // via page-locked memory: the caller's arrays may be temporaries, and a pageable source would make the
        // copy (and the launch behind it) wait for the host
        // (stream-ordered: a rescue kernel of the previous call may still be reading them; the slot itself is free
        // again because every call ends behind a synchronisation that follows its copies)
        // (ints [16, 3072) of the page-locked block hold up to 32 x 32 + 32 doubles; wider designs stage behind the trend's
        // outputs at int 4096, where 48 x 48 + 48 doubles fit into the block's second half)
        // (the second launch of a fit in two, dsq_lfc_set_part: the first one's copy - the same arguments - is in place,
        // and that launch may still be reading it)
        const bool in_place = ctx->lfc_part != nullptr &&
                              (ctx->lfc_phase == 2 || (ctx->lfc_phase == 1 && ctx->lfc_prepared_wald &&
                                                       ctx->lfc_prepared_N == N && ctx->lfc_prepared_P == P));
        if (!in_place) {
            double* h_stage = (double*)(ctx->h_pin + (P <= 32 ? 16 : 4096));
            std::memcpy(h_stage, h_ridge, (size_t)P * P * sizeof(double));
            std::memcpy(h_stage + P * P, h_contrast, (size_t)P * sizeof(double));
            DSQ_HIP(hipMemcpyAsync(d_ridge, h_stage, (size_t)(P * P + P) * sizeof(double), hipMemcpyHostToDevice,
                                   ctx->stream));
        }
        ex.ridge = d_ridge; ex.contrast = d_contrast; ex.lfc_null = lfc_null; ex.alt = alt;
        ex.pvals = d_pvals; ex.stats = d_stats; ex.se = d_se;
    }
    return run_irls(ctx, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, full_rank, d_disp, min_mu, beta_tol, min_beta,
                    max_beta, maxiter, d_beta, d_mu, d_hat, d_converged, d_iters, &ex);
}

int dsq_dev_lfc_fit(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                    const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank, const double* d_disp,
                    double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter, double* d_beta,
                    double* d_mu, double* d_hat, uint8_t* d_converged, int32_t* d_iters, const dsq_cells* cells,
                    const double* d_robust_disp, const uint8_t* d_flags, double cutoff, double* d_cooks,
                    uint8_t* d_any_all, uint8_t* d_any_use, uint8_t* d_any_use_nr, uint8_t* d_few_above,
                    const double* h_ridge, const double* h_contrast, double lfc_null, int alt, double* d_pvals,
                    double* d_stats, double* d_se) {
    return dsq_dev_lfc_fit2(ctx, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, full_rank, d_disp, min_mu, beta_tol,
                            min_beta, max_beta, maxiter, d_beta, d_mu, d_hat, d_converged, d_iters, cells, d_robust_disp,
                            d_flags, cutoff, d_cooks, d_any_all, d_any_use, d_any_use_nr, d_few_above, h_ridge, h_contrast,
                            lfc_null, alt, d_pvals, d_stats, d_se, nullptr, 0);
}

int dsq_dev_cooks(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_mu,
                  const double* d_hat, const int32_t* d_cell_offsets, const int32_t* d_cell_index,
                  int n_cells, int whole, int max_cell, const uint8_t* d_flags, int N, int G, int P,
                  double cutoff, double* d_cooks, double* d_robust_disp, uint8_t* d_any_all,
                  uint8_t* d_any_use, uint8_t* d_any_use_nr, uint8_t* d_few_above) {
    DSQ_CHECK_ARG((whole ? N : max_cell) <= 16384, "a design cell with more than 16384 samples is not supported");
    DSQ_HIP(dsq::launch_cooks(ctx->stream, d_y, ldn, d_sf, d_mu, d_hat, d_cell_offsets, d_cell_index,
                              n_cells, whole, max_cell, d_flags, N, G, P, cutoff, d_cooks, d_robust_disp,
                              d_any_all, d_any_use, d_any_use_nr, d_few_above));
    return DSQ_OK;
}

int dsq_dev_replace_outliers2(dsq_ctx* ctx, const int32_t* d_y, const double* d_cooks, int ldn,
                              const double* d_sf, const uint8_t* d_flags, const int32_t* d_gene_idx,
                              int n_sel, int N, double cutoff, int32_t* d_y_out, uint8_t* d_all_zero, int cooks_ld,
                              const dsq_mix* mix) {
    DSQ_CHECK_ARG(cooks_ld == 0 || (mix != nullptr && mix->d.N == N && cooks_ld >= mix->d.Ns),
                  "cooks_ld: a slot-ordered Cook's layer comes with the mixed design that wrote it");
    DSQ_HIP(dsq::launch_replace(ctx->stream, d_y, d_cooks, ldn, d_sf, d_flags, d_gene_idx, n_sel, N, cutoff,
                                d_y_out, d_all_zero, cooks_ld, cooks_ld != 0 ? mix->d.slot_of : nullptr));
    return DSQ_OK;
}

int dsq_dev_replace_outliers(dsq_ctx* ctx, const int32_t* d_y, const double* d_cooks, int ldn,
                             const double* d_sf, const uint8_t* d_flags, const int32_t* d_gene_idx,
                             int n_sel, int N, double cutoff, int32_t* d_y_out, uint8_t* d_all_zero) {
    return dsq_dev_replace_outliers2(ctx, d_y, d_cooks, ldn, d_sf, d_flags, d_gene_idx, n_sel, N, cutoff, d_y_out,
                                     d_all_zero, 0, nullptr);
}

int dsq_dev_wald(dsq_ctx* ctx, const double* d_mu, int ldn, const double* d_sf, const double* d_Xt, int ldx,
                 int N, int G, int P, const double* d_disp, const double* d_beta, const double* h_ridge,
                 const double* h_contrast, double lfc_null, int alt, double* d_pvals, double* d_stats,
                 double* d_se) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(alt >= 0 && alt <= 4, "unknown alternative hypothesis");
    double* d_ridge = ctx->d_scratch + 1664;  // (behind the trend kernels' partials and outputs)
    double* d_contrast = d_ridge + DSQ_MAX_P * DSQ_MAX_P;
    DSQ_HIP(hipMemcpyAsync(d_ridge, h_ridge, (size_t)P * P * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(d_contrast, h_contrast, (size_t)P * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    DSQ_HIP(dsq::launch_wald(ctx->stream, d_mu, ldn, d_sf, d_Xt, ldx, N, G, P, d_disp, d_beta, d_ridge,
                             d_contrast, lfc_null, alt, d_pvals, d_stats, d_se));
    return DSQ_OK;
}

int dsq_dev_gather_rows_f64(dsq_ctx* ctx, const double* d_src, int ld, const int32_t* d_idx, int n_idx,
                            int ncols, double* d_dst) {
    DSQ_HIP(dsq::launch_gather_rows_f64(ctx->stream, d_src, ld, d_idx, n_idx, ncols, d_dst));
    return DSQ_OK;
}

int dsq_dev_lfc_shrink3(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_offset, const double* d_Xt,
                        int ldx, int N, int G, int P, const double* d_size, double prior_no_shrink_scale,
                        double prior_scale, int shrink_index, double* d_beta, double* d_inv_hessian,
                        uint8_t* d_converged, double* d_ih_entry, int optimizer) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_SHRINK_MAX_P, "P out of range (apeGLM shrinkage: at most 48 design columns)");
    DSQ_CHECK_ARG(shrink_index >= 0 && shrink_index < P, "shrink_index out of range");
    DSQ_CHECK_ARG(d_ih_entry == nullptr || P <= DSQ_BFGS_MAX_P, "d_ih_entry: designs of at most 12 columns (wider: d_inv_hessian)");
    DSQ_CHECK_ARG(optimizer >= 0 && optimizer <= 2, "optimizer: 0 (L-BFGS-B), 1 (BFGS) or 2 (Newton-CG)");
    DSQ_CHECK_ARG(optimizer == 0 || P <= DSQ_BFGS_MAX_P, "optimizer BFGS / Newton-CG: designs of at most 12 columns");
    DSQ_HIP(dsq::launch_shrink(ctx->stream, d_y, ldn, d_offset, d_Xt, ldx, N, G, P, d_size, prior_no_shrink_scale,
                               prior_scale, shrink_index, d_beta, d_inv_hessian, d_converged, d_ih_entry, optimizer));
    return DSQ_OK;
}

int dsq_dev_lfc_shrink2(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_offset, const double* d_Xt,
                        int ldx, int N, int G, int P, const double* d_size, double prior_no_shrink_scale,
                        double prior_scale, int shrink_index, double* d_beta, double* d_inv_hessian,
                        uint8_t* d_converged, double* d_ih_entry) {
    return dsq_dev_lfc_shrink3(ctx, d_y, ldn, d_offset, d_Xt, ldx, N, G, P, d_size, prior_no_shrink_scale, prior_scale,
                               shrink_index, d_beta, d_inv_hessian, d_converged, d_ih_entry, 0);
}

int dsq_dev_lfc_shrink(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_offset, const double* d_Xt,
                       int ldx, int N, int G, int P, const double* d_size, double prior_no_shrink_scale,
                       double prior_scale, int shrink_index, double* d_beta, double* d_inv_hessian,
                       uint8_t* d_converged) {
    return dsq_dev_lfc_shrink2(ctx, d_y, ldn, d_offset, d_Xt, ldx, N, G, P, d_size, prior_no_shrink_scale, prior_scale,
                               shrink_index, d_beta, d_inv_hessian, d_converged, nullptr);
}

int dsq_dev_vst(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G, const double* d_sf, int mode,
                double a0, double a1, double* d_out) {
    DSQ_CHECK_ARG(mode == 0 || mode == 1, "mode: 0 parametric trend, 1 mean dispersion");
    DSQ_HIP(dsq::launch_vst(ctx->stream, d_counts_sm, count_type, N, G, d_sf, mode, a0, a1, d_out));
    return DSQ_OK;
}

int dsq_dev_trend_eval(dsq_ctx* ctx, const double* d_normed_means, int n, double a0, double a1, double* d_fitted) {
    DSQ_HIP(dsq::launch_trend_eval(ctx->stream, d_normed_means, n, a0, a1, d_fitted));
    return DSQ_OK;
}

int dsq_dev_select_dispersions(dsq_ctx* ctx, double* d_genewise_raw, double* d_map_raw,
                               const double* d_fitted, int n, double min_disp, double max_disp,
                               double squared_logres, double* d_disp, uint8_t* d_outlier) {
    DSQ_HIP(dsq::launch_select_disp(ctx->stream, d_genewise_raw, d_map_raw, d_fitted, n, min_disp, max_disp,
                                    2.0 * sqrt(squared_logres), d_disp, d_outlier));
    return DSQ_OK;
}

int dsq_dev_select_dispersions_part(dsq_ctx* ctx, double* d_genewise_raw, double* d_map_raw, const double* d_fitted, int n,
                                    double min_disp, double max_disp, double squared_logres, double* d_disp,
                                    uint8_t* d_outlier, uint8_t* d_map_converged, const uint8_t* d_conv_late, uint8_t* d_part,
                                    int mode, int ready_limit) {
    DSQ_CHECK_ARG((mode == 0 || mode == 1) && d_part != nullptr && d_map_converged != nullptr,
                  "mode 1 (the genes the full-size launch finished) or 0 (the rest), the MAP flags and the part vector");
    if (mode == 0 && ctx->lfc_pending_G != 0 && ctx->ev_lfc_part != nullptr)
        // the part vector is written on the forked stream (first thing there, long done - by construction from here on)
        DSQ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_lfc_part, 0));
    DSQ_HIP(dsq::launch_select_disp_part(ctx->stream, d_genewise_raw, d_map_raw, d_fitted, n, min_disp, max_disp,
                                         2.0 * sqrt(squared_logres), d_disp, d_outlier, d_map_converged, d_conv_late,
                                         d_part, mode, ready_limit));
    if (mode == 1 && ctx->lfc_stream != nullptr && ctx->stream == ctx->lfc_stream)
        DSQ_HIP(hipEventRecord(ctx->ev_lfc_part, ctx->lfc_stream));
    return DSQ_OK;
}

// The launches of the next dispersion fit that are enqueued after its hook's point (continuation of the parked fits, genes
// beside the row kernels) write their convergence flags to d_conv_late (one-shot): d_converged then holds, from the end of
// the full-size launch on, only what that launch wrote.  dsq_dev_select_dispersions_part (mode 0) merges them back.
int dsq_alpha_set_late_flags(dsq_ctx* ctx, uint8_t* d_conv_late) {
    ctx->alpha_conv_late = d_conv_late;
    return DSQ_OK;
}

// What the first launch of an LFC fit in two would otherwise enqueue in front of its kernel - on a stream that shares the
// device with the dispersion stage's tail, where every small operation waits tens of microseconds for a slot: the logs of
// the size factors, the Wald test's ridge and contrast, the zeroed counters.  Called on the main stream BEFORE the
// dispersion stage; consumed (one-shot) by the phase-1 call with the same N and P.
int dsq_lfc_prepare(dsq_ctx* ctx, const double* d_sf, int N, const double* h_ridge, const double* h_contrast, int P) {
    DSQ_CHECK_ARG(N >= 1 && P >= 1 && P <= DSQ_MAX_P, "N, P out of range");
    DSQ_CHECK_ARG(ctx->lfc_pending_G == 0, "a forked LFC launch is still waiting for its second launch");
    if ((size_t)N > ctx->lsf_cap) {
        if (ctx->d_lsf) (void)hipFree(ctx->d_lsf);
        ctx->d_lsf = nullptr; ctx->lsf_cap = 0;
        DSQ_HIP(hipMalloc((void**)&ctx->d_lsf, (size_t)N * sizeof(double)));
        ctx->lsf_cap = (size_t)N;
    }
    DSQ_HIP(dsq::launch_log_vec(ctx->stream, d_sf, N, ctx->d_lsf));
    DSQ_HIP(hipMemsetAsync(ctx->d_counter + 8, 0, 2 * sizeof(int32_t), ctx->stream));
    ctx->lfc_prepared_wald = 0;
    if (h_ridge != nullptr && h_contrast != nullptr) {
        double* d_ridge = ctx->d_scratch + 1664;
        double* h_stage = (double*)(ctx->h_pin + (P <= 32 ? 16 : 4096));
        std::memcpy(h_stage, h_ridge, (size_t)P * P * sizeof(double));
        std::memcpy(h_stage + P * P, h_contrast, (size_t)P * sizeof(double));
        DSQ_HIP(hipMemcpyAsync(d_ridge, h_stage, (size_t)(P * P + P) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        ctx->lfc_prepared_wald = 1;
    }
    ctx->lfc_prepared_N = N;
    ctx->lfc_prepared_P = P;
    return DSQ_OK;
}

int dsq_lfc_set_part(dsq_ctx* ctx, const uint8_t* d_part, int want, int phase) {
    DSQ_CHECK_ARG(d_part == nullptr || phase == 1 || phase == 2, "phase: 1 (forked launch) or 2 (the rest, join, rescue)");
    ctx->lfc_part = d_part;
    ctx->lfc_want = want;
    ctx->lfc_phase = d_part != nullptr ? phase : 0;
    return DSQ_OK;
}

int dsq_lfc_takes_parts(int N, int P, const dsq_cells* cells, const dsq_mix* mix, int full_rank) {
    return dsq::irls_takes_parts(N, P, cells != nullptr ? cells->n_cells : 0, mix != nullptr ? &mix->d : nullptr, full_rank)
               ? 1 : 0;
}

int dsq_dev_scatter_rows_f64(dsq_ctx* ctx, const double* d_src, const int32_t* d_idx, int n_idx, int width,
                             double* d_dst) {
    DSQ_HIP(dsq::launch_scatter_rows(ctx->stream, d_src, d_idx, n_idx, width, d_dst));
    return DSQ_OK;
}

// ---- adjusted p-values (ds.py:486-542).  Workspace layout inside ctx->d_sum for n genes:
//   [sort temp][work 4n u64][rank n i32][out 200 f64][counters 4 i32]
namespace {
struct SumWs { void* sort_tmp; void* work; int* rank; double* out; int* counters; };
hipError_t sum_workspace(dsq_ctx* ctx, int n, SumWs& w) {
    const size_t sort_b = (dsq::summary_sort_temp_bytes(n) + 255) & ~(size_t)255;
    const size_t work_b = (size_t)n * 4 * 8, rank_b = (((size_t)n * 4) + 255) & ~(size_t)255;
    const size_t total = sort_b + work_b + rank_b + 200 * 8 + 64;
    if (total > ctx->sum_cap) {
        if (ctx->d_sum) (void)hipFree(ctx->d_sum);
        ctx->d_sum = nullptr; ctx->sum_cap = 0;
        hipError_t e = hipMalloc(&ctx->d_sum, total);
        if (e != hipSuccess) return e;
        ctx->sum_cap = total;
    }
    ctx->sum_sort_bytes = sort_b;
    char* p = (char*)ctx->d_sum;
    w.sort_tmp = p; p += sort_b;
    w.work = p; p += work_b;
    w.rank = (int*)p; p += rank_b;
    w.out = (double*)p; p += 200 * 8;
    w.counters = (int*)p;
    return hipSuccess;
}
}  // namespace

int dsq_dev_padj_prepare(dsq_ctx* ctx, const double* d_base_mean, const double* d_pvalue, int n, double alpha,
                         unsigned long long* d_sorted_p, int32_t* d_sorted_idx, uint8_t* d_bins,
                         double* h_out200, int* h_n_valid) {
    DSQ_CHECK_ARG(n >= 1, "no genes");
    SumWs w;
    DSQ_HIP(sum_workspace(ctx, n, w));
    DSQ_HIP(dsq::launch_padj_prepare(ctx->stream, d_base_mean, d_pvalue, n, alpha, w.sort_tmp, ctx->sum_sort_bytes,
                                     w.work, d_sorted_p, d_sorted_idx, d_bins, w.out, w.counters));
    int cnt[4] = {0, 0, 0, 0};
    DSQ_HIP(hipMemcpyAsync(cnt, w.counters, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    DSQ_HIP(dsq::launch_padj_numrej(ctx->stream, d_sorted_p, d_sorted_idx, d_bins, cnt[1], alpha, w.out));
    DSQ_HIP(hipMemcpyAsync(h_out200, w.out, 200 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    *h_n_valid = cnt[1];
    return DSQ_OK;
}

int dsq_dev_padj_finish(dsq_ctx* ctx, const unsigned long long* d_sorted_p, const int32_t* d_sorted_idx,
                        const uint8_t* d_bins, int n, int n_valid, int j, double* d_padj) {
    DSQ_CHECK_ARG(n >= 1 && n_valid >= 0 && n_valid <= n && j >= -1 && j < 50, "bad pass / sizes");
    SumWs w;
    DSQ_HIP(sum_workspace(ctx, n, w));
    DSQ_HIP(dsq::launch_padj_finish(ctx->stream, d_sorted_p, d_sorted_idx, d_bins, n, n_valid, j, w.rank, d_padj));
    return DSQ_OK;
}


int dsq_dev_gather_rows_i32(dsq_ctx* ctx, const int32_t* d_src, int ld, const int32_t* d_idx, int n_idx,
                            int ncols, int32_t* d_dst) {
    DSQ_HIP(dsq::launch_gather_rows_i32(ctx->stream, d_src, ld, d_idx, n_idx, ncols, d_dst));
    return DSQ_OK;
}

int dsq_dev_trend_fit(dsq_ctx* ctx, const double* d_disp, const double* d_means, int n, double min_disp,
                      double max_disp, uint8_t* d_keep, double* h_coeffs2, int* h_ok, int* h_n_outer) {
    double* d_out = ctx->d_scratch + 1536;
    if (ctx->d_trend_grid == nullptr) DSQ_HIP(hipMalloc(&ctx->d_trend_grid, dsq::trend_grid_mem_bytes()));
    static const int force_grid = getenv("DSQ_TREND_GRID") ? atoi(getenv("DSQ_TREND_GRID")) : 0;
    DSQ_HIP(dsq::launch_trend_fit(ctx->stream, d_disp, d_means, n, min_disp, max_disp, d_keep, d_out,
                                  ctx->d_trend_grid, force_grid));
    double out5[5];
    DSQ_HIP(hipMemcpyAsync(out5, d_out, sizeof(out5), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (out5[2] < 0.0) return fail(ctx, DSQ_ERR_HIP, "trend fit: the workgroups' exchange timed out");
    h_coeffs2[0] = out5[0]; h_coeffs2[1] = out5[1];
    if (h_ok) *h_ok = (int)out5[2];
    if (h_n_outer) *h_n_outer = (int)out5[3];
    return DSQ_OK;
}

// Parametric trend, its fitted values and the MAD prior in one call with ONE host synchronisation: the fitted values
// are evaluated from the coefficients the trend kernel left on the device, the prior kernel follows, and the five
// scalars come back together through page-locked memory.  *h_ok = 0 (the fit did not converge, dds.py:811-823): the
// caller falls back to the mean trend with dsq_dev_trend_eval + dsq_dev_prior_mad; d_fitted / *h_squared_logres are
// then meaningless.
int dsq_dev_trend_prior(dsq_ctx* ctx, const double* d_disp, const double* d_means, int n, double min_disp,
                        double max_disp, uint8_t* d_keep, double* d_fitted, double* d_work, double* h_coeffs2,
                        int* h_ok, int* h_n_outer, double* h_squared_logres) {
    double* d_out = ctx->d_scratch + 1536;  // {c0, c1, ok, n_outer, -}  then {squared_logres, status} at + 64
    double* d_out2 = ctx->d_scratch + 1600;
    if (ctx->d_trend_grid == nullptr) DSQ_HIP(hipMalloc(&ctx->d_trend_grid, dsq::trend_grid_mem_bytes()));
    static const int force_grid = getenv("DSQ_TREND_GRID") ? atoi(getenv("DSQ_TREND_GRID")) : 0;
    // with a CU split (dsq_side_begin) the three kernels run on the stream that owns the reserved compute units
    hipStream_t st = ctx->stream;
    if (ctx->small_stream != nullptr && ctx->stream == ctx->main_stream) {
        DSQ_HIP(hipEventRecord(ctx->ev_small0, ctx->stream));
        DSQ_HIP(hipStreamWaitEvent(ctx->small_stream, ctx->ev_small0, 0));
        st = ctx->small_stream;
    }
    DSQ_HIP(dsq::launch_trend_fit(st, d_disp, d_means, n, min_disp, max_disp, d_keep, d_out,
                                  ctx->d_trend_grid, force_grid));
    DSQ_HIP(dsq::launch_trend_eval_dev(st, d_means, n, d_out, d_fitted));
    DSQ_HIP(dsq::launch_prior_mad(st, d_disp, d_fitted, n, min_disp, max_disp, d_work, d_out2));
    double* h = (double*)(ctx->h_pin + 3072);  // 12 KiB into the page-locked block (behind the ridge / contrast staging)
    DSQ_HIP(hipMemcpyAsync(h, d_out, 5 * sizeof(double), hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(h + 8, d_out2, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (st != ctx->stream) {
        DSQ_HIP(hipEventRecord(ctx->ev_small1, st));
        DSQ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_small1, 0));
    }
    DSQ_HIP(hipStreamSynchronize(st));
    if (h[2] < 0.0) return fail(ctx, DSQ_ERR_HIP, "trend fit: the workgroups' exchange timed out");
    h_coeffs2[0] = h[0]; h_coeffs2[1] = h[1];
    *h_ok = (int)h[2];
    if (h_n_outer) *h_n_outer = (int)h[3];
    *h_squared_logres = h[8];
    return DSQ_OK;
}

size_t dsq_prior_mad_work_doubles(int n) { return dsq::prior_mad_work_doubles(n); }

int dsq_dev_prior_mad(dsq_ctx* ctx, const double* d_gw_raw, const double* d_fitted, int n, double min_disp,
                      double max_disp, double* d_work, double* h_squared_logres) {
    double* d_out = ctx->d_scratch + 1600;
    DSQ_HIP(dsq::launch_prior_mad(ctx->stream, d_gw_raw, d_fitted, n, min_disp, max_disp, d_work, d_out));
    double out2[2];
    DSQ_HIP(hipMemcpyAsync(out2, d_out, sizeof(out2), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    *h_squared_logres = out2[0];
    return DSQ_OK;
}

int dsq_dev_trend_loss_grad(dsq_ctx* ctx, const double* d_cov, const double* d_targets, const uint8_t* d_keep,
                            int n, double a0, double a1, double* loss, double* grad2) {
    double* d_part = ctx->d_scratch + 256;  // kTrendPartials x 4 doubles = 8 KiB
    DSQ_HIP(dsq::launch_trend_loss_grad(ctx->stream, d_cov, d_targets, d_keep, n, a0, a1, d_part));
    std::vector<double> part((size_t)dsq::kTrendPartials * 4);
    DSQ_HIP(hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < dsq::kTrendPartials; ++b)
        for (int k = 0; k < 4; ++k) s[k] += part[(size_t)b * 4 + k];
    const double cnt = s[3];
    *loss = s[0] / cnt;
    grad2[0] = -s[1] / cnt;
    grad2[1] = -s[2] / cnt;
    return DSQ_OK;
}


}  // extern "C"
