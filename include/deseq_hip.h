/* deseq_hip.h — C ABI of libdeseq_hip.so, the MI355X (gfx950) engine for the
 * PyDESeq2 `DeseqDataSet.deseq2()` -> Wald hot path.
 *
 * Boundary.  PyDESeq2's plug-in point for this path is the abstract class
 * `pydeseq2.inference.Inference` (pydeseq2/inference.py:9-362); its default
 * implementation is `DefaultInference` (pydeseq2/default_inference.py:14-264), which
 * fans each method out to one joblib task per gene.  Every `dsq_inf_*` function below
 * replaces exactly one of those methods (cited per function) with host pointers in and
 * out, so a binding is a thin ctypes/cffi stub.  The `dsq_dev_*`
 * functions expose the same stages on device-resident buffers for a fused pipeline that
 * keeps counts / mu / hat-diagonals in HBM between stages (pydeseq2_amd/pipeline.py).
 *
 * Conventions
 *   - plain C, no exceptions: every function returns 0 on success, a negative dsq_status
 *     on failure; dsq_last_error(ctx) returns a message for the last failure.
 *   - matrices are described by (pointer, layout): DSQ_SAMPLE_MAJOR is the reference's
 *     N x G C-order array (element (n,g) at n*G+g); DSQ_GENE_MAJOR is G x N C-order
 *     (= the F-ordered N x G arrays the reference gets from `mu_hat_.T`,
 *     default_inference.py:81,119-124).  Device-resident gene-major buffers have a row
 *     pitch `ldn` (elements), a multiple of 16.
 *   - counts are non-negative integers < 2^31 (int32 or int64 storage); everything else
 *     is IEEE double.  Natural-log fold changes, dispersion alpha with var = mu + alpha mu^2.
 *   - N = samples, G = genes, P = design columns (1..DSQ_MAX_P = 48).
 *   - statistical non-convergence is NOT an error: it is reported in the `converged`
 *     arrays exactly like the reference does.
 */
#ifndef DESEQ_HIP_H
#define DESEQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSQ_ABI_VERSION 5   /* bumped whenever an exported signature changes; dsq_abi_version() returns the library's */
#define DSQ_MAX_P 48        /* design columns; up to 12 run the register / cell kernels, wider ones the LDS + MFMA path (round 6: 32 -> 48) */
#define DSQ_SHRINK_MAX_P 48 /* apeGLM shrinkage (dsq_*_lfc_shrink*): up to 12 columns in registers, 13 ... 48 run-time p */
#define DSQ_BFGS_MAX_P 12   /* optimizer = "BFGS" of the dispersion fit / the IRLS rescue: register kernels only */

typedef struct dsq_ctx dsq_ctx;

enum dsq_status {
    DSQ_OK = 0,
    DSQ_ERR_HIP = -1,       /* a HIP runtime call failed (message has the HIP error string) */
    DSQ_ERR_ARG = -2,       /* bad argument (P out of range, N == P, null pointer, ...) */
    DSQ_ERR_RANGE = -3,     /* a count does not fit int32 or is negative */
    DSQ_ERR_NOMEM = -4
};

enum dsq_layout { DSQ_SAMPLE_MAJOR = 0, DSQ_GENE_MAJOR = 1 };
/* alpha-independent constant of a gene's NLL, sum lgamma(y+1) - y log(mu) (utils.py:227-234): the
 * genewise (MLE) and MAP fits of one deseq2() run see the same counts and mu_hat, so the second
 * launch can re-use what the first one stored (8 B per gene) instead of recomputing it. */
enum dsq_const_mode { DSQ_CONST_COMPUTE = 0, DSQ_CONST_STORE = 1, DSQ_CONST_LOAD = 2 };
enum dsq_count_type { DSQ_I32 = 0, DSQ_I64 = 1 };
/* alternative hypotheses of utils.wald_test (pydeseq2/utils.py:778-806) */
enum dsq_alt { DSQ_ALT_NONE = 0, DSQ_ALT_GREATER_ABS = 1, DSQ_ALT_LESS_ABS = 2,
               DSQ_ALT_GREATER = 3, DSQ_ALT_LESS = 4 };

/* ------------------------------------------------------------------ context */
int dsq_create(int device_id, dsq_ctx** out);
/* Deferred second passes (sticky until set again).  The dispersion fit's grid-search pass (utils.py:556-564) and the
 * IRLS rescue (utils.py:374-413) normally wait for the host to read how many genes need them.  With on != 0, batches of
 * at most 2048 genes on the register kernels (P <= 12) enqueue those passes for every gene of the batch as a capacity
 * and the kernels read the count from device memory: no host synchronisation inside dsq_dev_alpha_mle* /
 * dsq_dev_lfc_fit / dsq_dev_irls (dsq_last_alpha_kernel then reports -1 ms / -1 genes).  Results are identical. */
int dsq_set_deferred(dsq_ctx* ctx, int on);
/* One-shot hook of the NEXT dispersion fit on this context (dsq_dev_alpha_mle*): fn(arg) is called once, on the calling
 * thread, when the full-size kernels of that fit have been enqueued and only its latency-bound tail remains (the
 * continuation of the few long fits, the grid-search pass of utils.py:556-564) - the moment to put independent work
 * on the side stream (dsq_side_begin ... dsq_side_end; the pipeline launches the robust dispersions of the Cook's
 * stage, utils.py:914-960, there).  fn may call any entry point of this library except another dispersion fit.
 * fn == NULL clears a hook that has not fired. */
typedef void (*dsq_hook_fn)(void* arg);
int dsq_set_alpha_hook(dsq_ctx* ctx, dsq_hook_fn fn, void* arg);
/* Performance hint for the NEXT dsq_dev_lfc_fit / dsq_dev_irls of G genes on this context (one-shot): d_iters[G] are
 * the iteration counts an earlier IRLS fit of the same genes returned (irls_solver's loop counter, utils.py:361-421;
 * the mu_hat fit of dds.py:757-765 before the LFC fit of dds.py:908-984).  Designs fitted with sixteen lanes per gene
 * place genes with equal counts in the same wavefront; without a hint they are grouped by dispersion.  Results do not
 * depend on it.  The array must stay valid until that call has been enqueued. */
int dsq_irls_order_hint(dsq_ctx* ctx, const int32_t* d_iters, int G);
void dsq_destroy(dsq_ctx* ctx);
const char* dsq_last_error(const dsq_ctx* ctx);
/* name (<= name_len bytes), compute units, total device memory, gcnArchName */
int dsq_device_info(dsq_ctx* ctx, char* name, int name_len, int* cu_count, size_t* mem_bytes,
                    char* arch, int arch_len);
int dsq_sync(dsq_ctx* ctx);
/* developer aid: text of a pending (unconsumed) HIP error of the calling thread, "" if none */
const char* dsq_debug_pending_error(void);
/* HIP-event stopwatch on the context's stream (used by bench.py for kernel timing) */
int dsq_timer_start(dsq_ctx* ctx);
int dsq_timer_stop(dsq_ctx* ctx, float* elapsed_ms);
/* duration of the k_alpha launch inside the last *_alpha_mle call (HIP events on the launch
 * stream) and the number of genes that needed the grid-search fallback */
int dsq_last_alpha_kernel(dsq_ctx* ctx, float* kernel_ms, int* n_grid_fallback);

/* ------------------------------------------------------------------ device memory */
int dsq_malloc(dsq_ctx* ctx, size_t bytes, void** dptr);
int dsq_free(dsq_ctx* ctx, void* dptr);
int dsq_memset(dsq_ctx* ctx, void* dptr, int value, size_t bytes);
int dsq_h2d(dsq_ctx* ctx, void* dst, const void* src, size_t bytes);
int dsq_d2h(dsq_ctx* ctx, void* dst, const void* src, size_t bytes);
/* pitched copies: `rows` rows of `row_bytes`, pitches in bytes */
int dsq_h2d_2d(dsq_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch,
               size_t row_bytes, size_t rows);
int dsq_d2h_2d(dsq_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch,
               size_t row_bytes, size_t rows);

/* ================================================================== Inference-level API
 * Host pointers in, host pointers out; results depend on the arguments only (safe for the re-entrant
 * use by DeseqDataSet._refit_without_outliers, dds.py:1392-1408).
 * design[N*P] is the row-major design matrix (obsm["design_matrix"].values).
 *
 * The reference calls its plug-in 7-9 times per deseq2() with the same matrices, each time as a fresh host copy
 * (dds.py:747-785, 901-911, 953-960, 1149-1157; ds.py:338-350).  Behind these entry points every N x G host matrix is
 * identified by an exact 128-bit content digest (every element takes part; host pointers are never trusted, a matrix
 * mutated in place is a different matrix) and its gene-major device copy stays resident per context; mu_hat / mu
 * matrices an entry point returns stay resident under the digest of what went back to the host, so the call that hands
 * them back (alpha_mle, wald_test) uploads nothing.  LRU over a byte budget (default: a quarter of the device memory;
 * environment DSQ_PLUGIN_CACHE_MB, DSQ_PLUGIN_CACHE=0 switches the cache off, DSQ_HASH_THREADS the digest's host
 * threads).  csrc/dsq_plugin_cache.h.                                                                          */
int dsq_abi_version(void);   /* DSQ_ABI_VERSION of the loaded library */
/* enabled / budget_bytes: < 0 leaves the setting as it is */
int dsq_plugin_cache_config(dsq_ctx* ctx, int enabled, long long budget_bytes);
int dsq_plugin_cache_clear(dsq_ctx* ctx);   /* frees every resident matrix and pooled buffer */
/* out[0..n): hits, misses, adopted (resident outputs), evictions, bytes uploaded, bytes downloaded (N x G layers), host
 * milliseconds spent in digests, resident bytes, pooled free bytes, resident matrices, hipMalloc calls, budget bytes,
 * hits that were re-uploaded and compared byte for byte with the resident copy (environment DSQ_PLUGIN_CACHE_VERIFY: a
 * mismatch - a digest collision - fails the call with DSQ_ERR_ARG) */
int dsq_plugin_cache_stats(dsq_ctx* ctx, double* out, int n);
/* The digest itself (needs no context / GPU): elem_type 0 int32, 1 int64 (counts: digest of the VALUES, so both types of
 * the same matrix agree), 2 double (bit patterns); layout as dsq_layout; out2 = the two 64-bit halves.  Independent of
 * layout and thread count by construction. */
int dsq_plugin_digest_host(const void* data, int elem_type, int layout, int N, int G, int n_threads,
                           unsigned long long* out2);

/* Inference.lin_reg_mu (inference.py:13-44; DefaultInference.lin_reg_mu
 * default_inference.py:58-81 -> utils.fit_lin_mu utils.py:682-715).
 * mu_out: G x N gene-major (caller returns its transpose view). */
int dsq_inf_lin_reg_mu(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                       const double* size_factors, const double* design, int N, int G, int P,
                       double min_mu, double* mu_out);

/* Inference.irls (inference.py:46-119; default_inference.py:83-124 -> utils.irls_solver
 * utils.py:273-438).  beta_out[G*P]; mu_out, hat_out: G x N gene-major; converged[G].
 * optimizer (utils.py:343, the rescue of diverged genes utils.py:389-399): 0 = "L-BFGS-B" (bounded; the default and the
 * only one dds.py / ds.py pass), 1 = "BFGS" (scipy's unbounded BFGS restated, csrc/dsq_bfgs.h; at most 12 columns). */
int dsq_inf_irls2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                  const double* size_factors, const double* design, const double* disp, int N,
                  int G, int P, double min_mu, double beta_tol, double min_beta, double max_beta,
                  int maxiter, double* beta_out, double* mu_out, double* hat_out,
                  uint8_t* converged, int optimizer);
/* (the signature of ABI versions < 5, without the trailing optimizer: L-BFGS-B) */
int dsq_inf_irls(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                 const double* size_factors, const double* design, const double* disp, int N,
                 int G, int P, double min_mu, double beta_tol, double min_beta, double max_beta,
                 int maxiter, double* beta_out, double* mu_out, double* hat_out,
                 uint8_t* converged);

/* Inference.alpha_mle (inference.py:121-178; default_inference.py:126-161 ->
 * utils.fit_alpha_mle utils.py:441-564, grid_search.grid_fit_alpha grid_search.py:54-142).
 * mu given in `mu_layout`; prior_disp_var ignored unless prior_reg; optimizer (utils.py:546-554): 0 = "L-BFGS-B",
 * 1 = "BFGS" (at most 12 design columns). */
int dsq_inf_alpha_mle2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                       const double* design, const double* mu, int mu_layout,
                       const double* alpha_hat, int N, int G, int P, double min_disp,
                       double max_disp, double prior_disp_var, int cr_reg, int prior_reg,
                       double* alpha_out, uint8_t* converged, int optimizer);
/* (the signature of ABI versions < 5, without the trailing optimizer: L-BFGS-B) */
int dsq_inf_alpha_mle(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                      const double* design, const double* mu, int mu_layout,
                      const double* alpha_hat, int N, int G, int P, double min_disp,
                      double max_disp, double prior_disp_var, int cr_reg, int prior_reg,
                      double* alpha_out, uint8_t* converged);

/* Inference.wald_test (inference.py:180-235; default_inference.py:163-198 ->
 * utils.wald_test utils.py:718-811).  lfc[G*P] natural log; ridge[P*P]; contrast[P];
 * lfc_null already multiplied by ln 2 by the caller (ds.py:345). */
int dsq_inf_wald_test(dsq_ctx* ctx, const double* design, const double* disp, const double* lfc,
                      const double* mu, int mu_layout, const double* ridge,
                      const double* contrast, double lfc_null, int alt, int N, int G, int P,
                      double* pvals, double* stats, double* se);

/* Inference.fit_rough_dispersions (inference.py:237-259; utils.py:814-853) and
 * Inference.fit_moments_dispersions (inference.py:261-282; utils.py:856-885) on
 * normalised counts (double, `layout`).  Returns DSQ_ERR_ARG when N == P (the reference
 * raises ValueError, utils.py:839-844). */
int dsq_inf_fit_rough_dispersions(dsq_ctx* ctx, const double* normed, int layout,
                                  const double* design, int N, int G, int P, double* alpha_out);
int dsq_inf_fit_moments_dispersions(dsq_ctx* ctx, const double* normed, int layout,
                                    const double* size_factors, int N, int G, double* alpha_out);
/* ... that also reports all_zero[G] (1: every normalised count of the gene is zero): the reference drops those columns
 * before taking the moments (utils.py:878), i.e. the caller drops the same entries of alpha_out */
int dsq_inf_fit_moments_dispersions2(dsq_ctx* ctx, const double* normed, int layout, const double* size_factors,
                                     int N, int G, double* alpha_out, uint8_t* all_zero);

/* Inference.dispersion_trend_gamma_glm (inference.py:284-308; default_inference.py:200-230): ONE gamma-GLM
 * fit targets ~ a0 + a1 * covariates (L-BFGS-B from (1, 1), lower bound 1e-12, scipy defaults; NaN entries
 * are skipped as numpy.nanmean does) run on the device.  coeffs2 = (a0, a1) (intercept first),
 * predictions[n] = a0 + a1 * covariates (may be NULL), *converged = scipy's res.success. */
int dsq_inf_dispersion_trend_gamma_glm(dsq_ctx* ctx, const double* covariates, const double* targets, int n,
                                       double* coeffs2, double* predictions, int* converged);

/* grid_search.grid_fit_alpha (grid_search.py:54-142) — what utils.fit_alpha_mle falls back to when its
 * L-BFGS-B run reports success == False (utils.py:556-564) — for EVERY gene of the batch, on the production
 * kernels of that fallback (100 wavefronts per gene and grid level).  Returns log(alpha) like the reference
 * (Cox-Reid term on, no prior: the reference passes six positional arguments only). */
int dsq_inf_grid_fit_alpha(dsq_ctx* ctx, const void* counts, int count_type, int count_layout, const double* design,
                           const double* mu, int mu_layout, int N, int G, int P, double min_disp, double max_disp,
                           double* log_alpha_out);
/* grid_search.grid_fit_beta (grid_search.py:145-221) — the P == 2 fallback of utils.irls_solver when IRLS
 * diverged and the bounded L-BFGS-B rescue failed too (utils.py:404-411): two-level grid_length x grid_length
 * search on [min_beta, max_beta]^2.  design: N x 2.  beta_out[G*2]. */
int dsq_inf_grid_fit_beta(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                          const double* size_factors, const double* design, const double* disp, int N, int G,
                          double min_mu, int grid_length, double min_beta, double max_beta, double* beta_out);

/* Loss and gradient of the 2-coefficient gamma GLM of
 * Inference.dispersion_trend_gamma_glm (inference.py:284-308; default_inference.py:200-230):
 *   loss = mean(t/m + log m), m = a0 + a1*cov ; grad as default_inference.py:213-217.
 * cov/targets are device pointers (n entries); used by the host L-BFGS-B driver. */
int dsq_dev_trend_loss_grad(dsq_ctx* ctx, const double* d_cov, const double* d_targets,
                            const uint8_t* d_keep, int n, double a0, double a1, double* loss,
                            double* grad2);

/* DeseqDataSet._fit_parametric_dispersion_trend (dds.py:1199-1275): the whole iterated
 * gamma-GLM trend fit (L-BFGS-B, default_inference.py:200-230) in one single-wavefront launch.
 * d_disp: raw genewise dispersions of the n non-zero genes (clipped to [min,max] on the fly),
 * d_means: their normalised means, d_keep: n bytes of device scratch.  *h_ok == 0: the fit did
 * not converge -> caller switches to the mean trend (dds.py:1243-1252). */
int dsq_dev_trend_fit(dsq_ctx* ctx, const double* d_disp, const double* d_means, int n,
                      double min_disp, double max_disp, uint8_t* d_keep, double* h_coeffs2,
                      int* h_ok, int* h_n_outer);

/* DeseqDataSet.fit_dispersion_prior (dds.py:866-879) + utils.mean_absolute_deviation
 * (utils.py:1210-1227): squared_logres = (MAD(log genewise - log fitted) / norm.ppf(0.75))^2 over
 * the genes with genewise >= 100*min_disp; exact medians by radix select in one workgroup.
 * d_gw_raw: raw genewise dispersions (clipped on the fly), d_fitted: trend values, n genes,
 * d_work: dsq_prior_mad_work_doubles(n) doubles of device scratch (n residuals; for n >= 32768 the medians
 * run as multi-workgroup radix passes — one cooperative launch with a grid barrier per pass — whose
 * global state follows the residuals). */
size_t dsq_prior_mad_work_doubles(int n);
int dsq_dev_prior_mad(dsq_ctx* ctx, const double* d_gw_raw, const double* d_fitted, int n,
                      double min_disp, double max_disp, double* d_work, double* h_squared_logres);

/* ================================================================== device-resident stage API
 * All pointers are device pointers unless named h_*.  Gene-major rows have pitch ldn.
 * Xt and pinvXt are [P][ldx] (design transposed; rows of (X^T X)^-1 X^T). */

/* counts (host layout as uploaded) -> int32 gene-major [G][ldn]; *h_bad set to 1 if a value
 * is negative or >= 2^31 */
int dsq_dev_counts_to_gene_major(dsq_ctx* ctx, const void* d_src, int count_type, int layout,
                                 int N, int G, int32_t* d_dst, int ldn, int* h_bad);
int dsq_dev_f64_to_gene_major(dsq_ctx* ctx, const double* d_src, int layout, int N, int G,
                              double* d_dst, int ldn);

/* preprocessing.deseq2_norm_fit (preprocessing.py:31-56): logmeans[G] (-inf if any zero),
 * nonzero[G] = any(count > 0) (dds.py:729) */
int dsq_dev_logmeans(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, double* d_logmeans,
                     uint8_t* d_nonzero);
/* preprocessing.deseq2_norm_transform (preprocessing.py:59-102): per-sample median over the
 * genes with finite logmeans (and d_gene_mask[g] != 0 if given) of log(count) - logmeans.
 * d_counts_sm: sample-major counts [N][G] of `count_type`; d_work: dsq_size_factors_work_doubles(N, G)
 * doubles of scratch (order-preserving keys of the usable genes only + their index list). */
size_t dsq_size_factors_work_doubles(int N, int G);
int dsq_dev_size_factors(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                         const double* d_logmeans, const uint8_t* d_gene_mask, double* d_work,
                         double* d_size_factors);
/* The same for NEW samples normalised with the training log means (DeseqDataSet.vst_transform on new counts,
 * dds.py:471-484 -> deseq2_norm_transform): a zero count in a usable gene is log(0) - logmean = -inf and takes
 * part in the sample's median at the low end, as in numpy. */
int dsq_dev_size_factors_new(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                             const double* d_logmeans, const uint8_t* d_gene_mask, double* d_work,
                             double* d_size_factors);
/* dsq_dev_mom + dsq_dev_lin_mu fused for the designs that take the linear-model mu_hat (#cells == p,
 * dds.py:747-756): two sweeps over a gene's counts instead of four, bit-identical outputs. */
int dsq_dev_mom_lin_mu(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                       const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                       double min_mu, double* d_normed_mean, double* d_mom, double* d_mu);
/* utils.fit_rough_dispersions + fit_moments_dispersions + dds.py:1157-1162 and
 * var["_normed_means"] (dds.py:708) */
int dsq_dev_mom(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp,
                double max_disp, double* d_normed_mean, double* d_rough, double* d_moments,
                double* d_mom);
int dsq_dev_lin_mu(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf,
                   const double* d_Xt, const double* d_pinvXt, int ldx, int N, int G, int P,
                   double min_mu, double* d_mu);
/* d_nfev may be null.  alpha is NOT clipped (the caller clips, dds.py:792-794).
 * d_nll_const [G] (may be null -> DSQ_CONST_COMPUTE) with const_mode: see dsq_const_mode. */
int dsq_dev_alpha_mle(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn,
                      const double* d_Xt, int ldx, int N, int G, int P, const double* d_alpha_hat,
                      double min_disp, double max_disp, double prior_disp_var, int cr_reg,
                      int prior_reg, double* d_alpha, uint8_t* d_converged, int32_t* d_nfev,
                      double* d_nll_const, int const_mode);
/* Designs whose rows take few distinct values (every purely categorical design): the design's "cells".
 * d_cell_of[ldx]: cell of every sample (0 beyond N); d_Xc[n_cells][P]: the cells' design rows;
 * d_XX[n_cells][P(P+1)/2]: products x_i x_j of a cell's row (packed lower triangle, entry i(i+1)/2 + j).
 * With n_cells <= 64 the dispersion and IRLS kernels accumulate per-cell weight sums instead of per-sample
 * outer products: their per-sample work no longer depends on P (used for P >= 3).  NULL / n_cells == 0: general path. */
typedef struct dsq_cells {
    const int32_t* d_cell_of;
    const double* d_Xc;
    const double* d_XX;
    int n_cells;
} dsq_cells;
/* dsq_dev_mom_lin_mu that also stores the OLS coefficients of the normalised counts, d_coef[G][P]; d_mu may then be
 * NULL: mu_hat = max(sf * (X coef), min_mu) is rebuilt inside the dispersion kernel (dsq_dev_alpha_mle2) instead of
 * being written once (8 N bytes per gene) and read twice. */
int dsq_dev_mom_lin_coef(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                         const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                         double min_mu, double* d_normed_mean, double* d_mom, double* d_mu, double* d_coef);
/* dsq_dev_alpha_mle with the optional cell path (cells) and, when d_mu == NULL, mu_hat from (d_coef, d_sf, min_mu). */
int dsq_dev_alpha_mle2(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx,
                       int N, int G, int P, const double* d_alpha_hat, double min_disp, double max_disp,
                       double prior_disp_var, int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_converged,
                       int32_t* d_nfev, double* d_nll_const, int const_mode, const dsq_cells* cells,
                       const double* d_coef, const double* d_sf, double min_mu);
/* dsq_dev_alpha_mle2 with the genes of the call partitioned into two lists (device int32, indices into the G genes):
 * those of d_rows run FOUR to a wavefront (csrc/dsq_k_alpha_rows.hip: linear-model mu_hat designs with at most 4
 * cells, see dsq_alpha_rows_eligible), those of d_waves one to a wavefront.  Which gene may go where depends on its
 * counts only (dsq_dev_alpha_row_split: -1 = keep it on d_waves, else the number of its samples with a count >= 512,
 * which cost the row kernel a second sweep per evaluation: queue such genes together), so the lists are built once per
 * data set.  The row kernel takes its genes in the order of d_rows.
 * d_rows == NULL: as dsq_dev_alpha_mle2.  Results do not depend on the partition beyond rounding.
 * d_cell_mu != NULL (with d_mu == d_coef == NULL): mu_hat in per-cell form, see dsq_alpha_rows_eligible. */
int dsq_dev_alpha_mle3(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx,
                       int N, int G, int P, const double* d_alpha_hat, double min_disp, double max_disp,
                       double prior_disp_var, int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_converged,
                       int32_t* d_nfev, double* d_nll_const, int const_mode, const dsq_cells* cells,
                       const double* d_coef, const double* d_sf, double min_mu, const int32_t* d_rows, int n_rows,
                       const int32_t* d_waves, int n_waves, const double* d_cell_mu);
/* MIXED designs: a few categorical columns (at most 32 distinct rows: "cells") + one to three continuous covariates,
 * at most 8 columns (BASELINE configs[4]; csrc/dsq_mix.h, dsq_k_alpha_mix.hip).  The reference treats every design alike
 * (utils.py:441-564, 273-438: N outer products x_n x_n^T per evaluation); here X^T W X splits into a per-cell block, a
 * per-cell x covariate block and a small covariate block, and the samples are walked sorted by cell.
 * dsq_mix_create analyses a design (host, row-major N x P) once and keeps its slot order, cell table and sorted
 * covariates on the device; *out = NULL with DSQ_OK when the design is not one these kernels take (no continuous column
 * needed, too many covariates / columns / cells, mostly padding, rows too long) - the caller then stays on the other
 * paths.  Environment: DSQ_NO_ALPHA_MIX=1 disables the family, DSQ_MIX_FORCE=1 lifts the padding limit (tests). */
typedef struct dsq_mix dsq_mix;
int dsq_mix_create(dsq_ctx* ctx, const double* design, int N, int P, dsq_mix** out);
void dsq_mix_destroy(dsq_mix* mix);
int dsq_mix_info(const dsq_mix* mix, int* n_slots, int* n_cells, int* n_continuous);
/* h_slot_of[N]: the slot of every sample (the kernels walk the samples sorted by design cell, cells padded to whole
 * loop iterations; a slot-ordered N x G layer - see dsq_dev_lfc_fit2 - is read through this map) */
int dsq_mix_slots(const dsq_mix* mix, int32_t* h_slot_of);
/* 1 when dsq_dev_lfc_fit2 runs such a design on k_irls_mix (and may therefore be given a slot-ordered Cook's layer), else
 * 0: the library's own routing rule */
/* Slot-ordered copies for the mixed-design kernels (round 5): the counts of a gene-major matrix as uint16 [G][Ns]
 * (Ns from dsq_mix_info; 0xFFFF in padding slots; counts >= 65 534 saturate and set d_big[g]) and the IRLS route's mu_hat
 * sf * exp(X beta) (dds.py:757-771, UNclamped) as fp64 [G][Ns] (0 in padding slots).  dsq_mix_bind hands them to the NEXT
 * dispersion fit (dsq_dev_alpha_mle*) or IRLS fit (dsq_dev_lfc_fit*, dsq_dev_irls) of the context - one-shot; a fit of a
 * mixed design that finds nothing bound builds its own copies per call. */
int dsq_mix_bind(dsq_ctx* ctx, const uint16_t* d_ys, const uint8_t* d_big, const double* d_mu_slots);
/* The same with the gene count the copies were built for: a consuming fit whose G differs ignores the binding and builds its
 * own copies (dsq_mix_bind leaves the count unknown: no check). */
int dsq_mix_bind2(dsq_ctx* ctx, const uint16_t* d_ys, const uint8_t* d_big, const double* d_mu_slots, int G);
int dsq_dev_mix_counts_to_slots(dsq_ctx* ctx, const int32_t* d_y, int ldn, int G, const dsq_mix* mix, uint16_t* d_ys,
                                uint8_t* d_big);
int dsq_dev_mix_mu_slots(dsq_ctx* ctx, const dsq_mix* mix, const double* d_beta, const double* d_sf, int G,
                         double* d_mu_slots);
int dsq_mix_takes_irls(const dsq_mix* mix, int full_rank);
/* diagnostics: launches of the mixed-design dispersion kernel by this process so far */
int dsq_mix_launch_count(void);
/* dsq_dev_alpha_mle3 with a mixed design: the genes of d_rows run k_alpha_mix (one gene per wavefront, counts staged as
 * uint16: dsq_dev_alpha_row_split says which genes fit), those of d_waves the general kernel (they need d_mu).
 * d_beta != NULL (with d_mu == d_coef == d_cell_mu == NULL, n_waves == 0): mu_hat = d_sf * exp(X d_beta), the UNclamped
 * mean of the IRLS mu_hat route (dds.py:757-771, utils.py:435-437), rebuilt inside the kernel from the [G][P]
 * coefficients - the N x G matrix is neither written nor read.  d_beta == NULL: mu_hat gathered from d_mu.  mix == NULL:
 * as dsq_dev_alpha_mle3. */
int dsq_dev_alpha_mle4(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx,
                       int N, int G, int P, const double* d_alpha_hat, double min_disp, double max_disp,
                       double prior_disp_var, int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_converged,
                       int32_t* d_nfev, double* d_nll_const, int const_mode, const dsq_cells* cells,
                       const double* d_coef, const double* d_sf, double min_mu, const int32_t* d_rows, int n_rows,
                       const int32_t* d_waves, int n_waves, const double* d_cell_mu, const dsq_mix* mix,
                       const double* d_beta);
/* 0: no row kernel for such a design; 1: at most 4 cells == columns (linear-model mu_hat); 2: up to 32 cells, p <= 8
 * (csrc/dsq_k_alpha_rowsc.hip; both mu_hat routes - the IRLS route hands over d_cell_mu, [G][n_cells] = exp(x_c . beta) from
 * dsq_dev_cell_mu, and mu_hat_n = sf_n * cell_mu[cell_of[n]] unclamped (utils.py:435-437) is never materialised). */
int dsq_alpha_rows_eligible(int N, int P, int n_cells);
/* 1 when dsq_dev_alpha_mle* needs mu_hat as a matrix for such a design (run-time-P kernels, or rows too long for the staged
 * kernel to rebuild it from d_coef / d_cell_mu), else 0: the library's own routing rule, so that callers do not mirror it. */
int dsq_alpha_needs_mu(int N, int P, int n_cells);
int dsq_dev_cell_mu(dsq_ctx* ctx, const double* d_beta, const dsq_cells* cells, int G, int P, double* d_cell_mu);
int dsq_dev_alpha_row_split(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, int32_t* d_flags);
/* The robust dispersion of utils.robust_method_of_moments_disp (utils.py:914-960) alone: the half of dsq_dev_cooks
 * that depends on counts, size factors and design cells only (arguments as dsq_dev_cooks). */
int dsq_dev_robust_disp(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const int32_t* d_cell_offsets,
                        const int32_t* d_cell_index, int n_cells, int whole, int max_cell, int N, int G,
                        double* d_robust_disp);
/* dsq_dev_robust_disp with the size of the design's smallest cell (min_cell; ignored when whole != 0): when every cell
 * has at least 129 samples and the largest at least 2048 - or there are no cells at all (designs with continuous
 * covariates: one trimmed variance over all N samples) - the trimmed sums recompute the normalised counts from the gene's
 * row instead of buffering a cell in LDS (csrc/dsq_stats.h, robust_disp_gene_lean; 8 KB instead of 8 * next_pow2(cell)
 * bytes per wavefront).  Same results up to the rounding of a fused multiply-add. */
int dsq_dev_robust_disp2(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const int32_t* d_cell_offsets,
                         const int32_t* d_cell_index, int n_cells, int whole, int max_cell, int min_cell, int N, int G,
                         double* d_robust_disp);
/* dsq_dev_irls with (a) the optional cell path, (b) the per-sample half of the Cook's stage fused into its epilogue
 * (d_flags != NULL: d_robust_disp from dsq_dev_robust_disp in, d_cooks (nullable layer) and the four flag vectors of
 * dsq_dev_cooks out) and (c) the Wald statistics of dsq_dev_wald (h_ridge != NULL) computed while mu is in registers.
 * d_mu / d_hat may then be NULL: the N x G layers are not written unless asked for (dsq_dev_irls_layers rebuilds
 * them from beta on demand). */
int dsq_dev_lfc_fit(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                    const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank, const double* d_disp,
                    double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter, double* d_beta,
                    double* d_mu, double* d_hat, uint8_t* d_converged, int32_t* d_iters, const dsq_cells* cells,
                    const double* d_robust_disp, const uint8_t* d_flags, double cutoff, double* d_cooks,
                    uint8_t* d_any_all, uint8_t* d_any_use, uint8_t* d_any_use_nr, uint8_t* d_few_above,
                    const double* h_ridge, const double* h_contrast, double lfc_null, int alt, double* d_pvals,
                    double* d_stats, double* d_se);
/* dsq_dev_lfc_fit with a mixed design (dsq_mix_create): the fit, its start values and its epilogue run k_irls_mix
 * (csrc/dsq_k_irls_mix.hip: samples sorted by design cell, X^T W X from per-cell + covariate sums); mix == NULL: as
 * dsq_dev_lfc_fit.  Same outputs; diverged genes take the same rescue pass.
 * cooks_ld != 0 (>= the design's slots, dsq_mix_info; needs mix): d_cooks is written in SLOT order with that row pitch -
 * coalesced stores instead of a scatter over the row (which cost 5 x the layer's bytes in HBM writes); its readers are
 * dsq_dev_replace_outliers2 and dsq_mix_slots.  cooks_ld == 0: sample order, pitch ldn, as dsq_dev_lfc_fit. */
int dsq_dev_lfc_fit2(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                     const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank, const double* d_disp,
                     double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter, double* d_beta,
                     double* d_mu, double* d_hat, uint8_t* d_converged, int32_t* d_iters, const dsq_cells* cells,
                     const double* d_robust_disp, const uint8_t* d_flags, double cutoff, double* d_cooks,
                     uint8_t* d_any_all, uint8_t* d_any_use, uint8_t* d_any_use_nr, uint8_t* d_few_above,
                     const double* h_ridge, const double* h_contrast, double lfc_null, int alt, double* d_pvals,
                     double* d_stats, double* d_se, const dsq_mix* mix, int cooks_ld);
/* dsq_dev_replace_outliers for a Cook's layer in slot order (cooks_ld, mix as given to dsq_dev_lfc_fit2); cooks_ld == 0:
 * as dsq_dev_replace_outliers. */
int dsq_dev_replace_outliers2(dsq_ctx* ctx, const int32_t* d_y, const double* d_cooks, int ldn, const double* d_sf,
                              const uint8_t* d_flags, const int32_t* d_gene_idx, int n_sel, int N, double cutoff,
                              int32_t* d_y_out, uint8_t* d_all_zero, int cooks_ld, const dsq_mix* mix);
int dsq_dev_irls_layers(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt, int ldx,
                        int N, int G, int P, const double* d_disp, const double* d_beta, double min_mu, double* d_mu,
                        double* d_hat);
/* Side stream: work enqueued between dsq_side_begin and dsq_side_end runs on a second HIP stream that starts after
 * everything enqueued so far and overlaps what the main stream does next; dsq_side_wait makes the main stream wait
 * for it.  (The pipeline puts dsq_dev_robust_disp underneath the latency-bound trend / prior kernels.) */
int dsq_side_begin(dsq_ctx* ctx);
int dsq_side_end(dsq_ctx* ctx);
int dsq_side_wait(dsq_ctx* ctx);
/* Error paths of the caller: back to the main stream from whatever state, then both streams synchronised (buffers the
 * side stream was writing may be recycled afterwards).  No-op on a context that never forked. */
int dsq_side_abort(dsq_ctx* ctx);
/* d_mu / d_hat may be null.  d_iters may be null. */
int dsq_dev_irls(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                 const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank,
                 const double* d_disp, double min_mu, double beta_tol, double min_beta,
                 double max_beta, int maxiter, double* d_beta, double* d_mu, double* d_hat,
                 uint8_t* d_converged, int32_t* d_iters);
/* Cook's distances (dds.py:986-1040) with the robust trimmed dispersion
 * (utils.py:914-960) + the per-gene ingredients of _replace_outliers / cooks_outlier
 * (dds.py:1325-1326, 1083-1101).  d_cell_offsets/index: samples grouped by design cell
 * (cells with >= 3 replicates); whole != 0: no such cell.  d_flags[N]: bit0 cell >= 3,
 * bit1 cell >= min_replicates.  d_cooks may be null. */
int dsq_dev_cooks(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_mu,
                  const double* d_hat, const int32_t* d_cell_offsets, const int32_t* d_cell_index,
                  int n_cells, int whole, int max_cell, const uint8_t* d_flags, int N, int G, int P,
                  double cutoff, double* d_cooks, double* d_robust_disp, uint8_t* d_any_all,
                  uint8_t* d_any_use, uint8_t* d_any_use_nr, uint8_t* d_few_above);
/* Outlier replacement (dds.py:1329-1358) for the genes listed in d_gene_idx[n_sel]:
 * writes new int32 gene-major rows d_y_out[n_sel][ldn] where samples with
 * cooks > cutoff in replaceable cells get int(trimmed_mean(normed, 0.2) * sf). */
int dsq_dev_replace_outliers(dsq_ctx* ctx, const int32_t* d_y, const double* d_cooks, int ldn,
                             const double* d_sf, const uint8_t* d_flags, const int32_t* d_gene_idx,
                             int n_sel, int N, double cutoff, int32_t* d_y_out,
                             uint8_t* d_all_zero);
/* d_mu may be null (then mu = sf * exp(X beta) is recomputed, ds.py:320-324) */
int dsq_dev_wald(dsq_ctx* ctx, const double* d_mu, int ldn, const double* d_sf, const double* d_Xt,
                 int ldx, int N, int G, int P, const double* d_disp, const double* d_beta,
                 const double* h_ridge, const double* h_contrast, double lfc_null, int alt,
                 double* d_pvals, double* d_stats, double* d_se);
/* row gathers for the refit sub-problem: dst[k][:] = src[idx[k]][:] */
/* O(G) glue that keeps the dispersion vectors device-resident between the stages:
 * fitted trend a0 + a1/normed_mean (dds.py:826-833; a1 == 0: mean trend), final dispersions with the
 * dispersion-outlier rule (dds.py:912-935; inputs are the UNclipped fits), and the write-back of the
 * refitted genes (dds.py:1410-1458). */
/* dsq_dev_trend_fit + the fitted values (dds.py:826-833) + dsq_dev_prior_mad with one host synchronisation: the
 * coefficients stay on the device between the three kernels.  *h_ok = 0: the trend fit did not converge
 * (dds.py:811-823); the caller then takes the mean trend through dsq_dev_trend_eval / dsq_dev_prior_mad. */
int dsq_dev_trend_prior(dsq_ctx* ctx, const double* d_disp, const double* d_means, int n, double min_disp,
                        double max_disp, uint8_t* d_keep, double* d_fitted, double* d_work, double* h_coeffs2,
                        int* h_ok, int* h_n_outer, double* h_squared_logres);
/* An LFC fit in two launches (dds.py:937-984 is per gene: a gene's fit needs its own final dispersion only).  The genes
 * whose MAP fit finished - and converged - in the dispersion stage's full-size launch are fitted from a stream of their
 * own while the main stream still runs that stage's latency-bound tail (the continuation of the parked fits, the
 * grid-search pass); the rest follow in a second launch that joins the first.  The pipeline's sequence, from inside the
 * hook of dsq_set_alpha_hook (before the stage: MAP flags and late flags filled with 0xFF, dsq_alpha_set_late_flags,
 * dsq_lfc_prepare):
 *   dsq_lfc_fork_begin; dsq_dev_select_dispersions_part(mode 1); dsq_lfc_set_part(part, 1, 1); dsq_dev_lfc_fit2;
 *   dsq_lfc_fork_end;
 * and, after the dispersion stage has returned:
 *   dsq_dev_select_dispersions_part(mode 0); dsq_lfc_set_part(part, 0, 2); dsq_dev_lfc_fit2 (the same arguments).
 * Every gene is fitted exactly once: the results are those of the single launch.  dsq_lfc_takes_parts: does the kernel
 * family of this design honour the part vector (the run-time-P LDS kernels do not)?  dsq_side_abort also ends a fork whose
 * second launch will not come. */
int dsq_lfc_fork_begin(dsq_ctx* ctx);
int dsq_lfc_fork_end(dsq_ctx* ctx);
int dsq_lfc_set_part(dsq_ctx* ctx, const uint8_t* d_part, int want, int phase);
int dsq_lfc_takes_parts(int N, int P, const dsq_cells* cells, const dsq_mix* mix, int full_rank);
/* dsq_dev_select_dispersions for one part of the genes: mode 1 - the genes g < ready_limit with d_map_converged[g] == 1
 * (d_part[g] = 1 for them, 0 for the others, which are left alone); mode 0 - the genes with d_part[g] == 0, whose late
 * flags (d_conv_late, may be NULL) move into d_map_converged.  dsq_alpha_set_late_flags (one-shot, before the dispersion
 * fit): the launches of that fit enqueued after its hook's point write their convergence flags to d_conv_late instead
 * of d_converged - both filled with 0xFF by the caller -, so that d_converged holds, from the end of the full-size launch
 * on, exactly the flags of the genes that launch finished: what mode 1 reads cannot depend on how far the tail has come. */
int dsq_dev_select_dispersions_part(dsq_ctx* ctx, double* d_genewise_raw, double* d_map_raw, const double* d_fitted, int n,
                                    double min_disp, double max_disp, double squared_logres, double* d_disp,
                                    uint8_t* d_outlier, uint8_t* d_map_converged, const uint8_t* d_conv_late, uint8_t* d_part,
                                    int mode, int ready_limit);
int dsq_alpha_set_late_flags(dsq_ctx* ctx, uint8_t* d_conv_late);
/* The small operations of the first launch (logs of the size factors, ridge and contrast of the Wald test, zeroed
 * counters), enqueued on the main stream ahead of the dispersion stage instead of beside its tail. */
int dsq_lfc_prepare(dsq_ctx* ctx, const double* d_sf, int N, const double* h_ridge, const double* h_contrast, int P);
int dsq_dev_trend_eval(dsq_ctx* ctx, const double* d_normed_means, int n, double a0, double a1,
                       double* d_fitted);
/* dispersion-outlier rule and final dispersions (dds.py:909-935); the genewise and MAP dispersions are clipped to
 * [min_disp, max_disp] IN PLACE, as the reference stores them (dds.py:792-794, 905-907). */
int dsq_dev_select_dispersions(dsq_ctx* ctx, double* d_genewise_raw, double* d_map_raw,
                               const double* d_fitted, int n, double min_disp, double max_disp,
                               double squared_logres, double* d_disp, uint8_t* d_outlier);
int dsq_dev_scatter_rows_f64(dsq_ctx* ctx, const double* d_src, const int32_t* d_idx, int n_idx, int width,
                             double* d_dst);
/* "iterative" size factors (dds.py:1460-1548).  The Powell search over the log size factors
 * is host code (scipy, as in the reference); the device supplies its objective: per-gene NLL of the counts
 * under mu_hat * scale_n (dsq_dev_nll_scaled; the alpha-only part once per outer iteration,
 * dsq_dev_nll_const), and the reference's start values for the dispersion fit, i.e. the method-of-moments
 * estimate on the RAW counts (d_ones = normalising size factors of 1) with mean(1/size factor) of the
 * current size factors (dsq_dev_mom_raw). */
int dsq_dev_mom_raw(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_ones, const double* d_sf,
                    const double* d_Xt, const double* d_pinvXt, int ldx, int N, int G, int P, double min_disp,
                    double max_disp, double* d_normed_mean, double* d_mom);
int dsq_dev_nll_const(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, const double* d_disp, double* d_cst);
int dsq_dev_nll_scaled(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, int N, int G,
                       const double* d_disp, const double* d_scale, const double* d_cst, double* d_nll);
/* "poscounts" size factors (dds.py:655-680): log geometric means over the positive counts
 * (zeros contribute 0 to the mean over all samples) and the usable-gene mask (finite, > 0); feed both to
 * dsq_dev_size_factors (which leaves a sample's zero counts out of its median) and divide the result by its
 * geometric mean.  d_gene_mask of dsq_dev_size_factors also serves `control_genes`. */
int dsq_dev_logmeans_poscounts(dsq_ctx* ctx, const int32_t* d_y, int ldn, int N, int G, double* d_logmeans,
                               uint8_t* d_usable);
/* Variance stabilising transformation of the normalised counts (dds.py:486-514) on the
 * sample-major matrix as uploaded: mode 0 parametric trend (a0, a1), mode 1 mean dispersion (a0). */
int dsq_dev_vst(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G, const double* d_sf, int mode,
                double a0, double a1, double* d_out);
/* apeGLM MAP log-fold changes: Inference.lfc_shrink_nbinom_glm (inference.py:306-362,
 * default_inference.py:232-264 -> utils.nbinomGLM, utils.py:990-1142).  size = 1/dispersion [G],
 * offset = log(size factors) [N]; outputs beta [G][P], inv_hessian [G][P][P] (the reference's matrix,
 * incl. its broadcast prior curvature), converged [G] (L-BFGS-B success; for P == 2 a failed gene has
 * already been re-fitted by the reference's grid search).
 * optimizer (utils.py:990-1121 hands the name to scipy.optimize.minimize): 0 "L-BFGS-B" (what ds.py:407 passes; every
 * design width up to 32), 1 "BFGS" (gtol = 1e-8), 2 "Newton-CG" (with the reference's Hessian; scipy's default xtol - the
 * ftol / gtol options are unknown to that method) - the latter two for designs of at most DSQ_BFGS_MAX_P = 12 columns;
 * `converged` is the chosen optimiser's res.success. */
int dsq_inf_lfc_shrink_nbinom_glm2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                                   const double* design, const double* size, const double* offset, int N, int G,
                                   int P, double prior_no_shrink_scale, double prior_scale, int shrink_index,
                                   double* beta_out, double* inv_hessian_out, uint8_t* converged, int optimizer);
/* (the signature of ABI versions < 5, without the trailing optimizer: L-BFGS-B) */
int dsq_inf_lfc_shrink_nbinom_glm(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                                  const double* design, const double* size, const double* offset, int N, int G,
                                  int P, double prior_no_shrink_scale, double prior_scale, int shrink_index,
                                  double* beta_out, double* inv_hessian_out, uint8_t* converged);
int dsq_dev_lfc_shrink(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_offset, const double* d_Xt,
                       int ldx, int N, int G, int P, const double* d_size, double prior_no_shrink_scale,
                       double prior_scale, int shrink_index, double* d_beta, double* d_inv_hessian,
                       uint8_t* d_converged);
/* dsq_dev_lfc_shrink that may also (or only) return d_ih_entry[G] = inv_hessian[g][shrink_index][shrink_index] - all that
 * DeseqStats.lfc_shrink derives the shrunken lfcSE from (ds.py:424-433): 8 bytes per gene back instead of 8 p^2.
 * d_inv_hessian and d_ih_entry may each be NULL; d_ih_entry needs P <= 12. */
int dsq_dev_lfc_shrink2(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_offset, const double* d_Xt,
                        int ldx, int N, int G, int P, const double* d_size, double prior_no_shrink_scale,
                        double prior_scale, int shrink_index, double* d_beta, double* d_inv_hessian,
                        uint8_t* d_converged, double* d_ih_entry);
/* ... with the optimiser as dsq_inf_lfc_shrink_nbinom_glm takes it (0 "L-BFGS-B", 1 "BFGS", 2 "Newton-CG": utils.py:1112-1121) */
int dsq_dev_lfc_shrink3(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_offset, const double* d_Xt,
                        int ldx, int N, int G, int P, const double* d_size, double prior_no_shrink_scale,
                        double prior_scale, int shrink_index, double* d_beta, double* d_inv_hessian,
                        uint8_t* d_converged, double* d_ih_entry, int optimizer);
/* Adjusted p-values of DeseqStats.summary() (ds.py:486-542).
 * prepare: sorts the p-values once, derives the 50 baseMean cut-offs (np.quantile of base_mean at
 *   theta = linspace(mean(base_mean == 0), 0.95 or 1, 50)), assigns every gene the number of cut-offs it
 *   passes (d_bins) and counts the BH rejections (adjusted p < alpha) of each of the 50 passes.
 *   h_out200 = theta[50], cutoffs[50], num_rej[50], m[50]; *h_n_valid = genes with a p-value.
 *   d_pvalue already carries the Cook's filter (NaN for outlier genes, ds.py:543-549).
 * finish: BH-adjusted p-values of pass j scattered to gene order (NaN for genes outside the pass);
 *   j = -1 is plain BH over all genes with a p-value (ds.py:529-542).
 * The lowess choice of j (ds.py:512-525, 50 points) is host code (pydeseq2_amd/summary.py). */
int dsq_dev_padj_prepare(dsq_ctx* ctx, const double* d_base_mean, const double* d_pvalue, int n, double alpha,
                         unsigned long long* d_sorted_p, int32_t* d_sorted_idx, uint8_t* d_bins,
                         double* h_out200, int* h_n_valid);
int dsq_dev_padj_finish(dsq_ctx* ctx, const unsigned long long* d_sorted_p, const int32_t* d_sorted_idx,
                        const uint8_t* d_bins, int n, int n_valid, int j, double* d_padj);
/* Host count matrix (int32 or int64, any layout: n_elems consecutive elements) -> int32 device buffer with the
 * same element order.  int64 is narrowed on the host by a few threads into two page-locked staging chunks
 * whose DMA overlaps the narrowing of the next chunk (half the PCIe bytes of the int64 matrix, no pageable
 * staging).  *h_bad = 1 if a value is negative or >= 2^31.  Synchronous on return. */
int dsq_upload_counts_i32(dsq_ctx* ctx, const void* counts, int count_type, size_t n_elems, int32_t* d_dst,
                          int* h_bad);
/* device-to-device copy on the context's stream */
int dsq_d2d(dsq_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* page-locked host memory + asynchronous copies on the context's stream (complete at dsq_sync) */
int dsq_host_alloc(dsq_ctx* ctx, size_t bytes, void** out);
int dsq_host_free(dsq_ctx* ctx, void* p);
int dsq_d2h_async(dsq_ctx* ctx, void* pinned_dst, const void* d_src, size_t bytes);
int dsq_h2d_async(dsq_ctx* ctx, void* d_dst, const void* pinned_src, size_t bytes);
int dsq_dev_gather_rows_f64(dsq_ctx* ctx, const double* d_src, int ld, const int32_t* d_idx,
                            int n_idx, int ncols, double* d_dst);
int dsq_dev_gather_rows_i32(dsq_ctx* ctx, const int32_t* d_src, int ld, const int32_t* d_idx,
                            int n_idx, int ncols, int32_t* d_dst);

/* ================================================================== multi-GPU exchanges (RCCL over xGMI)
 * One process per GPU; genes shard across ranks.  Only two steps of the path need other ranks'
 * genes: the per-sample median of log-ratios (size factors) and the dispersion
 * trend / prior.  librccl is dlopen()ed on first use. */
int dsq_comm_unique_id(dsq_ctx* ctx, char* out128, int len);           /* rank 0, then broadcast */
int dsq_comm_init(dsq_ctx* ctx, const char* uid128, int rank, int world);
int dsq_comm_destroy(dsq_ctx* ctx);
int dsq_comm_allreduce_sum(dsq_ctx* ctx, void* d_buf, size_t count, int dtype /*0 u32, 1 f64*/);
int dsq_comm_allgather(dsq_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* nranks / rank as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank) */
int dsq_comm_info(dsq_ctx* ctx, int* nranks, int* rank);
/* number of host-side waits (hipStreamSynchronize / hipEventSynchronize) this library has made in this process */
unsigned long long dsq_host_sync_count(void);
/* Gene-sharded trend exchange: d_send[0..len) = d_a[0..n) then NaN, d_send[len..2 len) = d_b[0..n) then NaN (one
 * all-gather carries both per-gene vectors); d_recv [world][2][len] -> d_a_all, d_b_all [world * len]. */
int dsq_dev_pack2(dsq_ctx* ctx, const double* d_a, const double* d_b, int n, int len, double* d_send);
int dsq_dev_unzip2(dsq_ctx* ctx, const double* d_recv, int world, int len, double* d_a_all, double* d_b_all);

/* size factors pass by pass: keys -> per-sample counts -> (all-reduce) -> init -> 8 x
 * [hist -> (all-reduce) -> pick] -> finish.  d_keys: N*G u64, d_prefix: 2*N u64, d_rank: 2*N u32,
 * d_hist: 2*N*256 u32. */
int dsq_dev_sf_keys(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                    const double* d_logmeans, const uint8_t* d_gene_mask, void* d_keys);
/* same, but only for the usable genes (finite logmean, mask): keys [N][*h_n_usable]; the following passes
 * then take G = *h_n_usable.  d_idx_work: G + 2 ints. */
int dsq_dev_sf_keys_compact(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                            const double* d_logmeans, const uint8_t* d_gene_mask, int32_t* d_idx_work, void* d_keys,
                            int* h_n_usable);
int dsq_dev_sf_count(dsq_ctx* ctx, const void* d_keys, int N, int G, uint32_t* d_counts);
int dsq_dev_sf_init(dsq_ctx* ctx, const uint32_t* d_total, int N, void* d_prefix, uint32_t* d_rank);
int dsq_dev_sf_hist(dsq_ctx* ctx, const void* d_keys, int N, int G, const void* d_prefix, int shift,
                    uint32_t* d_hist);
int dsq_dev_sf_pick(dsq_ctx* ctx, const uint32_t* d_hist, int N, int shift, void* d_prefix,
                    uint32_t* d_rank);
int dsq_dev_sf_finish(dsq_ctx* ctx, const void* d_prefix, const uint32_t* d_total, int N, double* d_sf);

#ifdef __cplusplus
}
#endif
#endif /* DESEQ_HIP_H */
