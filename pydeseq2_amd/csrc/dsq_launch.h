// dsq_launch.h — host-side launch entry points implemented in the dsq_k_*.hip units.
// One gene per 64-lane wavefront, 4 genes per 256-thread workgroup (WPB waves / block).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/deseq_hip.h"
#include "dsq_linalg.h"
#include "dsq_mix.h"

namespace dsq {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

inline int genes_to_blocks(int G) { return (G + kWavesPerBlock - 1) / kWavesPerBlock; }

// compute units of the CURRENT device (the persistent kernels size their grids by it; a process may drive several
// different GPUs, one context each: cached per device ordinal, not per process); 0 on error
inline int current_device_cus() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) cache[dev] = prop.multiProcessorCount;
    return prop.multiProcessorCount;
}

// waves per SIMD requested for the cell-path kernels: their sample loops need few registers at any P, but the
// p x p algebra between the loops (Cholesky, inverse, Wald) holds 2-3 packed matrices
#ifndef DSQ_CELL_WAVES_WIDE
#define DSQ_CELL_WAVES_WIDE 2
#endif
constexpr int cell_min_waves(int p) { return p <= 6 ? 3 : DSQ_CELL_WAVES_WIDE; }

// ---- dsq_k_wide.hip: run-time-P kernels (LDS matrices, matrix-core Gram accumulation), P <= 32
int wide_min_p();
bool wide_with_cells();  // DSQ_WIDE_CELLS=1 (measurements): designs with cells may take the LDS path as well
  // designs at least this wide take them (default: DSQ_REG_MAX_P + 1)
struct IrlsExtras;
hipError_t launch_wide_mom(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                           const double* pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                           double min_mu, double* normed_mean, double* rough, double* moments, double* mom,
                           double* mu, double* coef, const double* d_s_mean_inv);
hipError_t launch_wide_rough_normed(hipStream_t st, const double* normed, int ldn, const double* Xt,
                                    const double* pinvXt, int ldx, int N, int G, int P, double* out);
hipError_t launch_wide_alpha(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx,
                             int N, int G, int P, const double* alpha_hat, double min_disp, double max_disp,
                             double prior_var, int cr_reg, int prior_reg, double* alpha, uint8_t* conv,
                             int32_t* nfev, double* nll_const, int const_mode, const CellDesign* cells);
hipError_t launch_wide_alpha_grid(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                                  int ldx, int N, int P, double min_disp, double max_disp, double* alpha,
                                  const int32_t* list, int n_list);
hipError_t launch_wide_irls(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* lsf,
                            const double* Xt, const double* pinvXt, int ldx, int N, int G, int P, int full_rank,
                            const double* disp, double min_mu, double beta_tol, double min_beta, double max_beta,
                            int maxiter, double* beta, double* mu, double* hat, uint8_t* conv, int32_t* iters,
                            int32_t* fb_count, int32_t* fb_list, const IrlsExtras* extras);
hipError_t launch_wide_irls_rescue(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* lsf,
                                   const double* Xt, const double* pinvXt, int ldx, int N, int P, int full_rank,
                                   const double* disp, double min_mu, double beta_tol, double min_beta,
                                   double max_beta, int maxiter, double* beta, double* mu, double* hat, uint8_t* conv,
                                   int32_t* iters, const int32_t* fb_list, int n_fb, const IrlsExtras* extras);
hipError_t launch_wide_irls_layers(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                                   int ldx, int N, int G, int P, const double* disp, const double* beta, double min_mu,
                                   double* mu, double* hat);
hipError_t launch_wide_wald(hipStream_t st, const double* mu, int ldn, const double* sf, const double* Xt, int ldx,
                            int N, int G, int P, const double* disp, const double* beta, const double* d_ridge,
                            const double* d_contrast, double lfc_null, int alt, double* pvals, double* stats,
                            double* se);

// ---- dsq_k_alpha.hip
// optional inputs of the dispersion kernel (zero-initialised = none)
struct AlphaExtras {
    CellDesign cells;    // cells.C > 0: the design's distinct rows (<= 64) -> per-cell weight sums (P >= 3)
    const double* coef;  // [G][P] OLS coefficients of the normalised counts (k_mom_lin_mu): mu_hat is computed
    const double* sf;    // while staging instead of being read from `mu` (which may then be null)
    double min_mu;
    // gene lists (dsq_k_alpha_rows.hip): the genes of `rows` [n_rows] run four to a wavefront, those of `waves`
    // [n_waves] one to a wavefront (k_alpha); together they are the G genes of the call.  nullptr: all on k_alpha.
    const int32_t* rows;
    const int32_t* waves;
    int n_rows, n_waves;
    const double* cell_mu;  // [G][cells.C] per-cell mu_hat / size factor of the IRLS mu_hat route (k_cell_mu): mu_hat_n =
                            // sf_n * cell_mu[cell_of[n]] UNclamped, instead of `mu` or `coef`
    const int32_t* cell_mu_of;  // (internal) cell of every sample / number of cells for cell_mu
    int cell_mu_C;
    const int32_t* list;  // (internal) k_alpha takes gene list[k] instead of gene k
    // Parking (row kernel, dsq_k_alpha_rows.hip).  A fit whose line search ends in rounding noise takes 20-34 evaluations
    // (0.1-0.3 % of the genes; the median is 5).  The row kernel stops a gene after eval_cap evaluations and parks it (the
    // optimiser's state, 296 bytes) in resume_state / resume_list; k_alpha then continues the parked genes (resume != 0,
    // genes = the parked list, count on the device), all at once, each on a wavefront of its own.  Same iterates.
    // (For k_alpha itself the same two-phase scheme was measured and brings nothing: c3 8.31 -> 8.52 ms per step.)
    int eval_cap;
    int resume;
    void* resume_state;       // [G] Lbfgsb1d records
    int32_t* resume_count;
    int32_t* resume_list;     // [G]
    const int32_t* n_dev;     // (internal) phase B: number of entries of `list` on the device
    // called once, on the launching thread, when the full-size kernel(s) of the fit are enqueued and the latency-bound
    // tail (continuation of the parked fits, second passes) is about to be (dsq_set_alpha_hook); may be null
    void (*mid_hook)(void*);
    void* mid_arg;
    // non-null: the launches enqueued AFTER the hook's point (continuation of the parked fits, the genes of `waves`) write
    // their convergence flags here instead of `conv` - which then holds, from the end of the full-size launch on, exactly
    // the flags of the genes that launch finished (dsq_alpha_set_late_flags: what the forked LFC launch selects its genes by)
    uint8_t* conv_late;
    // Mixed designs (dsq_mix.h, dsq_k_alpha_mix.hip): the genes of `rows` run k_alpha_mix.
    // mix_ys [G][Ns] uint16: the counts in slot order (launch_mix_counts_to_slots); mix_mu [G][Ns]: mu_hat in slot order
    // (launch_mix_mu_slots from the IRLS coefficients, or launch_mix_f64_to_slots from a caller's matrix) - the kernel
    // streams both rows; mix_beta stays for the rows of the grid-search genes (launch_mu_from_beta).
    const MixDesign* mix;
    const double* mix_beta;
    const uint16_t* mix_ys;
    const double* mix_mu;
};
constexpr int kAlphaEvalCap = 8;
size_t alpha_resume_bytes(int G);
// ---- dsq_k_alpha_rows.hip: four genes per wavefront for linear-model mu_hat designs with <= 4 cells
constexpr int kRowTail = 512;    // counts below this enter a gene's tail-count table
bool alpha_rows_eligible(int N, int P, int n_cells, bool has_coef, int cr_reg);
hipError_t launch_alpha_rows(hipStream_t st, const int32_t* y, int ldn, int N, const int32_t* list, int n_list,
                             int32_t* queue, const double* coef, const double* sf, const CellDesign& cells, int P,
                             double min_mu, const double* alpha_hat, double min_disp, double max_disp, double prior_var,
                             int cr_reg, int prior_reg, double* alpha, uint8_t* conv, int32_t* nfev, int32_t* grid_count,
                             int32_t* grid_list, double* nll_const, int const_mode, int eval_cap, void* park_state,
                             int32_t* park_count, int32_t* park_list);
hipError_t launch_count_big(hipStream_t st, const int32_t* y, int ldn, int N, int G, int32_t* out);
// ---- dsq_k_alpha_rowsc.hip: four genes per wavefront for designs with up to 32 cells (per-cell tables in LDS)
int alpha_rowsc_tail(int N, int P, int n_cells);  // tail-count table size (0: not eligible)
hipError_t launch_alpha_rows_c(hipStream_t st, const int32_t* y, int ldn, int N, const int32_t* list, int n_list,
                               int32_t* queue, const double* coef, const double* cell_mu, const double* sf,
                               const CellDesign& cells, int P, double min_mu, const double* alpha_hat, double min_disp,
                               double max_disp, double prior_var, int prior_reg, double* alpha, uint8_t* conv,
                               int32_t* nfev, int32_t* grid_count, int32_t* grid_list, double* nll_const,
                               int const_mode, int eval_cap, void* park_state, int32_t* park_count,
                               int32_t* park_list);
hipError_t launch_cell_mu(hipStream_t st, const double* beta, const double* Xc, int C, int G, int P, double* cell_mu);
hipError_t launch_mu_from_cells(hipStream_t st, const double* cell_mu, int C, const double* sf, const int32_t* cell_of,
                                int N, const int32_t* list, int n_list, double* dst, int ldn, int32_t* idx_out,
                                const int32_t* n_dev = nullptr);
// ---- dsq_k_alpha_mix.hip (one translation unit per number of continuous covariates): mixed designs, dsq_mix.h
bool alpha_mix_enabled();
int alpha_mix_launches();  // launches of k_alpha_mix by this process so far (tests: did the design take that route?)
bool alpha_mix_fits(const MixDesign& D);  // rows of D.Ns slots fit the kernel's LDS
// ys [G][Ns] uint16 / mu_s [G][Ns] fp64: the genes' counts and mu_hat in slot order (launch_mix_counts_to_slots, ...)
hipError_t launch_alpha_mix(hipStream_t st, const uint16_t* ys, const double* mu_s, const MixDesign& D, const int32_t* list,
                            int n_list, const int32_t* n_dev, int32_t* queue, const double* alpha_hat, double min_disp,
                            double max_disp, double prior_var, int prior_reg, double* alpha, uint8_t* conv, int32_t* nfev,
                            int32_t* grid_count, int32_t* grid_list, double* nll_const, int const_mode, int eval_cap,
                            int resume, void* park_state, int32_t* park_count, int32_t* park_list);
// rows of mu_hat = sf * exp(X beta) (unclamped) for a gene list (grid-search pass when no N x G mu_hat exists)
hipError_t launch_mu_from_beta(hipStream_t st, const double* beta, const double* sf, const double* Xt, int ldx, int N,
                               int P, const int32_t* list, int n_list, double* dst, int ldn, int32_t* idx_out,
                               const int32_t* n_dev = nullptr);
bool alpha_wg_eligible(int N);
hipError_t launch_alpha_wg(hipStream_t st, const int32_t* y, int ldn, int N, const int32_t* list, const int32_t* n_dev,
                           int n_cap, const double* coef, const double* sf, const CellDesign& cells, int P,
                           double min_mu, const double* alpha_hat, double prior_var, int prior_reg, double* alpha,
                           uint8_t* conv, int32_t* nfev, int32_t* grid_count, int32_t* grid_list,
                           const double* nll_const, const void* park_state);
// optimizer="BFGS" variant of the dispersion fit (P <= DSQ_REG_MAX_P, mu_hat as a matrix)
hipError_t launch_alpha_bfgs(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx,
                             int N, int G, int P, const double* alpha_hat, double min_disp, double max_disp,
                             double prior_var, int cr_reg, int prior_reg, double* alpha, uint8_t* conv,
                             int32_t* nfev, int32_t* grid_count, int32_t* grid_list);
hipError_t launch_alpha(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                        int ldx, int N, int G, int P, const double* alpha_hat, double min_disp,
                        double max_disp, double prior_var, int cr_reg, int prior_reg, double* alpha,
                        uint8_t* conv, int32_t* nfev, int32_t* grid_count, int32_t* grid_list,
                        double* nll_const, int const_mode, const AlphaExtras* extras = nullptr, int32_t* queue = nullptr);
hipError_t launch_alpha_grid(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                             int ldx, int N, int P, double min_disp, double max_disp, double* alpha,
                             const int32_t* grid_list, int n_grid, double* work, const int32_t* n_dev = nullptr,
                             bool mu_compact = false);
constexpr int kAlphaGridWorkDoubles = 103;  // per listed gene: interval, 100 grid values, the completion counter

// ---- dsq_k_irls.hip
// optional inputs / fused outputs of the IRLS kernel (zero-initialised = none)
struct IrlsExtras {
    CellDesign cells;            // cells.C > 0: per-cell path (dsq_irls.h, irls_sweep_cell)
    // fused Cook's bookkeeping (LfcEpilogue): on when flags != nullptr
    const double* robust_disp;   // [G] from launch_robust_disp
    const uint8_t* flags;        // [N]
    double cutoff;
    double* cooks;               // [G][ldn] or null
    int cooks_ld;                // mixed designs: != 0: the layer is written in SLOT order with this pitch (>= mix->Ns)
    double* cooks_tmp;           // rescue of diverged genes under cooks_ld != 0: [n_fb][ldn] scratch the general rescue kernels
                                 // write their Cook's rows to in SAMPLE order (row k = k-th gene of the fallback list);
                                 // launch_irls_rescue then scatters them into the slot-ordered layer
    uint8_t *any_all, *any_use, *any_use_nr, *few_above;  // [G]
    // fused Wald statistics: on when ridge != nullptr (device pointers)
    const double* ridge;         // [P*P]
    const double* contrast;      // [P]
    double lfc_null;
    int alt;
    double *pvals, *stats, *se;  // [G]
    int optimizer;               // rescue of diverged genes: 0 L-BFGS-B (bounded, default), 1 BFGS (utils.py:389-399)
    // k_irls_row: slot -> gene (launch_irls_order), so that the four genes of a wavefront need about the same
    // number of sweeps; null: identity.  Results do not depend on it.
    const int32_t* order;
    // mixed designs (dsq_mix.h, dsq_k_irls_mix.hip): mix != null routes the fit to k_irls_mix; mix_work: irls_mix_work_bytes
    // of device scratch (slot-ordered size factors, their logs, Cook's flags), mix_queue: a zeroed int32 gene counter
    const MixDesign* mix;
    void* mix_work;
    size_t mix_work_bytes;
    int32_t* mix_queue;
    const uint16_t* mix_ys;      // [G][Ns] counts in slot order (launch_mix_counts_to_slots) and, per gene, whether a count
    const uint8_t* mix_big;      // did not fit its 16 bits (such a gene gathers its counts from the int32 row)
    // a fit in two launches (dsq_lfc_set_part): part != null: only the genes with part[g] == part_want are fitted, the
    // others are left alone (no output of theirs is touched).  The register kernels and the mixed-design kernels take it
    // (irls_takes_parts); part_shared_ready: the launch's shared tables (mix_work) were written by the partner launch.
    const uint8_t* part;
    int part_want;
    int part_shared_ready;
};
// does launch_irls honour IrlsExtras::part for this design?  (the run-time-P LDS kernels do not)
bool irls_takes_parts(int N, int P, int n_cells, const MixDesign* mix, int full_rank);
bool irls_takes_mix(const MixDesign* mix, int full_rank);
// device scratch of a mixed-design fit of G genes that writes n_layers N x G layers (Cook's distances, mu, hat diagonal)
size_t irls_mix_work_bytes(const MixDesign& D, int G, int n_layers);
hipError_t launch_replace(hipStream_t st, const int32_t* y, const double* cooks, int ldn,
                          const double* sf, const uint8_t* flags, const int32_t* gene_idx, int n_sel,
                          int N, double cutoff, int32_t* y_out, uint8_t* all_zero, int cooks_ld,
                          const int32_t* slot_of);
// does launch_irls fit this design with sixteen lanes per gene (k_irls_row)?
bool irls_takes_rows(int N, int P, int n_cells);
// order[0..G) = genes by decreasing predicted number of IRLS sweeps: `hint_iters` (the iteration counts of an earlier
// fit of the same genes) when given, else the dispersion (noisier genes take more sweeps)
// work: irls_order_work_ints() int32 of device scratch
hipError_t launch_irls_order(hipStream_t st, const double* disp, const int32_t* hint_iters, int G, int32_t* order,
                             int32_t* work);
int irls_order_work_ints();
hipError_t launch_irls(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* lsf,
                       const double* Xt,
                       const double* pinvXt, int ldx, int N, int G, int P, int full_rank,
                       const double* disp, double min_mu, double beta_tol, double min_beta,
                       double max_beta, int maxiter, double* beta, double* mu, double* hat,
                       uint8_t* conv, int32_t* iters, int32_t* fb_count, int32_t* fb_list,
                       const IrlsExtras* extras = nullptr);
hipError_t launch_irls_rescue(hipStream_t st, const int32_t* y, int ldn, const double* sf,
                              const double* lsf, const double* Xt, const double* pinvXt, int ldx, int N, int P,
                              int full_rank, const double* disp, double min_mu, double beta_tol,
                              double min_beta, double max_beta, int maxiter, double* beta, double* mu,
                              double* hat, uint8_t* conv, int32_t* iters, const int32_t* fb_list,
                              int n_fb, const IrlsExtras* extras = nullptr, const int32_t* n_dev = nullptr);

bool irls_is_wide(int P, int n_cells);
bool alpha_is_wide(int P, int n_cells);
bool alpha_needs_mu(int N, int P, int n_cells);  // the design takes the run-time-P (LDS) kernels of dsq_k_wide.hip

// ---- dsq_k_stats.hip
hipError_t launch_widen_u16(hipStream_t st, const uint16_t* src, int32_t* dst, size_t n);
hipError_t launch_transpose_counts(hipStream_t st, const void* src, int count_type, int layout, int N,
                                   int G, int32_t* dst, int ldn, int* bad_flag);
hipError_t launch_transpose_f64(hipStream_t st, const double* src, int layout, int N, int G,
                                double* dst, int ldn);
hipError_t launch_logmeans(hipStream_t st, const int32_t* y, int ldn, int N, int G, double* logmeans,
                           uint8_t* nonzero);
hipError_t launch_size_factors(hipStream_t st, const void* counts_sm, int count_type, int N, int G,
                               const double* logmeans, const uint8_t* gene_mask, double* work,
                               double* sf, int zeros_low = 0);
hipError_t launch_mom(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                      const double* pinvXt, int ldx, int N, int G, int P, double min_disp,
                      double max_disp, double* normed_mean, double* rough, double* moments,
                      double* mom, double* d_scalar, const double* sf_moments = nullptr);
hipError_t launch_mom_lin_mu(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                             const double* pinvXt, int ldx, int N, int G, int P, double min_disp, double max_disp,
                             double min_mu, double* normed_mean, double* mom, double* mu, double* d_scalar,
                             double* coef = nullptr);
hipError_t launch_mu_from_coef(hipStream_t st, const double* coef, const double* sf, const double* Xt, int ldx, int N,
                               int P, double min_mu, const int32_t* list, int n_list, double* dst, int ldn,
                               int32_t* idx_out, const int32_t* n_dev = nullptr);
hipError_t launch_irls_layers(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt, int ldx,
                              int N, int G, int P, const double* disp, const double* beta, double min_mu, double* mu,
                              double* hat);
hipError_t launch_nll_const(hipStream_t st, const int32_t* y, int ldn, int N, int G, const double* disp, double* cst);
hipError_t launch_nll_scaled(hipStream_t st, const int32_t* y, const double* mu, int ldn, int N, int G,
                             const double* disp, const double* scale, const double* cst, double* nll);
hipError_t launch_lin_mu(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt,
                         const double* pinvXt, int ldx, int N, int G, int P, double min_mu,
                         double* mu);
hipError_t launch_wald(hipStream_t st, const double* mu, int ldn, const double* sf, const double* Xt,
                       int ldx, int N, int G, int P, const double* disp, const double* beta,
                       const double* d_ridge, const double* d_contrast, double lfc_null, int alt,
                       double* pvals, double* stats, double* se);
hipError_t launch_cooks(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* mu,
                        const double* hat, const int32_t* cell_offsets, const int32_t* cell_index,
                        int n_cells, int whole, int max_cell, const uint8_t* flags, int N, int G,
                        int P, double cutoff, double* cooks, double* robust_disp, uint8_t* any_all,
                        uint8_t* any_use, uint8_t* any_use_nr, uint8_t* few_above);
// robust dispersion of utils.robust_method_of_moments_disp (the design-only half of launch_cooks)
// min_cell: the smallest of the design's cells; redo: G + 1 int32 of scratch for the kernel without the LDS buffer
// (designs whose cells all have >= 129 samples), or null
hipError_t launch_robust_disp(hipStream_t st, const int32_t* y, int ldn, const double* sf,
                              const int32_t* cell_offsets, const int32_t* cell_index, int n_cells, int whole,
                              int max_cell, int N, int G, double* robust_disp, int min_cell = 0, int32_t* redo = nullptr);
hipError_t launch_gather_rows_f64(hipStream_t st, const double* src, int ld, const int32_t* idx,
                                  int n_idx, int ncols, double* dst);
// ---- dsq_k_shrink.hip (apeGLM MAP LFC)
hipError_t launch_shrink(hipStream_t st, const int32_t* y, int ldn, const double* offset, const double* Xt, int ldx,
                         int N, int G, int P, const double* size, double sigma0, double sigma, int shrink_index,
                         double* beta, double* invh, uint8_t* conv, double* ih_entry = nullptr, int optimizer = 0);
// ---- dsq_k_summary.hip (adjusted p-values of DeseqStats.summary())
size_t summary_sort_temp_bytes(int n);
hipError_t launch_padj_prepare(hipStream_t st, const double* base_mean, const double* pvalue, int n, double alpha,
                               void* sort_tmp, size_t sort_tmp_bytes, void* work, unsigned long long* sorted_p,
                               int* sorted_idx, unsigned char* bins, double* out200, int* counters);
hipError_t launch_padj_numrej(hipStream_t st, const unsigned long long* sorted_p, const int* sorted_idx,
                              const unsigned char* bins, int n_valid, double alpha, double* out200);
hipError_t launch_padj_finish(hipStream_t st, const unsigned long long* sorted_p, const int* sorted_idx,
                              const unsigned char* bins, int n, int n_valid, int j, int* rank_tmp, double* padj);
hipError_t launch_logmeans_pos(hipStream_t st, const int32_t* y, int ldn, int N, int G, double* logmeans,
                               uint8_t* usable);
size_t prior_mad_work_doubles(int n);
hipError_t launch_sf_compact(hipStream_t st, const double* logmeans, const uint8_t* gene_mask, int G, int* idx_work);
hipError_t launch_sf_keys_compact(hipStream_t st, const void* counts_sm, int count_type, int N, int G,
                                  const double* logmeans, const int* idx_work, unsigned long long* keys);
size_t size_factors_work_doubles(int N, int G);
hipError_t launch_vst(hipStream_t st, const void* counts_sm, int count_type, int N, int G, const double* sf, int mode,
                      double a0, double a1, double* out);
hipError_t launch_trend_eval(hipStream_t st, const double* nm, int n, double a0, double a1, double* fitted);
hipError_t launch_trend_eval_dev(hipStream_t st, const double* nm, int n, const double* coef, double* fitted);
hipError_t launch_select_disp(hipStream_t st, double* gw_raw, double* map_raw, const double* fitted,
                              int n, double min_disp, double max_disp, double two_sd, double* disp,
                              uint8_t* outlier);
hipError_t launch_select_disp_part(hipStream_t st, double* gw_raw, double* map_raw, const double* fitted, int n,
                                   double min_disp, double max_disp, double two_sd, double* disp, uint8_t* outlier,
                                   uint8_t* map_conv, const uint8_t* conv_late, uint8_t* part, int mode,
                                   int ready_limit);
// n_dev (here and below): the kernel is launched for n_* rows as a CAPACITY and reads the actual count from device
// memory - second passes can be enqueued without the host having seen how many genes need them
hipError_t launch_scatter_rows(hipStream_t st, const double* src, const int32_t* idx, int n_idx, int width,
                               double* dst, const int32_t* n_dev = nullptr);
hipError_t launch_gather_rows_i32(hipStream_t st, const int32_t* src, int ld, const int32_t* idx,
                                  int n_idx, int ncols, int32_t* dst, const int32_t* n_dev = nullptr);
constexpr int kTrendPartials = 256;  // rows of 4 doubles
hipError_t launch_trend_loss_grad(hipStream_t st, const double* cov, const double* targets,
                                  const uint8_t* keep, int n, double a0, double a1, double* partials);
size_t trend_grid_mem_bytes();
hipError_t launch_trend_fit(hipStream_t st, const double* disp, const double* means, int n, double min_disp,
                            double max_disp, uint8_t* keep, double* out5, void* grid_mem, int force_grid);
hipError_t launch_trend_glm(hipStream_t st, const double* targets, const double* cov, int n, uint8_t* keep,
                            double* out5);
// ---- grid searches as stand-alone entry points (grid_search.py:54-221)
hipError_t launch_grid_beta(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt, int ldx,
                            int N, int G, const double* disp, double min_mu, double min_beta, double max_beta,
                            int grid_length, double* beta);
// distributed size factors (per-pass radix select, histograms all-reduced between passes)
hipError_t launch_sf_keys(hipStream_t st, const void* counts_sm, int count_type, int N, int G,
                          const double* logmeans, const uint8_t* gene_mask, unsigned long long* keys);
hipError_t launch_sf_count(hipStream_t st, const unsigned long long* keys, int N, int G, unsigned int* counts);
hipError_t launch_sf_init(hipStream_t st, const unsigned int* total, int N, unsigned long long* prefix,
                          unsigned int* rank);
hipError_t launch_sf_hist(hipStream_t st, const unsigned long long* keys, int N, int G,
                          const unsigned long long* prefix, int shift, unsigned int* hist);
hipError_t launch_sf_pick(hipStream_t st, const unsigned int* hist, int N, int shift, unsigned long long* prefix,
                          unsigned int* rank);
hipError_t launch_sf_finish(hipStream_t st, const unsigned long long* prefix, const unsigned int* total, int N,
                            double* sf);
hipError_t launch_prior_mad(hipStream_t st, const double* gw_raw, const double* fitted, int n, double min_disp,
                            double max_disp, double* res_scratch, double* out2);
hipError_t launch_log_vec(hipStream_t st, const double* in, int n, double* out);
// mixed designs: slot-ordered copies (dsq_mix.h): counts as uint16 [G][Ns] (0xFFFF padding, 0xFFFE saturated + big[g]),
// fp64 rows [G][Ns] (0 padding), the IRLS route's mu_hat from its coefficients
hipError_t launch_mix_counts_to_slots(hipStream_t st, const int32_t* y, int ldn, const MixDesign& D, int G, uint16_t* ys,
                                      uint8_t* big);
hipError_t launch_mix_f64_to_slots(hipStream_t st, const double* m, int ldn, const MixDesign& D, int G, double* ms);
hipError_t launch_mix_mu_slots(hipStream_t st, const double* beta, const double* sf, const MixDesign& D, int G, double* mu);
hipError_t launch_pack2(hipStream_t st, const double* a, const double* b, int n, int len, double* send);
hipError_t launch_unzip2(hipStream_t st, const double* recv, int world, int len, double* a_all, double* b_all);
// normed counts (double, gene-major) based rough / moments for the Inference-level API
hipError_t launch_rough_from_normed(hipStream_t st, const double* normed, int ldn, const double* Xt,
                                    const double* pinvXt, int ldx, int N, int G, int P, double* out);
hipError_t launch_moments_from_normed(hipStream_t st, const double* normed, int ldn, int N, int G,
                                      double s_mean_inv, double* out);

}  // namespace dsq
