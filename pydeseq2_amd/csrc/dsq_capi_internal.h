// dsq_capi_internal.h — what the translation units of the C ABI share (round 6: dsq_capi.hip was one 2400-line unit):
//   dsq_capi_ctx.hip   context, device / host memory, streams, the count upload
//   dsq_capi_dev.hip   the stages on device-resident buffers (dsq_dev_*, dsq_mix_*)
//   dsq_capi_inf.hip   the Inference-level entry points (dsq_inf_*) and their device cache (dsq_plugin_cache.h)
//   dsq_capi_comm.hip  RCCL exchanges and the sample-sharded size-factor kernels (dsq_comm_*, dsq_dev_sf_*)
// Host-side glue only: context, argument checks, grow-only workspaces, kernel launches.  No math lives here.
// The helpers below sit in unnamed namespaces: every unit gets its own copy (they hold no state of their own - all
// state is in dsq_ctx).
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <atomic>
#include <mutex>

// every host-side wait of this library is counted (dsq_host_sync_count: bench.py reports host synchronisations per step)
extern std::atomic<unsigned long long> g_dsq_host_syncs;  // (defined in dsq_capi_ctx.hip)
#define hipStreamSynchronize(s) (++g_dsq_host_syncs, (hipStreamSynchronize)(s))
#define hipEventSynchronize(e) (++g_dsq_host_syncs, (hipEventSynchronize)(e))

#include "../../include/deseq_hip.h"
#include "dsq_launch.h"
namespace dsq_pc {
struct Cache;  // dsq_plugin_cache.h (dsq_capi_inf.hip only)
}

// a mixed design on the device (dsq_mix_create): csrc/dsq_mix.h
struct dsq_mix {
    dsq::MixDesign d{};
    void* d_block = nullptr;  // one allocation behind all of d's pointers
    int device = 0;
};

struct dsq_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t main_stream = nullptr, side_stream = nullptr;  // `stream` is whichever of the two is current
    hipStream_t small_stream = nullptr;  // CU-masked stream of the latency-bound cross-gene kernels (dsq_side_begin)
    hipEvent_t ev_small0 = nullptr, ev_small1 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
    float last_kernel_ms = 0.0f;  // k_alpha launch of the last dsq_*_alpha_mle call (HIP events)
    int last_n_grid = 0;          // genes that took the grid-search fallback in that call
    double* d_scratch = nullptr;  // 32 KiB of device scratch (kScratchBytes: scalars, trend partials, ridge / contrast)
    int32_t* d_counter = nullptr; // IRLS fallback / dispersion grid-search counters
    int32_t* d_list = nullptr;    // gene index list of the rare second-pass kernels (grown on demand)
    size_t list_cap = 0;
    void* d_trend_grid = nullptr; // global-memory mailbox of the multi-workgroup trend fit
    double* d_lsf = nullptr;      // log(size factors) of the current IRLS call (grown on demand)
    void* d_sum = nullptr;        // workspace of the adjusted-p-value kernels (grown on demand)
    size_t sum_cap = 0, sum_sort_bytes = 0;
    size_t lsf_cap = 0;
    void* stage[2] = {nullptr, nullptr};  // page-locked staging chunks of dsq_upload_counts_i32
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    void* d_stage16[2] = {nullptr, nullptr};  // device side of a chunk that travels as uint16
    const int32_t* d_irls_hint = nullptr;  // dsq_irls_order_hint: iteration counts of an earlier fit (one-shot)
    int irls_hint_genes = 0;
    void (*alpha_hook)(void*) = nullptr;  // dsq_set_alpha_hook (one-shot)
    void* alpha_hook_arg = nullptr;
    int deferred = 0;             // dsq_set_deferred: second passes of small batches enqueued without a host round trip
    int32_t* h_pin = nullptr;     // 64 KiB of page-locked host memory: counters read back / small arguments sent (ints [0, 16): counters;
                                  // [16, 3072) + [4096, 16384): ridge / contrast staging up to p = 48; [3072, 4096): trend outputs)
    void* d_ws = nullptr;         // workspace of the rare second-pass kernels (grown on demand, never shrunk)
    size_t ws_cap = 0;
    void* d_resume = nullptr;     // parked optimiser states + gene list of the two-phase dispersion launch (grow-only)
    size_t resume_cap = 0;
    void* d_mix = nullptr;        // slot-ordered copies (counts, mu_hat) of a mixed-design call whose caller bound none (grow-only)
    size_t mix_cap = 0;
    const uint16_t* bind_ys = nullptr;  // dsq_mix_bind (one-shot: the next dispersion / IRLS fit consumes it)
    const uint8_t* bind_big = nullptr;
    const double* bind_mu = nullptr;
    int bind_G = 0;               // genes the bound copies were built for (dsq_mix_bind2; 0: not stated)
    void* d_mixw = nullptr;       // slot-ordered per-sample vectors of the mixed-design IRLS kernel (grow-only)
    size_t mixw_cap = 0;
    int32_t* d_redo = nullptr;    // genes the buffer-less robust-dispersion kernel hands back (side stream; grow-only)
    size_t redo_cap = 0;
    // An LFC fit in two launches (dsq_lfc_fork_begin / dsq_lfc_set_part): the genes whose MAP dispersion is final after the
    // dispersion stage's full-size launch are fitted on lfc_stream while the main stream runs that stage's latency-bound tail
    hipStream_t lfc_stream = nullptr, lfc_return = nullptr;
    hipEvent_t ev_lfc_fork = nullptr, ev_lfc_done = nullptr, ev_lfc_part = nullptr;
    const uint8_t* lfc_part = nullptr;  // dsq_lfc_set_part (one-shot: the next LFC fit consumes it)
    int lfc_want = 0, lfc_phase = 0;
    uint8_t* alpha_conv_late = nullptr;  // dsq_alpha_set_late_flags (one-shot: the next dispersion fit consumes it)
    int lfc_prepared_N = 0, lfc_prepared_P = 0, lfc_prepared_wald = 0;  // dsq_lfc_prepare (one-shot, consumed by phase 1)
    int lfc_pending_G = 0;        // genes of a phase-1 launch whose phase-2 partner (join, rescue of both lists) is still to come
    int32_t* d_lfc_aux = nullptr; // the phase-1 launch's own fallback list and slot order (ctx->d_list serves the main stream)
    size_t lfc_aux_cap = 0;
    dsq_pc::Cache* pc = nullptr;  // device-buffer cache + pool of the Inference-level entry points (dsq_plugin_cache.h)
    struct PluginDesign* designs = nullptr;  // factorised designs (+ mixed-design descriptors) of the last few calls
    void* comm = nullptr;         // ncclComm_t (RCCL), set by dsq_comm_init
    int comm_rank = 0, comm_world = 1;
    std::string err;
};

// frees the plug-in path's designs, device cache and buffer pool of a context (dsq_capi_inf.hip)
void dsq_internal_destroy_plugin(dsq_ctx* ctx);

namespace {
// the one-shot hook of dsq_set_alpha_hook (arg: the context)
void fire_alpha_hook(void* c) {
    dsq_ctx* ctx = (dsq_ctx*)c;
    if (ctx->alpha_hook == nullptr) return;
    void (*fn)(void*) = ctx->alpha_hook;
    ctx->alpha_hook = nullptr;
    fn(ctx->alpha_hook_arg);
}
}  // namespace

namespace {

constexpr size_t kScratchBytes = 32 * 1024;  // doubles [0, 1664): scalars, trend partials and outputs; [1664, 4096): ridge + contrast (48 x 48 + 48)
constexpr int kDeferredMaxGenes = 2048;  // deferred second passes are launched for every gene of the batch

int fail(dsq_ctx* c, int code, const std::string& msg) {
    if (c) {
        c->err = msg;
        // a call that fails before it reaches the fit must not leave the one-shot slot-ordered copies of dsq_mix_bind behind
        // for the next, unrelated call (they belong to the caller's matrix of THIS call)
        c->bind_ys = nullptr;
        c->bind_big = nullptr;
        c->bind_mu = nullptr;
        c->bind_G = 0;
    }
    return code;
}

#define DSQ_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(ctx, DSQ_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
    } while (0)

#define DSQ_CHECK_ARG(cond, msg)                          \
    do {                                                  \
        if (!(cond)) return fail(ctx, DSQ_ERR_ARG, msg);  \
    } while (0)

inline int pad16(int n) { return (n + 15) & ~15; }

hipError_t ensure_list(dsq_ctx* c, size_t n) {
    if (n <= c->list_cap) return hipSuccess;
    if (c->d_list) (void)hipFree(c->d_list);
    c->d_list = nullptr;
    c->list_cap = 0;
    hipError_t e = hipMalloc((void**)&c->d_list, n * sizeof(int32_t));
    if (e == hipSuccess) c->list_cap = n;
    return e;
}

// grow-only device workspace (a hipMalloc / hipFree pair per call costs tens of microseconds and the free
// synchronises the device: the grid-search pass runs in EVERY full-size dispersion launch)
hipError_t ensure_ws(dsq_ctx* c, size_t bytes) {
    if (bytes <= c->ws_cap) return hipSuccess;
    if (c->d_ws) (void)hipFree(c->d_ws);
    c->d_ws = nullptr;
    c->ws_cap = 0;
    const size_t cap = bytes + bytes / 2;
    hipError_t e = hipMalloc(&c->d_ws, cap);
    if (e == hipSuccess) c->ws_cap = cap;
    return e;
}

// dispersion fit + (rare) grid-search second pass
int run_alpha(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx, int N,
              int G, int P, const double* d_alpha_hat, double min_disp, double max_disp, double prior_var,
              int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_conv, int32_t* d_nfev,
              double* d_nll_const = nullptr, int const_mode = DSQ_CONST_COMPUTE,
              const dsq::AlphaExtras* extras = nullptr, int optimizer = 0);

// Householder QR of the N x P design (row-major) -> Xt [P][ldx], pinvXt [P][ldx] =
// rows of R^-1 Q^T (the reference's beta_init = solve(R, Q^T y), utils.py:350-352, and
// sklearn's least-squares fit, utils.py:711-713 / 846-848), full_rank flag
// (numpy.linalg.matrix_rank(X) == P, utils.py:349).
void design_factor(const double* X, int N, int P, int ldx, std::vector<double>& Xt,
                   std::vector<double>& pinvXt, int& full_rank) {
    Xt.assign((size_t)P * ldx, 0.0);
    pinvXt.assign((size_t)P * ldx, 0.0);
    for (int n = 0; n < N; ++n)
        for (int j = 0; j < P; ++j) Xt[(size_t)j * ldx + n] = X[(size_t)n * P + j];
    // A (column-major copy), Q accumulated explicitly as N x P (thin)
    std::vector<double> A((size_t)N * P), R((size_t)P * P, 0.0);
    for (int n = 0; n < N; ++n)
        for (int j = 0; j < P; ++j) A[(size_t)j * N + n] = X[(size_t)n * P + j];
    // modified Gram-Schmidt with re-orthogonalisation (P <= 12, N >> P): Q in A, R upper
    for (int j = 0; j < P; ++j) {
        double* aj = &A[(size_t)j * N];
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < j; ++i) {
                const double* qi = &A[(size_t)i * N];
                long double s = 0.0L;
                for (int n = 0; n < N; ++n) s += (long double)qi[n] * aj[n];
                const double sd = (double)s;
                R[(size_t)i * P + j] += sd;
                for (int n = 0; n < N; ++n) aj[n] -= sd * qi[n];
            }
        }
        long double nn = 0.0L;
        for (int n = 0; n < N; ++n) nn += (long double)aj[n] * aj[n];
        const double nrm = std::sqrt((double)nn);
        R[(size_t)j * P + j] = nrm;
        if (nrm > 0.0)
            for (int n = 0; n < N; ++n) aj[n] /= nrm;
    }
    double rmax = 0.0;
    for (int j = 0; j < P; ++j) rmax = std::fmax(rmax, std::fabs(R[(size_t)j * P + j]));
    full_rank = 1;
    const double tol = rmax * (double)(N > P ? N : P) * 2.220446049250313e-16;
    for (int j = 0; j < P; ++j)
        if (!(std::fabs(R[(size_t)j * P + j]) > tol)) full_rank = 0;
    if (!full_rank) return;
    // pinv = R^-1 Q^T : back substitution per sample
    for (int n = 0; n < N; ++n) {
        double b[DSQ_MAX_P];
        for (int j = 0; j < P; ++j) b[j] = A[(size_t)j * N + n];
        for (int i = P - 1; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < P; ++k) s -= R[(size_t)i * P + k] * b[k];
            b[i] = s / R[(size_t)i * P + i];
        }
        for (int j = 0; j < P; ++j) pinvXt[(size_t)j * ldx + n] = b[j];
    }
}

int run_alpha(dsq_ctx* ctx, const int32_t* d_y, const double* d_mu, int ldn, const double* d_Xt, int ldx, int N,
              int G, int P, const double* d_alpha_hat, double min_disp, double max_disp, double prior_var,
              int cr_reg, int prior_reg, double* d_alpha, uint8_t* d_conv, int32_t* d_nfev,
              double* d_nll_const, int const_mode, const dsq::AlphaExtras* extras, int optimizer) {
    // optimizer: 0 = "L-BFGS-B" (the reference's default and the only one dds.py / ds.py use), 1 = "BFGS" (utils.py:546-554)
    if (G <= 0) return DSQ_OK;
    DSQ_CHECK_ARG(optimizer == 0 || optimizer == 1, "optimizer: 0 (L-BFGS-B) or 1 (BFGS)");
    struct Unbind {  // dsq_mix_bind is one-shot: whatever this call does with it, the next one starts unbound
        dsq_ctx* c;
        ~Unbind() { c->bind_ys = nullptr; c->bind_big = nullptr; c->bind_mu = nullptr; c->bind_G = 0; c->alpha_conv_late = nullptr; }
    } unbind{ctx};
    if (ctx->bind_G != 0 && ctx->bind_G != G) {  // copies of another matrix (a failed or skipped call left them): not ours
        ctx->bind_ys = nullptr; ctx->bind_big = nullptr; ctx->bind_mu = nullptr; ctx->bind_G = 0;
    }
    DSQ_HIP(ensure_list(ctx, (size_t)G));
    // [0] grid-search genes, [1] gene queue of the row kernel, [2] parked genes, [3] gene queue of the continuation launch
    int32_t* d_cnt = ctx->d_counter + 4;
    DSQ_HIP(hipMemsetAsync(d_cnt, 0, 4 * sizeof(int32_t), ctx->stream));
    // two-phase launch (dsq_launch.h, AlphaExtras): parking space for the genes phase A does not finish
    dsq::AlphaExtras ex2{};
    if (extras != nullptr) ex2 = *extras;
    if (optimizer == 0) {
        const size_t need = dsq::alpha_resume_bytes(G) + (size_t)G * sizeof(int32_t) + 256;
        if (need > ctx->resume_cap) {
            if (ctx->d_resume) (void)hipFree(ctx->d_resume);
            ctx->d_resume = nullptr; ctx->resume_cap = 0;
            DSQ_HIP(hipMalloc(&ctx->d_resume, need + need / 4));
            ctx->resume_cap = need + need / 4;
        }
        static const int cap_env = getenv("DSQ_ALPHA_EVAL_CAP") ? atoi(getenv("DSQ_ALPHA_EVAL_CAP")) : 0;  // A/B switch
        ex2.eval_cap = cap_env > 0 ? cap_env : dsq::kAlphaEvalCap;
        ex2.resume_state = ctx->d_resume;
        ex2.resume_list = (int32_t*)((char*)ctx->d_resume + ((dsq::alpha_resume_bytes(G) + 255) & ~(size_t)255));
        ex2.resume_count = d_cnt + 2;
        ex2.mid_hook = ctx->alpha_hook != nullptr ? fire_alpha_hook : nullptr;
        ex2.mid_arg = ctx;
        ex2.conv_late = ctx->alpha_conv_late;
        if (ex2.mix != nullptr && ex2.rows != nullptr && ex2.n_rows > 0) {
            if (!dsq::alpha_mix_fits(*ex2.mix)) {
                ex2.mix = nullptr;  // rows too long for that kernel: the general one takes every gene
            } else {
                // the kernel streams the counts and mu_hat from slot-ordered copies: the caller's (dsq_mix_bind), or built
                // here (plug-in entry points; mu_hat from the caller's matrix or from the IRLS coefficients)
                const size_t Ns = (size_t)ex2.mix->Ns;
                const uint16_t* ys = ctx->bind_ys;
                const double* mus = ctx->bind_mu;
                size_t need = 256;
                if (mus == nullptr) need += (size_t)G * Ns * sizeof(double);
                if (ys == nullptr) need += (size_t)G * Ns * sizeof(uint16_t) + (size_t)G + 256;
                if ((mus == nullptr || ys == nullptr) && need > ctx->mix_cap) {
                    if (ctx->d_mix) (void)hipFree(ctx->d_mix);
                    ctx->d_mix = nullptr; ctx->mix_cap = 0;
                    DSQ_HIP(hipMalloc(&ctx->d_mix, need));
                    ctx->mix_cap = need;
                }
                char* w = (char*)ctx->d_mix;
                if (mus == nullptr) {
                    double* t = (double*)w;
                    w += (size_t)G * Ns * sizeof(double);
                    if (d_mu != nullptr)
                        DSQ_HIP(dsq::launch_mix_f64_to_slots(ctx->stream, d_mu, ldn, *ex2.mix, G, t));
                    else if (ex2.mix_beta != nullptr && ex2.sf != nullptr)
                        DSQ_HIP(dsq::launch_mix_mu_slots(ctx->stream, ex2.mix_beta, ex2.sf, *ex2.mix, G, t));
                    else
                        return fail(ctx, DSQ_ERR_ARG, "mixed-design dispersion fit: no mu_hat (matrix, bound slots or beta)");
                    mus = t;
                }
                if (ys == nullptr) {
                    uint16_t* t = (uint16_t*)w;
                    DSQ_HIP(dsq::launch_mix_counts_to_slots(ctx->stream, d_y, ldn, *ex2.mix, G, t, (uint8_t*)(t + (size_t)G * Ns)));
                    ys = t;
                }
                ex2.mix_ys = ys;
                ex2.mix_mu = mus;
            }
        }
        if (ex2.mix == nullptr && d_mu == nullptr && ex2.mix_beta != nullptr)
            return fail(ctx, DSQ_ERR_ARG, "mu_hat from IRLS coefficients needs the mixed-design kernel (rows too long)");
        extras = &ex2;
    }
    DSQ_HIP(hipEventRecord(ctx->evk0, ctx->stream));
    if (optimizer == 1) {
        DSQ_CHECK_ARG(d_mu != nullptr && P <= DSQ_BFGS_MAX_P,
                      "optimizer=\"BFGS\" takes mu_hat as a matrix and designs of at most 12 columns");
        DSQ_HIP(dsq::launch_alpha_bfgs(ctx->stream, d_y, d_mu, ldn, d_Xt, ldx, N, G, P, d_alpha_hat, min_disp,
                                       max_disp, prior_var, cr_reg, prior_reg, d_alpha, d_conv, d_nfev, d_cnt,
                                       ctx->d_list));
        extras = nullptr;  // (the grid pass below then reads the mu_hat matrix)
    } else {
        DSQ_HIP(dsq::launch_alpha(ctx->stream, d_y, d_mu, ldn, d_Xt, ldx, N, G, P, d_alpha_hat, min_disp, max_disp,
                                  prior_var, cr_reg, prior_reg, d_alpha, d_conv, d_nfev, d_cnt, ctx->d_list,
                                  d_nll_const, const_mode, extras, d_cnt + 1));
    }
    DSQ_HIP(hipEventRecord(ctx->evk1, ctx->stream));
    fire_alpha_hook(ctx);  // (a route that did not pass the hook's point: now)
    int32_t* h_cnt = ctx->h_pin + 1;
    // Deferred mode (dsq_set_deferred; small batches on the register kernels): the grid-search pass is enqueued for
    // ALL G genes as a capacity and the kernels read the number of fallback genes from the device - no host round trip
    // between the fit and its second pass (the refit of the outlier genes is a chain of ~15 tiny launches whose
    // synchronisations cost more than its kernels).
    const bool deferred = ctx->deferred && G <= kDeferredMaxGenes && optimizer == 0 &&
                          !dsq::alpha_is_wide(P, extras != nullptr ? extras->cells.C : 0);
    const int32_t* n_dev = deferred ? d_cnt : nullptr;
    int32_t n_grid = G;
    if (!deferred) {
        DSQ_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));
        DSQ_HIP(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->evk0, ctx->evk1));
        n_grid = *h_cnt;
        ctx->last_n_grid = n_grid;
    } else {
        ctx->last_kernel_ms = -1.0f;  // not measured: nobody waited for the launch
        ctx->last_n_grid = -1;
    }
    if (n_grid > 0) {
        // everything below is stream-ordered behind the launch above and ahead of whatever the caller enqueues
        // next: no host synchronisation, no allocation (workspace carved from ctx->d_ws)
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const bool from_cells = extras != nullptr && extras->cell_mu != nullptr;
        const bool from_beta = extras != nullptr && extras->mix_beta != nullptr && d_mu == nullptr;
        const bool rebuild = extras != nullptr && (extras->coef != nullptr || from_cells || from_beta);
        const size_t b_work = up((size_t)n_grid * dsq::kAlphaGridWorkDoubles * sizeof(double));
        const size_t b_mu = rebuild ? up((size_t)n_grid * ldn * sizeof(double)) : 0;
        const size_t b_idx = rebuild ? up((size_t)n_grid * sizeof(int32_t)) : 0;
        DSQ_HIP(ensure_ws(ctx, b_work + b_mu + b_idx));
        char* w = (char*)ctx->d_ws;
        double* work = (double*)w;
        if (rebuild) {
            // no N x G mu_hat exists: rebuild the rows of the (few) fallback genes, compacted; the grid kernels read the
            // counts and write the result through the list
            double* musub = (double*)(w + b_work);
            int32_t* idx = (int32_t*)(w + b_work + b_mu);
            if (from_beta)
                DSQ_HIP(dsq::launch_mu_from_beta(ctx->stream, extras->mix_beta, extras->sf, d_Xt, ldx, N, P, ctx->d_list,
                                                 n_grid, musub, ldn, idx, n_dev));
            else if (from_cells)
                DSQ_HIP(dsq::launch_mu_from_cells(ctx->stream, extras->cell_mu, extras->cells.C, extras->sf,
                                                  extras->cells.cell_of, N, ctx->d_list, n_grid, musub, ldn, idx, n_dev));
            else
                DSQ_HIP(dsq::launch_mu_from_coef(ctx->stream, extras->coef, extras->sf, d_Xt, ldx, N, P, extras->min_mu,
                                                 ctx->d_list, n_grid, musub, ldn, idx, n_dev));
            DSQ_HIP(dsq::launch_alpha_grid(ctx->stream, d_y, musub, ldn, d_Xt, ldx, N, P, min_disp, max_disp, d_alpha,
                                           ctx->d_list, n_grid, work, n_dev, true));
        } else {
            DSQ_HIP(dsq::launch_alpha_grid(ctx->stream, d_y, d_mu, ldn, d_Xt, ldx, N, P, min_disp, max_disp, d_alpha,
                                           ctx->d_list, n_grid, work, n_dev));
        }
    }
    return DSQ_OK;
}

}  // namespace


namespace {
constexpr int kIrlsOrderMinGenes = 1024;  // below: a few workgroups, nothing to balance
bool irls_order_enabled() {
    static const bool v = getenv("DSQ_NO_IRLS_ORDER") == nullptr;  // A/B switch
    return v;
}
}  // namespace

namespace {
int run_irls(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
             const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank, const double* d_disp,
             double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter,
             double* d_beta, double* d_mu, double* d_hat, uint8_t* d_converged, int32_t* d_iters,
             const dsq::IrlsExtras* extras, int optimizer = 0) {
    // optimizer of the rescue of diverged genes (utils.py:343, 389-399): 0 = bounded L-BFGS-B (default), 1 = BFGS
    if (G <= 0) return DSQ_OK;
    DSQ_CHECK_ARG(optimizer == 0 || optimizer == 1, "optimizer: 0 (L-BFGS-B) or 1 (BFGS)");
    struct Unbind {  // (see run_alpha)
        dsq_ctx* c;
        ~Unbind() { c->bind_ys = nullptr; c->bind_big = nullptr; c->bind_mu = nullptr; c->bind_G = 0; }
    } unbind{ctx};
    if (ctx->bind_G != 0 && ctx->bind_G != G) {  // (see run_alpha)
        ctx->bind_ys = nullptr; ctx->bind_big = nullptr; ctx->bind_mu = nullptr; ctx->bind_G = 0;
    }
    dsq::IrlsExtras ex_local{};
    if (extras != nullptr) ex_local = *extras;
    const dsq::MixDesign* const extras_in_mix = ex_local.mix;
    ex_local.optimizer = optimizer;
    DSQ_CHECK_ARG(optimizer == 0 || P <= DSQ_BFGS_MAX_P, "optimizer=\"BFGS\": designs of at most 12 columns");
    extras = &ex_local;
    // A fit in two launches (dsq_lfc_set_part, one-shot).  Phase 1: the genes with part[g] == want, on the current stream
    // (the caller forked: dsq_lfc_fork_begin), with a fallback list, counters and slot order of its own - no host round trip,
    // no rescue.  Phase 2: the other genes; then the join with phase 1, ONE synchronisation, the rescue of both lists.
    const uint8_t* const part = ctx->lfc_part;
    const int part_want = ctx->lfc_want, phase = part != nullptr ? ctx->lfc_phase : 0;
    ctx->lfc_part = nullptr; ctx->lfc_phase = 0;
    if (phase != 0) {
        DSQ_CHECK_ARG(phase == 1 || phase == 2, "dsq_lfc_set_part: phase 1 or 2");
        DSQ_CHECK_ARG(optimizer == 0 && !ctx->deferred &&
                          dsq::irls_takes_parts(N, P, ex_local.cells.C, ex_local.mix, full_rank),
                      "a fit in two launches: the default rescue optimiser, not deferred, a design the part-aware kernels take");
        DSQ_CHECK_ARG(phase == 1 ? ctx->lfc_pending_G == 0 : ctx->lfc_pending_G == G,
                      "a fit in two launches: phase 2 follows the phase 1 of the same genes");
        ex_local.part = part;
        ex_local.part_want = part_want;
        ex_local.part_shared_ready = phase == 2 ? 1 : 0;
    }
    // sixteen-lane kernel: slots ordered by the predicted number of sweeps (the list lives behind the fallback list)
    const bool ordered = G >= kIrlsOrderMinGenes && dsq::irls_takes_rows(N, P, ex_local.cells.C) && irls_order_enabled();
    const size_t list_ints = (size_t)G * (ordered ? 2 : 1) + (ordered ? (size_t)dsq::irls_order_work_ints() : 0);
    int32_t* fb_list = nullptr;
    int32_t* fb_count = ctx->d_counter;  // [0] fallback genes, [1] gene queue
    if (phase == 1) {
        if (list_ints > ctx->lfc_aux_cap) {
            if (ctx->d_lfc_aux) (void)hipFree(ctx->d_lfc_aux);
            ctx->d_lfc_aux = nullptr; ctx->lfc_aux_cap = 0;
            DSQ_HIP(hipMalloc((void**)&ctx->d_lfc_aux, list_ints * sizeof(int32_t)));
            ctx->lfc_aux_cap = list_ints;
        }
        fb_list = ctx->d_lfc_aux;
        fb_count = ctx->d_counter + 8;   // [8] fallback genes, [9] gene queue of the phase-1 launch
    } else {
        DSQ_HIP(ensure_list(ctx, list_ints));
        fb_list = ctx->d_list;
    }
    if (ordered) {
        int32_t* d_order = fb_list + G;
        DSQ_HIP(dsq::launch_irls_order(ctx->stream, d_disp, ctx->irls_hint_genes == G ? ctx->d_irls_hint : nullptr, G,
                                       d_order, d_order + G));
        ex_local.order = d_order;
    }
    ctx->d_irls_hint = nullptr; ctx->irls_hint_genes = 0;  // one-shot
    // (dsq_lfc_prepare did this launch's small operations on the main stream, ahead of the dispersion stage)
    const bool prepared = phase == 1 && ctx->lfc_prepared_N == N && ctx->lfc_prepared_P == P;
    ctx->lfc_prepared_N = 0; ctx->lfc_prepared_P = 0; ctx->lfc_prepared_wald = 0;
    if (phase != 2 && !prepared) {  // (phase 2: the vector of phase 1 - the same size factors - which that launch may still be reading)
        if ((size_t)N > ctx->lsf_cap) {
            if (ctx->d_lsf) (void)hipFree(ctx->d_lsf);
            ctx->d_lsf = nullptr; ctx->lsf_cap = 0;
            DSQ_HIP(hipMalloc((void**)&ctx->d_lsf, (size_t)N * sizeof(double)));
            ctx->lsf_cap = (size_t)N;
        }
        DSQ_HIP(dsq::launch_log_vec(ctx->stream, d_sf, N, ctx->d_lsf));
    }
    if (!prepared) DSQ_HIP(hipMemsetAsync(fb_count, 0, 2 * sizeof(int32_t), ctx->stream));
    if (ex_local.mix != nullptr && dsq::irls_takes_mix(ex_local.mix, full_rank)) {
        const int n_layers = ((ex_local.flags != nullptr && ex_local.cooks != nullptr) ? 1 : 0) + (d_mu != nullptr ? 1 : 0) +
                             (d_hat != nullptr ? 1 : 0);
        const size_t need = dsq::irls_mix_work_bytes(*ex_local.mix, G, n_layers);
        if (need > ctx->mixw_cap) {
            if (ctx->d_mixw) (void)hipFree(ctx->d_mixw);
            ctx->d_mixw = nullptr; ctx->mixw_cap = 0;
            DSQ_HIP(hipMalloc(&ctx->d_mixw, need));
            ctx->mixw_cap = need;
        }
        ex_local.mix_work = ctx->d_mixw;
        ex_local.mix_work_bytes = ctx->mixw_cap;
        ex_local.mix_queue = fb_count + 1;
        // the counts in slot order: the caller's copy (dsq_mix_bind) or one built here
        ex_local.mix_ys = ctx->bind_ys;
        ex_local.mix_big = ctx->bind_big;
        if (ex_local.mix_ys == nullptr || ex_local.mix_big == nullptr) {
            const size_t Ns = (size_t)ex_local.mix->Ns;
            const size_t need_s = (size_t)G * Ns * sizeof(uint16_t) + (size_t)G + 256;
            if (need_s > ctx->mix_cap) {
                if (ctx->d_mix) (void)hipFree(ctx->d_mix);
                ctx->d_mix = nullptr; ctx->mix_cap = 0;
                DSQ_HIP(hipMalloc(&ctx->d_mix, need_s));
                ctx->mix_cap = need_s;
            }
            uint16_t* t = (uint16_t*)ctx->d_mix;
            DSQ_HIP(dsq::launch_mix_counts_to_slots(ctx->stream, d_y, ldn, *ex_local.mix, G, t, (uint8_t*)(t + (size_t)G * Ns)));
            ex_local.mix_ys = t;
            ex_local.mix_big = (const uint8_t*)(t + (size_t)G * Ns);
        }
    } else {
        ex_local.mix = nullptr;
    }
    DSQ_HIP(dsq::launch_irls(ctx->stream, d_y, ldn, d_sf, ctx->d_lsf, d_Xt, d_pinvXt, ldx, N, G, P, full_rank, d_disp,
                             min_mu, beta_tol, min_beta, max_beta, maxiter, d_beta, d_mu, d_hat,
                             d_converged, d_iters, fb_count, fb_list, extras));
    if (phase == 1) {  // its partner joins, synchronises and rescues
        ctx->lfc_pending_G = G;
        return DSQ_OK;
    }
    int32_t* h_cnt = ctx->h_pin;
    // deferred mode (see run_alpha): the rescue pass is enqueued for all G genes as a capacity, count on the device
    const bool deferred = ctx->deferred && G <= kDeferredMaxGenes && optimizer == 0 &&
                          !dsq::irls_is_wide(P, extras->cells.C);
    int32_t n_fb = G, n_fb1 = 0;
    if (phase == 2) {  // behind the phase-1 launch from here on (dsq_lfc_fork_end recorded its end)
        ctx->lfc_pending_G = 0;
        DSQ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_lfc_done, 0));
        DSQ_HIP(hipMemcpyAsync(h_cnt + 2, ctx->d_counter + 8, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!deferred) {
        DSQ_HIP(hipMemcpyAsync(h_cnt, ctx->d_counter, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));
        n_fb = *h_cnt;
        if (phase == 2) n_fb1 = h_cnt[2];
    }
    ex_local.part = nullptr;  // (the rescue kernels run from the lists)
    for (int leg = 0; leg < 2; ++leg) {
        const int32_t n_leg = leg == 0 ? n_fb : n_fb1;
        const int32_t* list_leg = leg == 0 ? ctx->d_list : ctx->d_lfc_aux;
        if (n_leg <= 0) continue;  // stream-ordered ahead of the caller's next work: no second synchronisation
        if (ex_local.cooks_ld != 0 && ex_local.cooks != nullptr && ex_local.flags != nullptr) {
            // slot-ordered Cook's layer (mixed designs): the general rescue kernels write sample order - into scratch rows
            // that launch_irls_rescue scatters through MixDesign::slot_of
            DSQ_HIP(ensure_ws(ctx, (size_t)(n_fb > n_fb1 ? n_fb : n_fb1) * ldn * sizeof(double)));
            ex_local.cooks_tmp = (double*)ctx->d_ws;
            ex_local.mix = extras_in_mix;
        }
        DSQ_HIP(dsq::launch_irls_rescue(ctx->stream, d_y, ldn, d_sf, ctx->d_lsf, d_Xt, d_pinvXt, ldx, N, P, full_rank,
                                        d_disp, min_mu, beta_tol, min_beta, max_beta, maxiter, d_beta, d_mu,
                                        d_hat, d_converged, d_iters, list_leg, n_leg, extras,
                                        (deferred && leg == 0) ? ctx->d_counter : nullptr));
    }
    return DSQ_OK;
}
}  // namespace

namespace {
dsq::CellDesign to_cells(const dsq_cells* c) {
    dsq::CellDesign d{};
    if (c != nullptr && c->n_cells > 0) { d.cell_of = c->d_cell_of; d.Xc = c->d_Xc; d.XX = c->d_XX; d.C = c->n_cells; }
    return d;
}
}  // namespace
