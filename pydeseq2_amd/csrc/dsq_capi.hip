// dsq_capi.hip — the C ABI of libdeseq_hip.so (see include/deseq_hip.h).
// Host-side glue only: context, device memory, argument checks, staging of host arrays
// into the gene-major device layout, kernel launches, copy-back.  No math lives here.
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

// every host-side wait of this library is counted (dsq_host_sync_count: bench.py reports host synchronisations per step)
static std::atomic<unsigned long long> g_dsq_host_syncs{0};
#define hipStreamSynchronize(s) (++g_dsq_host_syncs, (hipStreamSynchronize)(s))
#define hipEventSynchronize(e) (++g_dsq_host_syncs, (hipEventSynchronize)(e))

#include "../../include/deseq_hip.h"
#include "dsq_launch.h"
#include "dsq_plugin_cache.h"

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
    double* d_scratch = nullptr;  // 8 KiB of device scratch (scalars, trend partials)
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
    int32_t* h_pin = nullptr;     // 16 KiB of page-locked host memory: counters read back / small arguments sent
    void* d_ws = nullptr;         // workspace of the rare second-pass kernels (grown on demand, never shrunk)
    size_t ws_cap = 0;
    void* d_resume = nullptr;     // parked optimiser states + gene list of the two-phase dispersion launch (grow-only)
    size_t resume_cap = 0;
    void* d_mix = nullptr;        // slot-ordered copies (counts, mu_hat) of a mixed-design call whose caller bound none (grow-only)
    size_t mix_cap = 0;
    const uint16_t* bind_ys = nullptr;  // dsq_mix_bind (one-shot: the next dispersion / IRLS fit consumes it)
    const uint8_t* bind_big = nullptr;
    const double* bind_mu = nullptr;
    void* d_mixw = nullptr;       // slot-ordered per-sample vectors of the mixed-design IRLS kernel (grow-only)
    size_t mixw_cap = 0;
    int32_t* d_redo = nullptr;    // genes the buffer-less robust-dispersion kernel hands back (side stream; grow-only)
    size_t redo_cap = 0;
    dsq_pc::Cache* pc = nullptr;  // device-buffer cache + pool of the Inference-level entry points (dsq_plugin_cache.h)
    struct PluginDesign* designs = nullptr;  // factorised designs (+ mixed-design descriptors) of the last few calls
    void* comm = nullptr;         // ncclComm_t (RCCL), set by dsq_comm_init
    int comm_rank = 0, comm_world = 1;
    std::string err;
};

static void destroy_plugin_designs(dsq_ctx* ctx);  // (defined with the Inference-level entry points)

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

constexpr size_t kScratchBytes = 16 * 1024;
constexpr int kDeferredMaxGenes = 2048;  // deferred second passes are launched for every gene of the batch

int fail(dsq_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
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
        ~Unbind() { c->bind_ys = nullptr; c->bind_big = nullptr; c->bind_mu = nullptr; }
    } unbind{ctx};
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

extern "C" {

// ------------------------------------------------------------------ context
int dsq_create(int device_id, dsq_ctx** out) {
    if (!out) return DSQ_ERR_ARG;
    *out = nullptr;
    dsq_ctx* ctx = new dsq_ctx();
    ctx->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evk0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evk1);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_scratch, kScratchBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counter, 64);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_pin, 16384, hipHostMallocDefault);
    if (e != hipSuccess) {
        fprintf(stderr, "dsq_create: %s\n", hipGetErrorString(e));
        delete ctx;
        return DSQ_ERR_HIP;
    }
    *out = ctx;
    return DSQ_OK;
}

int dsq_set_deferred(dsq_ctx* ctx, int on) {
    ctx->deferred = on ? 1 : 0;
    return DSQ_OK;
}

void dsq_destroy(dsq_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_counter) (void)hipFree(ctx->d_counter);
    if (ctx->d_list) (void)hipFree(ctx->d_list);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_resume) (void)hipFree(ctx->d_resume);
    if (ctx->d_mix) (void)hipFree(ctx->d_mix);
    if (ctx->d_mixw) (void)hipFree(ctx->d_mixw);
    if (ctx->d_redo) (void)hipFree(ctx->d_redo);
    if (ctx->pc) {
        dsq_pc::destroy(*ctx->pc);
        delete ctx->pc;
        ctx->pc = nullptr;
    }
    destroy_plugin_designs(ctx);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->d_lsf) (void)hipFree(ctx->d_lsf);
    if (ctx->d_trend_grid) (void)hipFree(ctx->d_trend_grid);
    if (ctx->d_sum) (void)hipFree(ctx->d_sum);
    for (int k = 0; k < 2; ++k) {
        if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
        if (ctx->d_stage16[k]) (void)hipFree(ctx->d_stage16[k]);
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    }
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->small_stream) (void)hipStreamDestroy(ctx->small_stream);
    if (ctx->ev_small0) (void)hipEventDestroy(ctx->ev_small0);
    if (ctx->ev_small1) (void)hipEventDestroy(ctx->ev_small1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evk0) (void)hipEventDestroy(ctx->evk0);
    if (ctx->evk1) (void)hipEventDestroy(ctx->evk1);
    if (ctx->main_stream) (void)hipStreamDestroy(ctx->main_stream);
    else if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* dsq_last_error(const dsq_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dsq_device_info(dsq_ctx* ctx, char* name, int name_len, int* cu_count, size_t* mem_bytes, char* arch,
                    int arch_len) {
    hipDeviceProp_t p;
    DSQ_HIP(hipGetDeviceProperties(&p, ctx->device));
    if (name && name_len > 0) { strncpy(name, p.name, name_len - 1); name[name_len - 1] = 0; }
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (mem_bytes) *mem_bytes = p.totalGlobalMem;
    return DSQ_OK;
}

int dsq_sync(dsq_ctx* ctx) {
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_last_alpha_kernel(dsq_ctx* ctx, float* kernel_ms, int* n_grid_fallback) {
    if (kernel_ms) *kernel_ms = ctx->last_kernel_ms;
    if (n_grid_fallback) *n_grid_fallback = ctx->last_n_grid;
    return DSQ_OK;
}

// developer aid: name of the thread's pending (unconsumed) HIP error, "" if none; clears it
const char* dsq_debug_pending_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? "" : hipGetErrorString(e);
}

int dsq_timer_start(dsq_ctx* ctx) {
    DSQ_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return DSQ_OK;
}

int dsq_timer_stop(dsq_ctx* ctx, float* ms) {
    DSQ_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    DSQ_HIP(hipEventSynchronize(ctx->ev1));
    DSQ_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return DSQ_OK;
}

// ------------------------------------------------------------------ memory
int dsq_malloc(dsq_ctx* ctx, size_t bytes, void** dptr) {
    DSQ_CHECK_ARG(dptr != nullptr, "dsq_malloc: null out pointer");
    DSQ_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e == hipErrorOutOfMemory) return fail(ctx, DSQ_ERR_NOMEM, "hipMalloc: out of device memory");
    DSQ_HIP(e);
    return DSQ_OK;
}
int dsq_free(dsq_ctx* ctx, void* dptr) {
    if (dptr) DSQ_HIP(hipFree(dptr));
    return DSQ_OK;
}
int dsq_memset(dsq_ctx* ctx, void* dptr, int value, size_t bytes) {
    DSQ_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return DSQ_OK;
}
int dsq_h2d(dsq_ctx* ctx, void* dst, const void* src, size_t bytes) {
    DSQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}
int dsq_d2h(dsq_ctx* ctx, void* dst, const void* src, size_t bytes) {
    DSQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}
int dsq_h2d_2d(dsq_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t row_bytes,
               size_t rows) {
    DSQ_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}
int dsq_d2h_2d(dsq_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t row_bytes,
               size_t rows) {
    DSQ_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

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

namespace {
constexpr int kIrlsOrderMinGenes = 1024;  // below: a few workgroups, nothing to balance
bool irls_order_enabled() {
    static const bool v = getenv("DSQ_NO_IRLS_ORDER") == nullptr;  // A/B switch
    return v;
}
}  // namespace

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
        ~Unbind() { c->bind_ys = nullptr; c->bind_big = nullptr; c->bind_mu = nullptr; }
    } unbind{ctx};
    dsq::IrlsExtras ex_local{};
    if (extras != nullptr) ex_local = *extras;
    const dsq::MixDesign* const extras_in_mix = ex_local.mix;
    ex_local.optimizer = optimizer;
    DSQ_CHECK_ARG(optimizer == 0 || P <= DSQ_BFGS_MAX_P, "optimizer=\"BFGS\": designs of at most 12 columns");
    extras = &ex_local;
    // sixteen-lane kernel: slots ordered by the predicted number of sweeps (the list lives behind the fallback list)
    const bool ordered = G >= kIrlsOrderMinGenes && dsq::irls_takes_rows(N, P, ex_local.cells.C) && irls_order_enabled();
    DSQ_HIP(ensure_list(ctx, (size_t)G * (ordered ? 2 : 1) + (ordered ? (size_t)dsq::irls_order_work_ints() : 0)));
    if (ordered) {
        int32_t* d_order = ctx->d_list + G;
        DSQ_HIP(dsq::launch_irls_order(ctx->stream, d_disp, ctx->irls_hint_genes == G ? ctx->d_irls_hint : nullptr, G,
                                       d_order, d_order + G));
        ex_local.order = d_order;
    }
    ctx->d_irls_hint = nullptr; ctx->irls_hint_genes = 0;  // one-shot
    if ((size_t)N > ctx->lsf_cap) {
        if (ctx->d_lsf) (void)hipFree(ctx->d_lsf);
        ctx->d_lsf = nullptr; ctx->lsf_cap = 0;
        DSQ_HIP(hipMalloc((void**)&ctx->d_lsf, (size_t)N * sizeof(double)));
        ctx->lsf_cap = (size_t)N;
    }
    DSQ_HIP(dsq::launch_log_vec(ctx->stream, d_sf, N, ctx->d_lsf));
    DSQ_HIP(hipMemsetAsync(ctx->d_counter, 0, 2 * sizeof(int32_t), ctx->stream));  // [0] fallback genes, [1] gene queue
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
        ex_local.mix_queue = ctx->d_counter + 1;
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
                             d_converged, d_iters, ctx->d_counter, ctx->d_list, extras));
    int32_t* h_cnt = ctx->h_pin;
    // deferred mode (see run_alpha): the rescue pass is enqueued for all G genes as a capacity, count on the device
    const bool deferred = ctx->deferred && G <= kDeferredMaxGenes && optimizer == 0 &&
                          !dsq::irls_is_wide(P, extras->cells.C);
    int32_t n_fb = G;
    if (!deferred) {
        DSQ_HIP(hipMemcpyAsync(h_cnt, ctx->d_counter, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));
        n_fb = *h_cnt;
    }
    if (n_fb > 0) {  // stream-ordered ahead of the caller's next work: no second synchronisation
        if (ex_local.cooks_ld != 0 && ex_local.cooks != nullptr && ex_local.flags != nullptr) {
            // slot-ordered Cook's layer (mixed designs): the general rescue kernels write sample order - into scratch rows
            // that launch_irls_rescue scatters through MixDesign::slot_of
            DSQ_HIP(ensure_ws(ctx, (size_t)n_fb * ldn * sizeof(double)));
            ex_local.cooks_tmp = (double*)ctx->d_ws;
            ex_local.mix = extras_in_mix;
        }
        DSQ_HIP(dsq::launch_irls_rescue(ctx->stream, d_y, ldn, d_sf, ctx->d_lsf, d_Xt, d_pinvXt, ldx, N, P, full_rank,
                                        d_disp, min_mu, beta_tol, min_beta, max_beta, maxiter, d_beta, d_mu,
                                        d_hat, d_converged, d_iters, ctx->d_list, n_fb, extras,
                                        deferred ? ctx->d_counter : nullptr));
    }
    return DSQ_OK;
}
}  // namespace

int dsq_dev_irls(dsq_ctx* ctx, const int32_t* d_y, int ldn, const double* d_sf, const double* d_Xt,
                 const double* d_pinvXt, int ldx, int N, int G, int P, int full_rank, const double* d_disp,
                 double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter,
                 double* d_beta, double* d_mu, double* d_hat, uint8_t* d_converged, int32_t* d_iters) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    return run_irls(ctx, d_y, ldn, d_sf, d_Xt, d_pinvXt, ldx, N, G, P, full_rank, d_disp, min_mu, beta_tol, min_beta,
                    max_beta, maxiter, d_beta, d_mu, d_hat, d_converged, d_iters, nullptr);
}

namespace {
dsq::CellDesign to_cells(const dsq_cells* c) {
    dsq::CellDesign d{};
    if (c != nullptr && c->n_cells > 0) { d.cell_of = c->d_cell_of; d.Xc = c->d_Xc; d.XX = c->d_XX; d.C = c->n_cells; }
    return d;
}
}  // namespace

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
    return DSQ_OK;
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
    DSQ_CHECK_ARG((whole ? N : max_cell) <= 16384, "a design cell with more than 16384 samples is not supported");
    if (whole) min_cell = N;
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

int dsq_side_begin(dsq_ctx* ctx) {
    DSQ_CHECK_ARG(ctx->stream == ctx->main_stream || ctx->main_stream == nullptr, "already on the side stream");
    if (ctx->main_stream == nullptr) ctx->main_stream = ctx->stream;
    if (ctx->side_stream == nullptr) {
        // DSQ_CU_SPLIT=K: the side stream (robust dispersions: fills every CU it may use) is kept off K compute units
        // and the trend / prior kernels - 32 workgroups that synchronise through grid barriers - get a stream that may
        // use only those K: both then run at their stand-alone speed side by side, instead of the barrier kernel's
        // workgroups time-slicing CUs with the other kernel's waves (measured: 0.51 ms alone, 1.32 ms co-scheduled).
        // (c3: 9.10 -> 8.52 ms per step at K = 32, 8.58 at 64, no gain at 16 - two barrier workgroups per CU)
        // (read per context; the 32 was tuned on a 256-CU part: a device with fewer than 4 x split compute units keeps
        // one unmasked side stream - a robust-dispersion kernel squeezed onto cus - 32 units would serialise the stage)
        const int split = getenv("DSQ_CU_SPLIT") ? atoi(getenv("DSQ_CU_SPLIT")) : 32;
        hipDeviceProp_t prop;
        DSQ_HIP(hipGetDeviceProperties(&prop, ctx->device));
        const int cus = prop.multiProcessorCount;
        if (split > 0 && cus >= 4 * split) {
            std::vector<uint32_t> big((size_t)(cus + 31) / 32, 0u), small((size_t)(cus + 31) / 32, 0u);
            // which compute units are reserved: DSQ_CU_SPLIT_MODE 0 = the first `split` mask bits, 1 = every (cus / split)-th
            const int mode = getenv("DSQ_CU_SPLIT_MODE") ? atoi(getenv("DSQ_CU_SPLIT_MODE")) : 0;
            const int stride = cus / split;
            for (int i = 0; i < cus; ++i) {
                const bool res = mode == 0 ? i < split : (i % stride == 0 && i / stride < split);
                (res ? small : big)[(size_t)i / 32] |= 1u << (i % 32);
            }
            DSQ_HIP(hipExtStreamCreateWithCUMask(&ctx->side_stream, (uint32_t)big.size(), big.data()));
            DSQ_HIP(hipExtStreamCreateWithCUMask(&ctx->small_stream, (uint32_t)small.size(), small.data()));
            DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_small0, hipEventDisableTiming));
            DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_small1, hipEventDisableTiming));
        } else {
            DSQ_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        }
        DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        DSQ_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    DSQ_HIP(hipEventRecord(ctx->ev_fork, ctx->main_stream));
    DSQ_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    ctx->stream = ctx->side_stream;
    return DSQ_OK;
}

int dsq_side_end(dsq_ctx* ctx) {
    DSQ_CHECK_ARG(ctx->side_stream != nullptr && ctx->stream == ctx->side_stream, "not on the side stream");
    ctx->stream = ctx->main_stream;  // first: a failing record must not leave the context on the side stream
    DSQ_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
    return DSQ_OK;
}

// Leave the side stream whatever state the context is in (error paths of the caller: a stage failed between
// dsq_side_begin and dsq_side_end) and wait until everything queued on either stream has run, so that buffers the
// side stream was writing may be recycled.  A no-op on a context that never forked.
int dsq_side_abort(dsq_ctx* ctx) {
    if (ctx->main_stream != nullptr) ctx->stream = ctx->main_stream;
    if (ctx->side_stream != nullptr) {
        (void)hipEventRecord(ctx->ev_join, ctx->side_stream);
        DSQ_HIP(hipStreamSynchronize(ctx->side_stream));
    }
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_side_wait(dsq_ctx* ctx) {
    if (ctx->side_stream == nullptr) return DSQ_OK;
    DSQ_HIP(hipStreamWaitEvent(ctx->main_stream, ctx->ev_join, 0));
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
        double* d_ridge = ctx->d_scratch + 16;
        double* d_contrast = d_ridge + P * P;  // (behind the matrix: both travel in one copy)
        // via page-locked memory: the caller's arrays may be temporaries, and a pageable source would make the
        // copy (and the launch behind it) wait for the host
        // (stream-ordered: a rescue kernel of the previous call may still be reading them; the slot itself is free
        // again because every call ends behind a synchronisation that follows its copies)
        double* h_stage = (double*)(ctx->h_pin + 16);
        std::memcpy(h_stage, h_ridge, (size_t)P * P * sizeof(double));
        std::memcpy(h_stage + P * P, h_contrast, (size_t)P * sizeof(double));
        DSQ_HIP(hipMemcpyAsync(d_ridge, h_stage, (size_t)(P * P + P) * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
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
    DSQ_CHECK_ARG(N <= 16384, "more than 16384 samples is not supported by the outlier replacement");
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
    double* d_ridge = ctx->d_scratch + 16;
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
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_SHRINK_MAX_P, "P out of range (apeGLM shrinkage: at most 32 design columns)");
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

int dsq_d2d(dsq_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (bytes) DSQ_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return DSQ_OK;
}
// Host count matrix (int64 as the reference holds it, or int32) -> int32 in HBM, same element order.
// The matrix is cut into chunks of kStageElems elements; a few host threads narrow a chunk into one of two
// page-locked staging buffers (checking 0 <= v < 2^31) while the DMA of the previous chunk is in flight, so
// the PCIe link carries half the bytes of the int64 matrix and never waits for pageable-memory staging.
extern "C++" {
namespace {
constexpr size_t kStageElems = (size_t)8 << 20;  // 32 MiB of int32 per chunk

template <class SrcT>
void narrow_chunk(const SrcT* src, int32_t* dst, size_t n, int n_threads, int* bad) {
    auto work = [=](size_t lo, size_t hi, int* flag) {
        int b = 0;
        for (size_t i = lo; i < hi; ++i) {
            const SrcT v = src[i];
            b |= (v < 0) | ((long long)v > 2147483647LL);
            dst[i] = (int32_t)v;
        }
        if (b) *flag = 1;
    };
    if (n_threads <= 1 || n < ((size_t)1 << 16)) {
        work(0, n, bad);
        return;
    }
    std::vector<std::thread> th;
    std::vector<int> flags((size_t)n_threads, 0);
    const size_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        th.emplace_back(work, lo, hi, &flags[(size_t)t]);
    }
    for (auto& x : th) x.join();
    for (int f : flags)
        if (f) *bad = 1;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
// the same, to uint16 - for a chunk whose counts are all below 65 536 (the usual RNA-seq matrix): the PCIe link, which is
// what bounds the upload (240 MB of int32 at ~25 GB/s of pinned-memory DMA: 10 of the 12 ms), carries a QUARTER of the
// int64 matrix's bytes; the device widens the chunk into its place.  *big is set if some count does not fit (the chunk
// is then narrowed to int32 as before); negative counts set *bad.
template <class SrcT>
void narrow_chunk_u16(const SrcT* src, uint16_t* dst, size_t n, int n_threads, int* bad, int* big) {
    auto work = [=](size_t lo, size_t hi, int* flags) {
        int b = 0, g = 0;
        for (size_t i = lo; i < hi; ++i) {
            const SrcT v = src[i];
            b |= (v < 0);
            g |= ((long long)v > 65535LL);
            dst[i] = (uint16_t)v;
        }
        if (b) flags[0] = 1;
        if (g) flags[1] = 1;
    };
    std::vector<int> flags((size_t)2 * (n_threads > 1 ? n_threads : 1), 0);
    if (n_threads <= 1 || n < ((size_t)1 << 16)) {
        work(0, n, flags.data());
    } else {
        std::vector<std::thread> th;
        const size_t per = (n + n_threads - 1) / n_threads;
        for (int t = 0; t < n_threads; ++t) {
            const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
            if (lo >= hi) break;
            th.emplace_back(work, lo, hi, &flags[(size_t)2 * t]);
        }
        for (auto& x : th) x.join();
    }
    for (size_t t = 0; t < flags.size(); t += 2) {
        if (flags[t]) *bad = 1;
        if (flags[t + 1]) *big = 1;
    }
}
}  // namespace
}  // extern "C++"

int dsq_upload_counts_i32(dsq_ctx* ctx, const void* counts, int count_type, size_t n_elems, int32_t* d_dst,
                          int* h_bad) {
    DSQ_CHECK_ARG(count_type == DSQ_I32 || count_type == DSQ_I64, "count_type");
    if (h_bad) *h_bad = 0;
    if (n_elems == 0) return DSQ_OK;
    for (int k = 0; k < 2; ++k) {
        if (!ctx->stage[k]) DSQ_HIP(hipHostMalloc(&ctx->stage[k], kStageElems * sizeof(int32_t), hipHostMallocDefault));
        if (!ctx->stage_ev[k]) DSQ_HIP(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    }
    static const int n_threads = [] {
        const char* e = getenv("DSQ_UPLOAD_THREADS");
        int t = e ? atoi(e) : (int)std::thread::hardware_concurrency() / 2;
        const int cap = e ? 128 : 16;
        return t < 1 ? 1 : (t > cap ? cap : t);
    }();
    static const bool no_u16 = getenv("DSQ_UPLOAD_NO_U16") != nullptr;  // A/B switch
    int bad = 0;
    size_t off = 0;
    for (int c = 0; off < n_elems; ++c, off += kStageElems) {
        const size_t n = n_elems - off < kStageElems ? n_elems - off : kStageElems;
        const int k = c & 1;
        if (c >= 2) DSQ_HIP(hipEventSynchronize(ctx->stage_ev[k]));  // the DMA out of this buffer has finished
        int32_t* st = (int32_t*)ctx->stage[k];
        int big = no_u16 ? 1 : 0;
        if (!big) {  // optimistic: the chunk as uint16
            if (count_type == DSQ_I64) narrow_chunk_u16((const int64_t*)counts + off, (uint16_t*)st, n, n_threads, &bad, &big);
            else narrow_chunk_u16((const int32_t*)counts + off, (uint16_t*)st, n, n_threads, &bad, &big);
        }
        if (!big) {
            if (!ctx->d_stage16[k]) DSQ_HIP(hipMalloc(&ctx->d_stage16[k], kStageElems * sizeof(uint16_t)));
            // (stream order: the widening kernel of chunk c - 2 has read this device buffer before this copy starts)
            DSQ_HIP(hipMemcpyAsync(ctx->d_stage16[k], st, n * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
            DSQ_HIP(hipEventRecord(ctx->stage_ev[k], ctx->stream));
            DSQ_HIP(dsq::launch_widen_u16(ctx->stream, (const uint16_t*)ctx->d_stage16[k], d_dst + off, n));
        } else {
            if (count_type == DSQ_I64) narrow_chunk((const int64_t*)counts + off, st, n, n_threads, &bad);
            else narrow_chunk((const int32_t*)counts + off, st, n, n_threads, &bad);
            DSQ_HIP(hipMemcpyAsync(d_dst + off, st, n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            DSQ_HIP(hipEventRecord(ctx->stage_ev[k], ctx->stream));
        }
    }
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (h_bad) *h_bad = bad;
    return DSQ_OK;
}

int dsq_host_alloc(dsq_ctx* ctx, size_t bytes, void** out) {
    DSQ_CHECK_ARG(out != nullptr, "null output pointer");
    DSQ_HIP(hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault));
    return DSQ_OK;
}
int dsq_host_free(dsq_ctx* ctx, void* p) {
    if (p) DSQ_HIP(hipHostFree(p));
    return DSQ_OK;
}
// asynchronous device -> pinned-host copy on the context's stream (pair with dsq_sync)
int dsq_d2h_async(dsq_ctx* ctx, void* pinned_dst, const void* d_src, size_t bytes) {
    if (bytes) DSQ_HIP(hipMemcpyAsync(pinned_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return DSQ_OK;
}
int dsq_h2d_async(dsq_ctx* ctx, void* d_dst, const void* pinned_src, size_t bytes) {
    if (bytes) DSQ_HIP(hipMemcpyAsync(d_dst, pinned_src, bytes, hipMemcpyHostToDevice, ctx->stream));
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

// ================================================================== Inference-level API (the drop-in boundary)
// Host arrays in, host arrays out - one entry point per method of pydeseq2.inference.Inference.  The N x G matrices go
// through the content-addressed device cache of dsq_plugin_cache.h: the 7-9 calls of one deseq2() upload the counts, the
// normalised counts and nothing else; mu_hat / mu stay resident between the call that produces them and the calls that
// take them back.  Every device buffer of these calls comes from the cache's free list.
struct PluginDesign {
    struct One {
        std::vector<double> X;  // the host design this entry was built from (N x P row-major): compared byte for byte
        int N = 0, P = 0, ldx = 0, full_rank = 1;
        double *Xt = nullptr, *pinv = nullptr;  // device [P][ldx]
        dsq_mix* mix = nullptr;                 // mixed-design descriptor (NULL: not such a design); built on first request
        int mix_ready = 0;
        uint64_t tick = 0;
    };
    std::vector<One> v;
    uint64_t tick = 0;
};

static void destroy_plugin_designs(dsq_ctx* ctx) {
    if (ctx->designs == nullptr) return;
    for (auto& o : ctx->designs->v) {
        if (o.Xt) (void)hipFree(o.Xt);
        dsq_mix_destroy(o.mix);
    }
    delete ctx->designs;
    ctx->designs = nullptr;
}

namespace {

constexpr int kPluginDesigns = 4;

dsq_pc::Cache& plugin_cache(dsq_ctx* ctx) {
    if (ctx->pc == nullptr) {
        ctx->pc = new dsq_pc::Cache();
        dsq_pc::Cache& c = *ctx->pc;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
        c.budget = total_b / 4;  // (MI355X: 72 GB; the matrices of BASELINE configs[4] at full size are 8.4 GB)
        if (const char* e = getenv("DSQ_PLUGIN_CACHE_MB")) c.budget = (size_t)atoll(e) << 20;
        if (const char* e = getenv("DSQ_PLUGIN_CACHE")) c.enabled = atoi(e) != 0;
        int hw = (int)std::thread::hardware_concurrency();
        c.hash_threads = hw >= 64 ? 32 : (hw >= 4 ? hw / 2 : 1);
        if (const char* e = getenv("DSQ_HASH_THREADS")) c.hash_threads = std::max(1, atoi(e));
        c.verify = getenv("DSQ_PLUGIN_CACHE_VERIFY") != nullptr;
    }
    return *ctx->pc;
}

// a device buffer of the running call, back on the free list when the call returns (or adopted by the cache)
struct PcBuf {
    dsq_ctx* ctx = nullptr;
    void* p = nullptr;
    size_t cap = 0;
    PcBuf() = default;
    PcBuf(const PcBuf&) = delete;
    PcBuf& operator=(const PcBuf&) = delete;
    ~PcBuf() {
        if (p) dsq_pc::give(*ctx->pc, p, cap);
    }
    hipError_t alloc(dsq_ctx* c, size_t bytes) {
        ctx = c;
        return dsq_pc::take(plugin_cache(c), bytes, &p, &cap);
    }
    void release() { p = nullptr; }  // (ownership went to a cache entry)
    template <class T>
    T* as() { return (T*)p; }
};

int pc_begin(dsq_ctx* ctx) {
    DSQ_HIP(hipSetDevice(ctx->device));
    dsq_pc::Cache& c = plugin_cache(ctx);
    if (c.d_acc == nullptr) {
        DSQ_HIP(hipMalloc((void**)&c.d_acc, 4 * sizeof(unsigned long long)));
        DSQ_HIP(hipHostMalloc((void**)&c.h_acc, 4 * sizeof(unsigned long long), hipHostMallocDefault));
    }
    dsq_pc::begin_call(c);
    return DSQ_OK;
}

int pc_upload_small(dsq_ctx* ctx, const void* src, size_t bytes, PcBuf& dst) {
    DSQ_HIP(dst.alloc(ctx, bytes));
    if (bytes) DSQ_HIP(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return DSQ_OK;
}

// verify mode: two gene-major device matrices ([G][ld_words] 32-bit words, n_words used per row) must agree word for word
int pc_verify(dsq_ctx* ctx, const void* fresh, const void* resident, int ld_words, int n_words, int G) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    DSQ_HIP(hipMemsetAsync(c.d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(dsq_pc::k_count_diff, dim3(std::min(G, 2048)), dim3(256), 0, ctx->stream, (const uint32_t*)fresh,
                       (const uint32_t*)resident, ld_words, n_words, G, c.d_acc);
    DSQ_HIP(hipGetLastError());
    DSQ_HIP(hipMemcpyAsync(c.h_acc, c.d_acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    ++c.st.verified;
    if (c.h_acc[3] != 0)
        return fail(ctx, DSQ_ERR_ARG, "plug-in cache: a matrix with the digest of a resident one differs from it "
                                           "(DSQ_PLUGIN_CACHE_VERIFY)");
    return DSQ_OK;
}

// host count matrix -> resident gene-major int32 [G][ldn] (the cache's, not to be freed by the caller)
int pc_counts(dsq_ctx* ctx, const void* counts, int count_type, int layout, int N, int G, int ldn,
              const int32_t** d_y, dsq_pc::Entry** ent = nullptr) {
    DSQ_CHECK_ARG(count_type == DSQ_I32 || count_type == DSQ_I64, "count_type");
    DSQ_CHECK_ARG(layout == DSQ_SAMPLE_MAJOR || layout == DSQ_GENE_MAJOR, "layout");
    dsq_pc::Cache& c = plugin_cache(ctx);
    dsq_pc::Timer t;
    const dsq_pc::Digest dg = count_type == DSQ_I64
                                  ? dsq_pc::digest_host((const int64_t*)counts, layout, N, G, c.hash_threads)
                                  : dsq_pc::digest_host((const int32_t*)counts, layout, N, G, c.hash_threads);
    c.st.hash_ms += t.ms();
    dsq_pc::Entry* hit = dsq_pc::find(c, dsq_pc::kCounts, N, G, dg);
    if (hit != nullptr && !c.verify) {
        ++c.st.hits;
        *d_y = (const int32_t*)hit->d;
        if (ent) *ent = hit;
        return DSQ_OK;
    }
    if (hit == nullptr) ++c.st.misses;
    PcBuf raw, y;
    DSQ_HIP(raw.alloc(ctx, (size_t)N * G * sizeof(int32_t)));
    DSQ_HIP(y.alloc(ctx, (size_t)G * ldn * sizeof(int32_t)));
    int bad = 0, rc;
    if ((rc = dsq_upload_counts_i32(ctx, counts, count_type, (size_t)N * G, raw.as<int32_t>(), &bad))) return rc;
    if (bad) return fail(ctx, DSQ_ERR_RANGE, "counts must be integers in [0, 2^31)");
    c.st.h2d_bytes += (size_t)N * G * (count_type == DSQ_I64 ? 8 : 4);
    DSQ_HIP(dsq::launch_transpose_counts(ctx->stream, raw.p, DSQ_I32, layout, N, G, y.as<int32_t>(), ldn,
                                         (int*)ctx->d_scratch));
    if (hit != nullptr) {  // DSQ_PLUGIN_CACHE_VERIFY: the digest matched - do the bytes?
        int rc2;
        if ((rc2 = pc_verify(ctx, y.p, hit->d, ldn, N, G))) return rc2;
        ++c.st.hits;
        *d_y = (const int32_t*)hit->d;
        if (ent) *ent = hit;
        return DSQ_OK;
    }
    dsq_pc::Entry e;
    e.kind = dsq_pc::kCounts; e.N = N; e.G = G; e.ld = ldn; e.dg = dg; e.d = y.p; e.cap = y.cap;
    y.release();
    dsq_pc::Entry* ne = dsq_pc::insert(c, e);
    *d_y = (const int32_t*)ne->d;
    if (ent) *ent = ne;
    return DSQ_OK;
}

// host fp64 matrix -> resident gene-major [G][ldn]; need_positive: refuse a matrix with an element that is not positive,
// finite and normal (the reference's loss is inf / NaN there: y * log(mu), utils.py:227-234; the kernels' table logarithm
// is undefined) - checked once per resident matrix, on the device
int pc_f64(dsq_ctx* ctx, const double* src, int layout, int N, int G, int ldn, bool need_positive, const double** d_out) {
    DSQ_CHECK_ARG(layout == DSQ_SAMPLE_MAJOR || layout == DSQ_GENE_MAJOR, "layout");
    dsq_pc::Cache& c = plugin_cache(ctx);
    dsq_pc::Timer t;
    const dsq_pc::Digest dg = dsq_pc::digest_host(src, layout, N, G, c.hash_threads);
    c.st.hash_ms += t.ms();
    dsq_pc::Entry* e = dsq_pc::find(c, dsq_pc::kF64, N, G, dg);
    if (e != nullptr && c.verify) {  // DSQ_PLUGIN_CACHE_VERIFY: upload again and compare with the resident copy
        PcBuf raw, m;
        DSQ_HIP(raw.alloc(ctx, (size_t)N * G * sizeof(double)));
        DSQ_HIP(m.alloc(ctx, (size_t)G * ldn * sizeof(double)));
        DSQ_HIP(hipMemcpyAsync(raw.p, src, (size_t)N * G * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        DSQ_HIP(dsq::launch_transpose_f64(ctx->stream, raw.as<double>(), layout, N, G, m.as<double>(), ldn));
        int rc2;
        if ((rc2 = pc_verify(ctx, m.p, e->d, 2 * ldn, 2 * N, G))) return rc2;
    }
    if (e != nullptr) {
        ++c.st.hits;
    } else {
        ++c.st.misses;
        PcBuf raw, m;
        DSQ_HIP(raw.alloc(ctx, (size_t)N * G * sizeof(double)));
        DSQ_HIP(m.alloc(ctx, (size_t)G * ldn * sizeof(double)));
        DSQ_HIP(hipMemcpyAsync(raw.p, src, (size_t)N * G * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        c.st.h2d_bytes += (size_t)N * G * sizeof(double);
        DSQ_HIP(dsq::launch_transpose_f64(ctx->stream, raw.as<double>(), layout, N, G, m.as<double>(), ldn));
        dsq_pc::Entry ne;
        ne.kind = dsq_pc::kF64; ne.N = N; ne.G = G; ne.ld = ldn; ne.dg = dg; ne.d = m.p; ne.cap = m.cap;
        m.release();
        e = dsq_pc::insert(c, ne);
    }
    if (need_positive && e->positive < 0) {
        DSQ_HIP(hipMemsetAsync(c.d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL((dsq_pc::k_digest<double, true>), dim3(std::min(G, 2048)), dim3(256), 0, ctx->stream,
                           (const double*)e->d, ldn, N, G, c.d_acc);
        DSQ_HIP(hipGetLastError());
        DSQ_HIP(hipMemcpyAsync(c.h_acc, c.d_acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));
        e->positive = c.h_acc[2] ? 0 : 1;
    }
    if (need_positive && e->positive == 0)
        return fail(ctx, DSQ_ERR_ARG,
                    "mu must be positive, finite and normal (the negative binomial log-likelihood takes log(mu))");
    *d_out = (const double*)e->d;
    return DSQ_OK;
}

// A gene-major fp64 matrix this call PRODUCED and has copied to the host stays resident under the digest of that host copy
// (computed here, on the device): the caller's next call that hands it back finds it.  Synchronises the stream.
int pc_adopt_f64(dsq_ctx* ctx, PcBuf& buf, int N, int G, int ldn) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    if (!c.enabled) return DSQ_OK;  // (the buffer goes back to the free list with its owner)
    DSQ_HIP(hipMemsetAsync(c.d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL((dsq_pc::k_digest<double, true>), dim3(std::min(G, 2048)), dim3(256), 0, ctx->stream,
                       buf.as<double>(), ldn, N, G, c.d_acc);
    DSQ_HIP(hipGetLastError());
    DSQ_HIP(hipMemcpyAsync(c.h_acc, c.d_acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    dsq_pc::Digest dg;
    dg.a = c.h_acc[0]; dg.b = c.h_acc[1];
    if (dsq_pc::find(c, dsq_pc::kF64, N, G, dg) != nullptr) return DSQ_OK;  // (the same matrix is resident already)
    dsq_pc::Entry e;
    e.kind = dsq_pc::kF64; e.N = N; e.G = G; e.ld = ldn; e.dg = dg; e.d = buf.p; e.cap = buf.cap;
    e.positive = c.h_acc[2] ? 0 : 1;
    buf.release();
    dsq_pc::insert(c, e);
    ++c.st.adopted;
    return DSQ_OK;
}

int pc_download_rows(dsq_ctx* ctx, double* dst, const double* d_src, int ldn, int N, int G) {
    DSQ_HIP(hipMemcpy2DAsync(dst, (size_t)N * sizeof(double), d_src, (size_t)ldn * sizeof(double),
                             (size_t)N * sizeof(double), (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    plugin_cache(ctx).st.d2h_bytes += (size_t)N * G * sizeof(double);
    return DSQ_OK;
}

// the design of a call: factorised once per distinct matrix (QR start values, rank), its device copies and - on request -
// its mixed-design descriptor kept for the next calls (one deseq2() passes the same design 8 times)
int pc_design(dsq_ctx* ctx, const double* design, int N, int P, bool want_mix, PluginDesign::One** out) {
    if (ctx->designs == nullptr) ctx->designs = new PluginDesign();
    PluginDesign& D = *ctx->designs;
    PluginDesign::One* hit = nullptr;
    for (auto& o : D.v)
        if (o.N == N && o.P == P && std::memcmp(o.X.data(), design, (size_t)N * P * sizeof(double)) == 0) hit = &o;
    if (hit == nullptr) {
        if ((int)D.v.size() >= kPluginDesigns) {  // replace the least recently used one
            int lru = 0;
            for (int i = 1; i < (int)D.v.size(); ++i)
                if (D.v[(size_t)i].tick < D.v[(size_t)lru].tick) lru = i;
            DSQ_HIP(hipStreamSynchronize(ctx->stream));
            if (D.v[(size_t)lru].Xt) (void)hipFree(D.v[(size_t)lru].Xt);
            dsq_mix_destroy(D.v[(size_t)lru].mix);
            D.v.erase(D.v.begin() + lru);
        }
        PluginDesign::One o;
        o.N = N; o.P = P; o.ldx = pad16(N);
        o.X.assign(design, design + (size_t)N * P);
        std::vector<double> Xt, pinv;
        design_factor(design, N, P, o.ldx, Xt, pinv, o.full_rank);
        const size_t bytes = Xt.size() * sizeof(double);
        DSQ_HIP(hipMalloc((void**)&o.Xt, 2 * bytes));
        o.pinv = o.Xt + Xt.size();
        DSQ_HIP(hipMemcpyAsync(o.Xt, Xt.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        DSQ_HIP(hipMemcpyAsync(o.pinv, pinv.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        DSQ_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
        D.v.push_back(std::move(o));
        hit = &D.v.back();
    }
    hit->tick = ++D.tick;
    if (want_mix && !hit->mix_ready) {
        int rc;
        if ((rc = dsq_mix_create(ctx, design, N, P, &hit->mix))) return rc;
        hit->mix_ready = 1;
    }
    *out = hit;
    return DSQ_OK;
}

// gene lists of the mixed-design dispersion kernel for a resident count matrix (kept with the cache entry)
int pc_mix_lists(dsq_ctx* ctx, dsq_pc::Entry* e, int N, int G, int ldn) {
    if (e->lists_ready) return DSQ_OK;
    dsq_pc::Cache& c = plugin_cache(ctx);
    PcBuf flags;
    DSQ_HIP(flags.alloc(ctx, (size_t)G * sizeof(int32_t)));
    DSQ_HIP(dsq::launch_count_big(ctx->stream, (const int32_t*)e->d, ldn, N, G, flags.as<int32_t>()));
    std::vector<int32_t> fl((size_t)G), lists((size_t)G);
    DSQ_HIP(hipMemcpyAsync(fl.data(), flags.p, (size_t)G * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    int nr = 0, nw = 0;  // rows first, waves from the end
    for (int g = 0; g < G; ++g) {
        if (fl[(size_t)g] >= 0) lists[(size_t)nr++] = g;
        else lists[(size_t)(G - 1 - nw++)] = g;
    }
    std::reverse(lists.begin() + nr, lists.end());
    void* d = nullptr;
    size_t cap = 0;
    DSQ_HIP(dsq_pc::take(c, (size_t)G * sizeof(int32_t), &d, &cap));
    DSQ_HIP(hipMemcpyAsync(d, lists.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));  // the host list goes out of scope
    e->d_lists = d; e->lists_cap = cap; e->n_rows = nr; e->n_waves = nw; e->lists_ready = 1;
    c.resident += cap;
    return DSQ_OK;
}

}  // namespace

extern "C" {

int dsq_abi_version(void) { return DSQ_ABI_VERSION; }

// the digest the cache identifies a host matrix by (no context, no GPU: CPU tests pin its layout / dtype independence)
int dsq_plugin_digest_host(const void* data, int elem_type, int layout, int N, int G, int n_threads,
                           unsigned long long* out2) {
    if (data == nullptr || out2 == nullptr || N < 0 || G < 0 || (layout != 0 && layout != 1)) return DSQ_ERR_ARG;
    dsq_pc::Digest d;
    if (elem_type == 0) d = dsq_pc::digest_host((const int32_t*)data, layout, N, G, n_threads);
    else if (elem_type == 1) d = dsq_pc::digest_host((const int64_t*)data, layout, N, G, n_threads);
    else if (elem_type == 2) d = dsq_pc::digest_host((const double*)data, layout, N, G, n_threads);
    else return DSQ_ERR_ARG;
    out2[0] = d.a;
    out2[1] = d.b;
    return DSQ_OK;
}

int dsq_plugin_cache_config(dsq_ctx* ctx, int enabled, long long budget_bytes) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    if (enabled >= 0) c.enabled = enabled != 0;
    if (budget_bytes >= 0) c.budget = (size_t)budget_bytes;
    return DSQ_OK;
}

int dsq_plugin_cache_clear(dsq_ctx* ctx) {
    DSQ_HIP(hipSetDevice(ctx->device));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->pc) dsq_pc::clear(*ctx->pc);
    return DSQ_OK;
}

int dsq_plugin_cache_stats(dsq_ctx* ctx, double* out, int n) {
    dsq_pc::Cache& c = plugin_cache(ctx);
    const double v[13] = {(double)c.st.hits, (double)c.st.misses, (double)c.st.adopted, (double)c.st.evictions,
                          (double)c.st.h2d_bytes, (double)c.st.d2h_bytes, c.st.hash_ms, (double)c.resident,
                          (double)c.pooled, (double)c.ents.size(), (double)c.st.mallocs, (double)c.budget,
                          (double)c.st.verified};
    for (int i = 0; i < n && i < 13; ++i) out[i] = v[i];
    return DSQ_OK;
}

// ------------------------------------------------------------------ grid searches + trend GLM as entry points
int dsq_inf_grid_fit_alpha(dsq_ctx* ctx, const void* counts, int count_type, int count_layout, const double* design,
                           const double* mu, int mu_layout, int N, int G, int P, double min_disp, double max_disp,
                           double* log_alpha_out) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    const double* d_mu;
    PluginDesign::One* D;
    PcBuf a, work;
    if ((rc = pc_f64(ctx, mu, mu_layout, N, G, ldn, true, &d_mu))) return rc;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    DSQ_HIP(a.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(work.alloc(ctx, (size_t)G * dsq::kAlphaGridWorkDoubles * sizeof(double)));
    DSQ_HIP(ensure_list(ctx, (size_t)G));
    std::vector<int32_t> all((size_t)G);
    for (int g = 0; g < G; ++g) all[(size_t)g] = g;
    DSQ_HIP(hipMemcpyAsync(ctx->d_list, all.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    // the production fallback path: 100 wavefronts per gene and level (k_alpha_grid_eval)
    DSQ_HIP(dsq::launch_alpha_grid(ctx->stream, d_y, d_mu, ldn, D->Xt, D->ldx, N, P, min_disp, max_disp, a.as<double>(),
                                   ctx->d_list, G, work.as<double>()));
    DSQ_HIP(hipMemcpyAsync(log_alpha_out, a.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    // the production kernel stores alpha = exp(best grid point), which is what fit_alpha_mle needs
    // (utils.py:557); grid_fit_alpha itself returns the grid point (grid_search.py:141-142): back to the log
    // (exp/log round trip: a few 1e-16 absolute)
    for (int g = 0; g < G; ++g) log_alpha_out[g] = std::log(log_alpha_out[g]);
    return DSQ_OK;
}

int dsq_inf_grid_fit_beta(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                          const double* size_factors, const double* design, const double* disp, int N, int G,
                          double min_mu, int grid_length, double min_beta, double max_beta, double* beta_out) {
    if (G <= 0) return DSQ_OK;
    DSQ_CHECK_ARG(grid_length >= 2, "grid_length must be at least 2");
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sf, d, b;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, 2, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size_factors, (size_t)N * sizeof(double), sf))) return rc;
    if ((rc = pc_upload_small(ctx, disp, (size_t)G * sizeof(double), d))) return rc;
    DSQ_HIP(b.alloc(ctx, (size_t)G * 2 * sizeof(double)));
    DSQ_HIP(dsq::launch_grid_beta(ctx->stream, d_y, ldn, sf.as<double>(), D->Xt, D->ldx, N, G, d.as<double>(), min_mu,
                                  min_beta, max_beta, grid_length, b.as<double>()));
    DSQ_HIP(hipMemcpyAsync(beta_out, b.p, (size_t)G * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_dispersion_trend_gamma_glm(dsq_ctx* ctx, const double* covariates, const double* targets, int n,
                                       double* coeffs2, double* predictions, int* converged) {
    DSQ_CHECK_ARG(n >= 1, "no genes");
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    PcBuf cov, tgt, keep;
    if ((rc = pc_upload_small(ctx, covariates, (size_t)n * sizeof(double), cov))) return rc;
    if ((rc = pc_upload_small(ctx, targets, (size_t)n * sizeof(double), tgt))) return rc;
    DSQ_HIP(keep.alloc(ctx, (size_t)n));
    double* d_out = ctx->d_scratch + 1536;
    DSQ_HIP(dsq::launch_trend_glm(ctx->stream, tgt.as<double>(), cov.as<double>(), n, keep.as<uint8_t>(), d_out));
    double out5[5];
    DSQ_HIP(hipMemcpyAsync(out5, d_out, sizeof(out5), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    coeffs2[0] = out5[0]; coeffs2[1] = out5[1];
    if (converged) *converged = (int)out5[2];
    if (predictions)  // covariates @ coeffs (default_inference.py:227)
        for (int i = 0; i < n; ++i) predictions[i] = out5[0] + covariates[i] * out5[1];
    return DSQ_OK;
}

// ------------------------------------------------------------------ the eight Inference methods
int dsq_inf_lin_reg_mu(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                       const double* size_factors, const double* design, int N, int G, int P, double min_mu,
                       double* mu_out) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sf, mu;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size_factors, (size_t)N * sizeof(double), sf))) return rc;
    DSQ_HIP(mu.alloc(ctx, (size_t)G * ldn * sizeof(double)));
    DSQ_HIP(dsq::launch_lin_mu(ctx->stream, d_y, ldn, sf.as<double>(), D->Xt, D->pinv, D->ldx, N, G, P, min_mu,
                               mu.as<double>()));
    if ((rc = pc_download_rows(ctx, mu_out, mu.as<double>(), ldn, N, G))) return rc;
    if ((rc = pc_adopt_f64(ctx, mu, N, G, ldn))) return rc;  // alpha_mle takes this matrix back (dds.py:778-785, 901-911)
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_irls2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                  const double* size_factors, const double* design, const double* disp, int N, int G, int P,
                  double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter,
                  double* beta_out, double* mu_out, double* hat_out, uint8_t* converged, int optimizer) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sf, d, beta, mu, hat, conv;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    // mixed designs: the kernel family of the pipeline (csrc/dsq_mix.h)
    if ((rc = pc_design(ctx, design, N, P, true, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size_factors, (size_t)N * sizeof(double), sf))) return rc;
    if ((rc = pc_upload_small(ctx, disp, (size_t)G * sizeof(double), d))) return rc;
    DSQ_HIP(beta.alloc(ctx, (size_t)G * P * sizeof(double)));
    DSQ_HIP(mu.alloc(ctx, (size_t)G * ldn * sizeof(double)));
    DSQ_HIP(hat.alloc(ctx, (size_t)G * ldn * sizeof(double)));
    DSQ_HIP(conv.alloc(ctx, (size_t)G));
    dsq::IrlsExtras exi{};
    if (D->mix != nullptr) exi.mix = &D->mix->d;
    rc = run_irls(ctx, d_y, ldn, sf.as<double>(), D->Xt, D->pinv, D->ldx, N, G, P, D->full_rank, d.as<double>(), min_mu,
                  beta_tol, min_beta, max_beta, maxiter, beta.as<double>(), mu.as<double>(), hat.as<double>(),
                  conv.as<uint8_t>(), nullptr, D->mix != nullptr ? &exi : nullptr, optimizer);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(beta_out, beta.p, (size_t)G * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(converged, conv.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = pc_download_rows(ctx, mu_out, mu.as<double>(), ldn, N, G))) return rc;
    if ((rc = pc_download_rows(ctx, hat_out, hat.as<double>(), ldn, N, G))) return rc;
    // mu comes back as mu_hat of alpha_mle (dds.py:757-785) or as mu of wald_test (ds.py:338-350)
    if ((rc = pc_adopt_f64(ctx, mu, N, G, ldn))) return rc;
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_irls(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                 const double* size_factors, const double* design, const double* disp, int N, int G, int P,
                 double min_mu, double beta_tol, double min_beta, double max_beta, int maxiter,
                 double* beta_out, double* mu_out, double* hat_out, uint8_t* converged) {
    return dsq_inf_irls2(ctx, counts, count_type, count_layout, size_factors, design, disp, N, G, P, min_mu, beta_tol,
                         min_beta, max_beta, maxiter, beta_out, mu_out, hat_out, converged, 0);
}

int dsq_inf_alpha_mle2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                       const double* design, const double* mu, int mu_layout, const double* alpha_hat, int N,
                       int G, int P, double min_disp, double max_disp, double prior_disp_var, int cr_reg,
                       int prior_reg, double* alpha_out, uint8_t* converged, int optimizer) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    const double* d_mu;
    dsq_pc::Entry* ye = nullptr;
    PluginDesign::One* D;
    PcBuf ah, a, conv;
    if ((rc = pc_f64(ctx, mu, mu_layout, N, G, ldn, true, &d_mu))) return rc;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y, &ye))) return rc;
    // mixed designs (categorical columns + up to three continuous covariates, csrc/dsq_mix.h): the kernel family of
    // the pipeline, here with mu gathered from the caller's matrix; genes with a count beyond its 16-bit staging stay on
    // the general kernel
    const bool want_mix = cr_reg != 0 && optimizer == 0;
    if ((rc = pc_design(ctx, design, N, P, want_mix, &D))) return rc;
    if ((rc = pc_upload_small(ctx, alpha_hat, (size_t)G * sizeof(double), ah))) return rc;
    DSQ_HIP(a.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(conv.alloc(ctx, (size_t)G));
    dsq::AlphaExtras ex{};
    const bool mix = want_mix && D->mix != nullptr;
    if (mix) {
        if ((rc = pc_mix_lists(ctx, ye, N, G, ldn))) return rc;
        ex.mix = &D->mix->d;
        ex.rows = (const int32_t*)ye->d_lists; ex.n_rows = ye->n_rows;
        ex.waves = (const int32_t*)ye->d_lists + ye->n_rows; ex.n_waves = ye->n_waves;
    }
    if ((rc = run_alpha(ctx, d_y, d_mu, ldn, D->Xt, D->ldx, N, G, P, ah.as<double>(), min_disp, max_disp,
                        prior_disp_var, cr_reg, prior_reg, a.as<double>(), conv.as<uint8_t>(), nullptr, nullptr,
                        DSQ_CONST_COMPUTE, mix ? &ex : nullptr, optimizer)))
        return rc;
    DSQ_HIP(hipMemcpyAsync(alpha_out, a.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(converged, conv.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_alpha_mle(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                      const double* design, const double* mu, int mu_layout, const double* alpha_hat, int N,
                      int G, int P, double min_disp, double max_disp, double prior_disp_var, int cr_reg,
                      int prior_reg, double* alpha_out, uint8_t* converged) {
    return dsq_inf_alpha_mle2(ctx, counts, count_type, count_layout, design, mu, mu_layout, alpha_hat, N, G, P, min_disp,
                              max_disp, prior_disp_var, cr_reg, prior_reg, alpha_out, converged, 0);
}

int dsq_inf_lfc_shrink_nbinom_glm2(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                                   const double* design, const double* size, const double* offset, int N, int G,
                                   int P, double prior_no_shrink_scale, double prior_scale, int shrink_index,
                                   double* beta_out, double* inv_hessian_out, uint8_t* converged, int optimizer) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_SHRINK_MAX_P, "P out of range (apeGLM shrinkage: at most 32 design columns)");
    DSQ_CHECK_ARG(shrink_index >= 0 && shrink_index < P, "shrink_index out of range");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const int32_t* d_y;
    PluginDesign::One* D;
    PcBuf sz, off, b, ih, conv;
    if ((rc = pc_counts(ctx, counts, count_type, count_layout, N, G, ldn, &d_y))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, size, (size_t)G * sizeof(double), sz))) return rc;
    if ((rc = pc_upload_small(ctx, offset, (size_t)N * sizeof(double), off))) return rc;
    DSQ_HIP(b.alloc(ctx, (size_t)G * P * sizeof(double)));
    DSQ_HIP(ih.alloc(ctx, (size_t)G * P * P * sizeof(double)));
    DSQ_HIP(conv.alloc(ctx, (size_t)G));
    if ((rc = dsq_dev_lfc_shrink3(ctx, d_y, ldn, off.as<double>(), D->Xt, D->ldx, N, G, P, sz.as<double>(),
                                  prior_no_shrink_scale, prior_scale, shrink_index, b.as<double>(), ih.as<double>(),
                                  conv.as<uint8_t>(), nullptr, optimizer)))
        return rc;
    DSQ_HIP(hipMemcpyAsync(beta_out, b.p, (size_t)G * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(inv_hessian_out, ih.p, (size_t)G * P * P * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    DSQ_HIP(hipMemcpyAsync(converged, conv.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_lfc_shrink_nbinom_glm(dsq_ctx* ctx, const void* counts, int count_type, int count_layout,
                                  const double* design, const double* size, const double* offset, int N, int G,
                                  int P, double prior_no_shrink_scale, double prior_scale, int shrink_index,
                                  double* beta_out, double* inv_hessian_out, uint8_t* converged) {
    return dsq_inf_lfc_shrink_nbinom_glm2(ctx, counts, count_type, count_layout, design, size, offset, N, G, P,
                                          prior_no_shrink_scale, prior_scale, shrink_index, beta_out, inv_hessian_out,
                                          converged, 0);
}

int dsq_inf_wald_test(dsq_ctx* ctx, const double* design, const double* disp, const double* lfc,
                      const double* mu, int mu_layout, const double* ridge, const double* contrast,
                      double lfc_null, int alt, int N, int G, int P, double* pvals, double* stats,
                      double* se) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(alt >= 0 && alt <= 4, "unknown alternative hypothesis");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const double* d_mu;
    PluginDesign::One* D;
    PcBuf d, b, o;
    if ((rc = pc_f64(ctx, mu, mu_layout, N, G, ldn, false, &d_mu))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    if ((rc = pc_upload_small(ctx, disp, (size_t)G * sizeof(double), d))) return rc;
    if ((rc = pc_upload_small(ctx, lfc, (size_t)G * P * sizeof(double), b))) return rc;
    DSQ_HIP(o.alloc(ctx, (size_t)3 * G * sizeof(double)));
    double* dp = o.as<double>();
    rc = dsq_dev_wald(ctx, d_mu, ldn, nullptr, D->Xt, D->ldx, N, G, P, d.as<double>(), b.as<double>(), ridge, contrast,
                      lfc_null, alt, dp, dp + G, dp + 2 * (size_t)G);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(pvals, dp, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(stats, dp + G, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipMemcpyAsync(se, dp + 2 * (size_t)G, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_fit_rough_dispersions(dsq_ctx* ctx, const double* normed, int layout, const double* design,
                                  int N, int G, int P, double* alpha_out) {
    DSQ_CHECK_ARG(P >= 1 && P <= DSQ_MAX_P, "P out of range");
    DSQ_CHECK_ARG(N != P, "The number of samples and the number of design variables are equal, i.e., "
                          "there are no replicates to estimate the dispersion. Please use a design with "
                          "fewer variables.");
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const double* d_v;
    PluginDesign::One* D;
    PcBuf o;
    if ((rc = pc_f64(ctx, normed, layout, N, G, ldn, false, &d_v))) return rc;
    if ((rc = pc_design(ctx, design, N, P, false, &D))) return rc;
    DSQ_HIP(o.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(dsq::launch_rough_from_normed(ctx->stream, d_v, ldn, D->Xt, D->pinv, D->ldx, N, G, P, o.as<double>()));
    DSQ_HIP(hipMemcpyAsync(alpha_out, o.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

// all_zero (nullable): per gene, 1 when every normalised count of the gene is zero - the reference drops those columns
// before it takes the moments (utils.py:878), so the caller drops the same entries of alpha_out
int dsq_inf_fit_moments_dispersions2(dsq_ctx* ctx, const double* normed, int layout, const double* size_factors,
                                     int N, int G, double* alpha_out, uint8_t* all_zero) {
    if (G <= 0) return DSQ_OK;
    const int ldn = pad16(N);
    int rc;
    if ((rc = pc_begin(ctx))) return rc;
    const double* d_v;
    PcBuf o, fl;
    if ((rc = pc_f64(ctx, normed, layout, N, G, ldn, false, &d_v))) return rc;
    double smi = 0.0;
    for (int n = 0; n < N; ++n) smi += 1.0 / size_factors[n];
    smi /= (double)N;
    DSQ_HIP(o.alloc(ctx, (size_t)G * sizeof(double)));
    DSQ_HIP(dsq::launch_moments_from_normed(ctx->stream, d_v, ldn, N, G, smi, o.as<double>()));
    DSQ_HIP(hipMemcpyAsync(alpha_out, o.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (all_zero != nullptr) {
        DSQ_HIP(fl.alloc(ctx, (size_t)G));
        hipLaunchKernelGGL(dsq_pc::k_rows_all_zero, dim3((G + 3) / 4), dim3(256), 0, ctx->stream, d_v, ldn, N, G,
                           fl.as<uint8_t>());
        DSQ_HIP(hipGetLastError());
        DSQ_HIP(hipMemcpyAsync(all_zero, fl.p, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    }
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    return DSQ_OK;
}

int dsq_inf_fit_moments_dispersions(dsq_ctx* ctx, const double* normed, int layout,
                                    const double* size_factors, int N, int G, double* alpha_out) {
    return dsq_inf_fit_moments_dispersions2(ctx, normed, layout, size_factors, N, G, alpha_out, nullptr);
}

}  // extern "C"

// ------------------------------------------------------------------ RCCL (multi-GPU exchanges)
// librccl is resolved lazily with dlopen so that single-GPU use does not depend on it.
struct Uid { char internal[128]; };
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /*ncclUniqueId by value: 128 bytes*/ Uid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
bool load_rccl(std::string& err) {
    if (g_rccl.h) return true;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("dlopen(librccl.so): ") + dlerror(); return false; }
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather) {
        err = "librccl.so lacks the expected nccl* symbols";
        return false;
    }
    g_rccl.h = h;
    return true;
}
#define DSQ_NCCL(call)                                                                             \
    do {                                                                                           \
        int r_ = (call);                                                                           \
        if (r_ != 0)                                                                               \
            return fail(ctx, DSQ_ERR_HIP, std::string(#call) + ": " +                              \
                                              (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error")); \
    } while (0)
}  // namespace

extern "C" {

int dsq_comm_unique_id(dsq_ctx* ctx, char* out128, int len) {
    DSQ_CHECK_ARG(len >= 128, "unique id buffer must hold 128 bytes");
    if (!load_rccl(ctx->err)) return DSQ_ERR_HIP;
    Uid id;
    DSQ_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out128, id.internal, 128);
    return DSQ_OK;
}

int dsq_comm_init(dsq_ctx* ctx, const char* uid128, int rank, int world) {
    DSQ_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
    if (!load_rccl(ctx->err)) return DSQ_ERR_HIP;
    DSQ_HIP(hipSetDevice(ctx->device));
    Uid id;
    memcpy(id.internal, uid128, 128);
    DSQ_NCCL(g_rccl.CommInitRank(&ctx->comm, world, id, rank));
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return DSQ_OK;
}

int dsq_comm_destroy(dsq_ctx* ctx) {
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    return DSQ_OK;
}

// in-place all-reduce (sum) on a device buffer; dtype: 0 = uint32, 1 = float64
int dsq_comm_allreduce_sum(dsq_ctx* ctx, void* d_buf, size_t count, int dtype) {
    DSQ_CHECK_ARG(ctx->comm != nullptr, "dsq_comm_init has not been called");
    const int nccl_type = dtype == 0 ? 3 /*ncclUint32*/ : 8 /*ncclFloat64*/;
    DSQ_NCCL(g_rccl.AllReduce(d_buf, d_buf, count, nccl_type, 0 /*ncclSum*/, ctx->comm, ctx->stream));
    return DSQ_OK;
}

// all-gather of `bytes_per_rank` bytes from every rank into d_recv (world * bytes_per_rank)
int dsq_comm_allgather(dsq_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    DSQ_CHECK_ARG(ctx->comm != nullptr, "dsq_comm_init has not been called");
    DSQ_NCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, 0 /*ncclInt8*/, ctx->comm, ctx->stream));
    return DSQ_OK;
}

// what the RCCL communicator itself reports (ncclCommCount / ncclCommUserRank): bench.py prints it in the result line
int dsq_comm_info(dsq_ctx* ctx, int* nranks, int* rank) {
    DSQ_CHECK_ARG(ctx->comm != nullptr, "dsq_comm_init has not been called");
    DSQ_CHECK_ARG(g_rccl.CommCount != nullptr && g_rccl.CommUserRank != nullptr, "librccl.so lacks ncclCommCount");
    if (nranks) DSQ_NCCL(g_rccl.CommCount(ctx->comm, nranks));
    if (rank) DSQ_NCCL(g_rccl.CommUserRank(ctx->comm, rank));
    return DSQ_OK;
}

unsigned long long dsq_host_sync_count(void) { return g_dsq_host_syncs.load(); }

// gene-sharded trend exchange (pydeseq2_amd/distributed.py): both per-gene vectors of a rank in ONE send buffer, NaN-padded
// to `len` genes each; after the all-gather the [world][2][len] block is split into the two [world * len] vectors the
// trend / prior kernels read
int dsq_dev_pack2(dsq_ctx* ctx, const double* d_a, const double* d_b, int n, int len, double* d_send) {
    DSQ_CHECK_ARG(n >= 0 && n <= len, "n out of range");
    DSQ_HIP(dsq::launch_pack2(ctx->stream, d_a, d_b, n, len, d_send));
    return DSQ_OK;
}
int dsq_dev_unzip2(dsq_ctx* ctx, const double* d_recv, int world, int len, double* d_a_all, double* d_b_all) {
    DSQ_HIP(dsq::launch_unzip2(ctx->stream, d_recv, world, len, d_a_all, d_b_all));
    return DSQ_OK;
}

// ---- size factors, one pass at a time (distributed median of ratios)
int dsq_dev_sf_keys(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G, const double* d_logmeans,
                    const uint8_t* d_gene_mask, void* d_keys) {
    DSQ_HIP(dsq::launch_sf_keys(ctx->stream, d_counts_sm, count_type, N, G, d_logmeans, d_gene_mask,
                                (unsigned long long*)d_keys));
    return DSQ_OK;
}
int dsq_dev_sf_keys_compact(dsq_ctx* ctx, const void* d_counts_sm, int count_type, int N, int G,
                            const double* d_logmeans, const uint8_t* d_gene_mask, int32_t* d_idx_work, void* d_keys,
                            int* h_n_usable) {
    DSQ_HIP(dsq::launch_sf_compact(ctx->stream, d_logmeans, d_gene_mask, G, d_idx_work));
    int gu = 0;
    DSQ_HIP(hipMemcpyAsync(&gu, d_idx_work + G, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DSQ_HIP(dsq::launch_sf_keys_compact(ctx->stream, d_counts_sm, count_type, N, G, d_logmeans, d_idx_work,
                                        (unsigned long long*)d_keys));
    DSQ_HIP(hipStreamSynchronize(ctx->stream));
    *h_n_usable = gu;
    return DSQ_OK;
}
int dsq_dev_sf_count(dsq_ctx* ctx, const void* d_keys, int N, int G, uint32_t* d_counts) {
    DSQ_HIP(dsq::launch_sf_count(ctx->stream, (const unsigned long long*)d_keys, N, G, d_counts));
    return DSQ_OK;
}
int dsq_dev_sf_init(dsq_ctx* ctx, const uint32_t* d_total, int N, void* d_prefix, uint32_t* d_rank) {
    DSQ_HIP(dsq::launch_sf_init(ctx->stream, d_total, N, (unsigned long long*)d_prefix, d_rank));
    return DSQ_OK;
}
int dsq_dev_sf_hist(dsq_ctx* ctx, const void* d_keys, int N, int G, const void* d_prefix, int shift,
                    uint32_t* d_hist) {
    DSQ_HIP(dsq::launch_sf_hist(ctx->stream, (const unsigned long long*)d_keys, N, G,
                                (const unsigned long long*)d_prefix, shift, d_hist));
    return DSQ_OK;
}
int dsq_dev_sf_pick(dsq_ctx* ctx, const uint32_t* d_hist, int N, int shift, void* d_prefix, uint32_t* d_rank) {
    DSQ_HIP(dsq::launch_sf_pick(ctx->stream, d_hist, N, shift, (unsigned long long*)d_prefix, d_rank));
    return DSQ_OK;
}
int dsq_dev_sf_finish(dsq_ctx* ctx, const void* d_prefix, const uint32_t* d_total, int N, double* d_sf) {
    DSQ_HIP(dsq::launch_sf_finish(ctx->stream, (const unsigned long long*)d_prefix, d_total, N, d_sf));
    return DSQ_OK;
}

}  // extern "C"
