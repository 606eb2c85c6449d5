// dsq_k_alpha.hip — dispersion MLE / MAP kernel (gfx950): one gene per wavefront.
// Algorithmic HBM traffic per gene and launch: 4N (counts) + 8N (mu) bytes read once
// (re-reads during the ~5 L-BFGS-B evaluations hit L1/L2: 12 KB per gene at N = 1000),
// 8 + 1 (+4) bytes written.  Compute: ~5 evaluations x N x (lgamma + digamma + 3 log).
#include <atomic>
#include <type_traits>

#include "dsq_alpha.h"
#include "dsq_dispatch.h"
#include "dsq_launch.h"

// minimum waves per SIMD requested from the register allocator: the kernel alternates long
// dependent fp64 chains with L2-latency loads, so 4 waves/SIMD (<= 128 VGPRs) beats the
// compiler's default 2 (measured 4.32 -> 3.69 ms per launch at 60k x 1k, p = 2); wide designs keep
// their p(p+1) accumulators in registers and are left at lower occupancy instead of spilling.
#ifndef DSQ_ALPHA_WAVES_WIDE
#define DSQ_ALPHA_WAVES_WIDE 2
#endif
#ifndef DSQ_ALPHA_WAVES_NARROW
#define DSQ_ALPHA_WAVES_NARROW 3
#endif
constexpr int alpha_min_waves(int p) { return p <= 3 ? DSQ_ALPHA_WAVES_NARROW : (p <= 5 ? 2 : DSQ_ALPHA_WAVES_WIDE); }

namespace dsq {

#if defined(DSQ_PHASE_TIMING)
__device__ unsigned long long g_phase_total[16];
#endif

// STAGE: each wave first copies its gene's counts and mu_hat (12 B per sample) into a wave-private
// LDS segment and runs all ~5 evaluations from there.  Without it every evaluation re-reads the
// row through L2 (16 resident genes x 12 KB per CU overflow L1 and the CU's share of L2), which the
// PMC counters showed as ~10x the algorithmic HBM traffic and ~40 % of wave time in s_waitcnt.
// CELL: the design has few distinct rows (AlphaExtras::cells): per-cell weight sums instead of p(p+1) register
// accumulators per lane (dsq_alpha.h), so the register budget no longer depends on P.
// ex.coef != nullptr (linear-mu designs, dds.py:747-756): mu_hat = max(sf * (x . coef_g), min_mu) is computed while
// staging - the same expression, in the same order, as mom_lin_mu_gene - instead of being read from an N x G matrix
// that k_mom_lin_mu would have had to write (8 N bytes per gene written once and read by both dispersion fits).
template <int P, bool STAGE, bool CELL = false>
__global__ __launch_bounds__(kBlock, CELL ? cell_min_waves(P) : alpha_min_waves(P)) void k_alpha(const int32_t* __restrict__ y,
                                                  const double* __restrict__ mu, int ldn,
                                                  const double* __restrict__ Xt, int ldx, int N, int G,
                                                  const double* __restrict__ alpha_hat,
                                                  double min_disp, double max_disp, double prior_var,
                                                  int cr_reg, int prior_reg, double* __restrict__ alpha,
                                                  uint8_t* __restrict__ conv, int32_t* __restrict__ nfev,
                                                  int32_t* __restrict__ grid_count,
                                                  int32_t* __restrict__ grid_list,
                                                  double* __restrict__ nll_const, int const_mode,
                                                  AlphaExtras ex) {
    // the (wave-uniform) optimiser state lives in LDS, not in every lane's registers
    __shared__ Lbfgsb1d machine[kWavesPerBlock];
    __shared__ typename std::conditional<CELL, CellWork<P>, char>::type cellw[kWavesPerBlock];
    __shared__ CellCtx cellctx[kWavesPerBlock];
    extern __shared__ __attribute__((aligned(16))) double stage[];
    const int w = threadIdx.x >> 6;
    const int gk = blockIdx.x * kWavesPerBlock + w;  // G counts the entries of ex.list when there is one
    if (ex.n_dev != nullptr) {  // phase B of a two-phase launch: launched for a capacity, the count is on the device
        G = min(G, *ex.n_dev);
        if ((int)(blockIdx.x * kWavesPerBlock) >= G) return;
    }
    log_tab_fill();  // the table of flog1p_t (dsq_math.h)
    if (!CELL) __syncthreads();
    if (CELL) {
        // the cells' tables (outer products and rows, <= 64 x (T + P) doubles) are read by every entry-parallel
        // rebuild of X^T W X: once per workgroup into LDS, behind the rows' staging area
        constexpr int T = Tri<P>::N;
        const int npad = (N + 63) & ~63;
        double* sXX = stage + (size_t)kWavesPerBlock * (npad + npad / 2);
        double* sXc = sXX + ex.cells.C * T;
        for (int i = threadIdx.x; i < ex.cells.C * T; i += kBlock) sXX[i] = ex.cells.XX[i];
        for (int i = threadIdx.x; i < ex.cells.C * P; i += kBlock) sXc[i] = ex.cells.Xc[i];
        __syncthreads();
        ex.cells.XX = sXX;
        ex.cells.Xc = sXc;
    }
    if (gk >= G) return;
    const int g = ex.list != nullptr ? ex.list[gk] : gk;
#if defined(DSQ_PHASE_TIMING)
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < kPhases; ++k) g_ph_acc[w][k] = 0;
        g_ph_last[w] = clock64();
        g_ph_cur[w] = 0;
    }
#endif
    if (CELL && (threadIdx.x & 63) == 0) {
        cellctx[w].D = ex.cells;
        cellctx[w].ws = (void*)&cellw[w];
    }
    const int32_t* yg = y + (size_t)g * ldn;
    const double* mg = mu + (size_t)g * ldn;
    int maxc = 0;
    if (STAGE) {
        // rows are padded to a multiple of 64 samples with (y, mu) = (0, 0): such samples contribute
        // exactly zero to every sum of alpha_eval, which then runs without per-sample masks
        const int npad = (N + 63) & ~63;
        double* ms = stage + (size_t)w * (npad + npad / 2);
        int32_t* ys = (int32_t*)(ms + npad);
        // Whole 1-KiB pieces of the two rows go global -> LDS directly (gfx950 LDS-DMA: 16 B per lane,
        // destination = wave-uniform base + lane * 16, no staging registers, all pieces in flight at
        // once); the ragged end and the zero padding go through registers.
        const int lane = threadIdx.x & 63;
        const int full_d = N / 128, full_i = N / 256;  // pieces of 128 doubles / 256 ints
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        if (ex.coef == nullptr && ex.cell_mu == nullptr) {
            for (int c = 0; c < full_d; ++c)
                __builtin_amdgcn_global_load_lds((glb_void*)(mg + c * 128 + lane * 2), (lds_void*)(ms + c * 128), 16, 0, 0);
        }
        for (int c = 0; c < full_i; ++c)
            __builtin_amdgcn_global_load_lds((glb_void*)(yg + c * 256 + lane * 4), (lds_void*)(ys + c * 256), 16, 0, 0);
        if (ex.cell_mu != nullptr) {  // IRLS mu_hat route, per-cell form: sf_n * exp(x_c . beta), unclamped
            const double* cm = ex.cell_mu + (size_t)g * ex.cell_mu_C;
            for (int n = lane; n < npad; n += 64) ms[n] = n < N ? ex.sf[n] * cm[ex.cell_mu_of[n]] : 0.0;
        } else if (ex.coef != nullptr) {
            double b[P];
#pragma unroll
            for (int j = 0; j < P; ++j) b[j] = ex.coef[(size_t)g * P + j];
            for (int n = lane; n < npad; n += 64) {
                double v = 0.0;
                if (n < N) {
                    double yh = 0.0;
#pragma unroll
                    for (int j = 0; j < P; ++j) yh += Xt[j * ldx + n] * b[j];
                    v = dmax(ex.sf[n] * yh, ex.min_mu);
                }
                ms[n] = v;
            }
        } else {
            for (int n = full_d * 128 + lane; n < npad; n += 64) ms[n] = n < N ? mg[n] : 0.0;
        }
        for (int n = full_i * 256 + lane; n < npad; n += 64) ys[n] = n < N ? yg[n] : 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's LDS-DMA has landed
        for (int n = lane; n < npad; n += 64) maxc = ys[n] > maxc ? ys[n] : maxc;
        yg = ys;
        mg = ms;
    } else {
        for (int n = threadIdx.x & 63; n < N; n += 64) maxc = yg[n] > maxc ? yg[n] : maxc;
    }
    maxc = DeviceWave::maxi(maxc);
    const int memo_blocks = __builtin_amdgcn_readfirstlane(min(kMemoBlocks, (maxc >> 6) + 1));
    constexpr int kStateDwords = (int)(sizeof(Lbfgsb1d) / 4);
    static_assert(sizeof(Lbfgsb1d) % 4 == 0, "optimiser state is copied dword by dword");
    if (ex.resume != 0) {  // phase B: the optimiser continues where phase A parked it
        const uint32_t* src = (const uint32_t*)((const Lbfgsb1d*)ex.resume_state + g);
        uint32_t* dst = (uint32_t*)&machine[w];
        for (int i = threadIdx.x & 63; i < kStateDwords; i += 64) dst[i] = src[i];
        DeviceWave::sync();
    }
    const AlphaOut o = fit_alpha_gene<DeviceWave, P, false, STAGE, CELL>(
        yg, mg, Xt, ldx, N, alpha_hat[g], min_disp, max_disp, prior_var, cr_reg != 0, prior_reg != 0, machine[w],
        (const_mode == DSQ_CONST_LOAD || (ex.resume != 0 && nll_const != nullptr)) ? nll_const + g : nullptr,
        (const_mode == DSQ_CONST_STORE && ex.resume == 0) ? nll_const + g : nullptr, memo_blocks,
        CELL ? &cellctx[w] : nullptr, ex.resume != 0 ? 0 : ex.eval_cap, ex.resume != 0);
    if (o.status < 0) {  // phase A ran out of its evaluation budget: park the gene for phase B
        DeviceWave::sync();
        uint32_t* dst = (uint32_t*)((Lbfgsb1d*)ex.resume_state + g);
        const uint32_t* src = (const uint32_t*)&machine[w];
        for (int i = threadIdx.x & 63; i < kStateDwords; i += 64) dst[i] = src[i];
        if ((threadIdx.x & 63) == 0) ex.resume_list[atomicAdd(ex.resume_count, 1)] = g;
    } else if ((threadIdx.x & 63) == 0) {
        alpha[g] = o.alpha;
        conv[g] = (uint8_t)o.converged;
        if (nfev != nullptr) nfev[g] = o.nfev;
        if (!o.converged) grid_list[atomicAdd(grid_count, 1)] = g;
    }
#if defined(DSQ_PHASE_TIMING)
    DSQ_PHASE(0);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < kPhases; ++k) atomicAdd(&g_phase_total[k], (unsigned long long)g_ph_acc[w][k]);
#endif
}

// Grid-search fallback (grid_search.py:54-142) for the (rare) genes whose L-BFGS-B run reported
// success = False.  A gene's 100 grid evaluations are independent, so they are spread over 100
// wavefronts (one grid point each) instead of one wave walking them serially (which cost ~2 ms of
// pure latency per launch).  The wavefront that finishes a gene's level LAST (a counter per gene) picks the
// minimum - stage 0: the refined interval, stage 1: the result - so the whole search is two launches
// (it used to be fill -> eval -> pick -> eval -> pick, each pick a serial scan by one thread: 8 launches and
// ~0.17 ms per dispersion stage with the gather / scatter around them).
constexpr int kGridLen = 100;

// candidate of numpy.argmin's order: the first NaN wins, else the first minimum
struct GridBest {
    double f;
    int i;
    bool nan;
};
__device__ __forceinline__ bool grid_better(const GridBest& a, const GridBest& b) {
    if (a.nan != b.nan) return a.nan;
    if (a.nan) return a.i < b.i;
    if (a.f != b.f) return a.f < b.f;
    return a.i < b.i;
}

// mu_compact: row k of `mu` belongs to the k-th listed gene (rows rebuilt for the list), else row grid_list[k]
template <int P>
__global__ __launch_bounds__(kBlock) void k_alpha_grid_eval(const int32_t* __restrict__ y,
                                                            const double* __restrict__ mu, int ldn,
                                                            const double* __restrict__ Xt, int ldx, int N,
                                                            const int32_t* __restrict__ grid_list, int n_grid,
                                                            int mu_compact, int stage, double lo0, double hi0,
                                                            double* __restrict__ lohi, double* __restrict__ ll,
                                                            int32_t* __restrict__ done, double* __restrict__ alpha,
                                                            const int32_t* __restrict__ n_dev) {
    if (n_dev != nullptr) {  // capacity launch: the number of genes lives on the device
        n_grid = min(n_grid, *n_dev);
        if ((int)(blockIdx.x * kWavesPerBlock) >= n_grid * kGridLen) return;
    }
    log_tab_fill();  // the table of flog1p_t (dsq_math.h)
    __syncthreads();
    const int w = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (w >= n_grid * kGridLen) return;
    const int lane = threadIdx.x & 63;
    const int k = w / kGridLen, i = w % kGridLen;
    const int g = grid_list[k];
    AlphaArgs A;
    A.y = y + (size_t)g * ldn; A.mu = mu + (size_t)(mu_compact ? k : g) * ldn; A.Xt = Xt; A.ldx = ldx; A.N = N;
    A.la_hat = 0.0; A.prior_var = 1.0;
    // the count memo of alpha_eval covers 64 * NB counts and relies on its caller to pick NB from the gene's
    // largest count (as k_alpha does); NB = 1 for a gene with counts >= 64 reads other counts' memo entries
    int maxc = 0;
    A.cst = alpha_const_max<DeviceWave>(A.y, A.mu, N, maxc);
    const int mb = __builtin_amdgcn_readfirstlane(min(kMemoBlocks, (maxc >> 6) + 1));
    const double lo = stage == 0 ? lo0 : lohi[2 * k], hi = stage == 0 ? hi0 : lohi[2 * k + 1];
    const double la = linspace_at(lo, hi, kGridLen, i);
    double f, gu;
    if (mb <= 1) alpha_eval<DeviceWave, P, false, false, 1>(A, la, true, false, f, gu);
    else if (mb == 2) alpha_eval<DeviceWave, P, false, false, 2>(A, la, true, false, f, gu);
    else alpha_eval<DeviceWave, P, false, false, 4>(A, la, true, false, f, gu);
    // publish, and find out whether this wavefront is the last of its gene
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(&ll[(size_t)k * kGridLen + i], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = (atomicAdd(&done[k], 1) == kGridLen - 1) ? 1 : 0;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    __threadfence();
    GridBest best{0.0, 0x7fffffff, false};
    for (int j = lane; j < kGridLen; j += 64) {
        const double fj = __hip_atomic_load(&ll[(size_t)k * kGridLen + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const GridBest c{fj, j, fj != fj};
        if (best.i == 0x7fffffff || grid_better(c, best)) best = c;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        GridBest o;
        o.f = __shfl_xor(best.f, m, 64);
        o.i = __shfl_xor(best.i, m, 64);
        o.nan = __shfl_xor((int)best.nan, m, 64) != 0;
        if (o.i != 0x7fffffff && (best.i == 0x7fffffff || grid_better(o, best))) best = o;
    }
    if (lane == 0) {
        const double c = linspace_at(lo, hi, kGridLen, best.i);
        if (stage == 0) {
            const double delta = linspace_at(lo, hi, kGridLen, 1) - linspace_at(lo, hi, kGridLen, 0);
            lohi[2 * k] = c - delta;
            lohi[2 * k + 1] = c + delta;
            done[k] = 0;  // for the second level
        } else {
            alpha[g] = exp(c);
        }
    }
}

// optimizer="BFGS" (utils.py:546-554): one gene per wavefront, rows read from global memory (a plug-in option of
// fit_alpha_mle that dds.py never selects - kept simple, not tuned)
// (same occupancy request as k_alpha: the out-of-line evaluation is shared with it, and the compiler gives a callee
// the loosest register budget among its callers - without this the un-staged k_alpha of wide designs dropped to one
// wave per SIMD: c5-shaped shard 5.1 -> 7.2 ms per launch)
template <int P>
__global__ __launch_bounds__(kBlock, alpha_min_waves(P)) void k_alpha_bfgs(const int32_t* __restrict__ y, const double* __restrict__ mu,
                                                       int ldn, const double* __restrict__ Xt, int ldx, int N, int G,
                                                       const double* __restrict__ alpha_hat, double min_disp,
                                                       double max_disp, double prior_var, int cr_reg, int prior_reg,
                                                       double* __restrict__ alpha, uint8_t* __restrict__ conv,
                                                       int32_t* __restrict__ nfev, int32_t* __restrict__ grid_count,
                                                       int32_t* __restrict__ grid_list) {
    log_tab_fill();  // the table of flog_t / flog1p_t (dsq_math.h)
    __syncthreads();
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const int32_t* yg = y + (size_t)g * ldn;
    int maxc = 0;
    for (int n = threadIdx.x & 63; n < N; n += 64) maxc = yg[n] > maxc ? yg[n] : maxc;
    maxc = DeviceWave::maxi(maxc);
    const int memo_blocks = __builtin_amdgcn_readfirstlane(min(kMemoBlocks, (maxc >> 6) + 1));
    const AlphaOut o = fit_alpha_gene_bfgs<DeviceWave, P, false>(yg, mu + (size_t)g * ldn, Xt, ldx, N, alpha_hat[g],
                                                                 min_disp, max_disp, prior_var, cr_reg != 0,
                                                                 prior_reg != 0, memo_blocks);
    if ((threadIdx.x & 63) == 0) {
        alpha[g] = o.alpha;
        conv[g] = (uint8_t)o.converged;
        if (nfev != nullptr) nfev[g] = o.nfev;
        if (!o.converged) grid_list[atomicAdd(grid_count, 1)] = g;
    }
}

bool no_two_phase();

// ---- mixed designs: one translation unit per number of continuous covariates (dsq_k_alpha_mix.hip)
#define DSQ_MIX_DECL(Q_)                                                                                               \
    void alpha_mix_grid_q##Q_(int Ns, int P, int n_list, int* blocks, int* nw);                                        \
    hipError_t launch_alpha_mix_q##Q_(hipStream_t, const uint16_t*, const double*, const MixDesign&, const int32_t*, int, \
                                      const int32_t*, int32_t*, const double*, double, double, double, int, double*,   \
                                      uint8_t*, int32_t*, int32_t*, int32_t*, double*, int, int, int, void*, int32_t*,  \
                                      int32_t*);
DSQ_MIX_DECL(1)
DSQ_MIX_DECL(2)
DSQ_MIX_DECL(3)
#undef DSQ_MIX_DECL

static std::atomic<int> g_mix_launches{0};
int alpha_mix_launches() { return g_mix_launches.load(std::memory_order_relaxed); }

bool alpha_mix_enabled() {
    static const bool off = getenv("DSQ_NO_ALPHA_MIX") != nullptr;  // A/B switch: such designs on the general kernels
    return !off;
}

bool alpha_mix_fits(const MixDesign& D) {
    int blocks = 0, nw = 0;
    if (D.Q == 1) alpha_mix_grid_q1(D.Ns, D.P, 1, &blocks, &nw);
    else if (D.Q == 2) alpha_mix_grid_q2(D.Ns, D.P, 1, &blocks, &nw);
    else if (D.Q == 3) alpha_mix_grid_q3(D.Ns, D.P, 1, &blocks, &nw);
    return blocks > 0;
}

hipError_t launch_alpha_mix(hipStream_t st, const uint16_t* ys, const double* mu_s, const MixDesign& D, const int32_t* list,
                            int n_list, const int32_t* n_dev, int32_t* queue, const double* alpha_hat, double min_disp,
                            double max_disp, double prior_var, int prior_reg, double* alpha, uint8_t* conv, int32_t* nfev,
                            int32_t* grid_count, int32_t* grid_list, double* nll_const, int const_mode, int eval_cap,
                            int resume, void* park_state, int32_t* park_count, int32_t* park_list) {
    g_mix_launches.fetch_add(1, std::memory_order_relaxed);
#define DSQ_MIX_CALL(Q_)                                                                                              \
    return launch_alpha_mix_q##Q_(st, ys, mu_s, D, list, n_list, n_dev, queue, alpha_hat, min_disp, max_disp, prior_var, \
                                  prior_reg, alpha, conv, nfev, grid_count, grid_list, nll_const, const_mode, eval_cap, \
                                  resume, park_state, park_count, park_list)
    switch (D.Q) {
        case 1: DSQ_MIX_CALL(1);
        case 2: DSQ_MIX_CALL(2);
        case 3: DSQ_MIX_CALL(3);
        default: return hipErrorInvalidValue;
    }
#undef DSQ_MIX_CALL
}

hipError_t launch_alpha_bfgs(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt, int ldx,
                             int N, int G, int P_, const double* alpha_hat, double min_disp, double max_disp,
                             double prior_var, int cr_reg, int prior_reg, double* alpha, uint8_t* conv,
                             int32_t* nfev, int32_t* grid_count, int32_t* grid_list) {
    if (G <= 0) return hipSuccess;
    if (P_ > DSQ_REG_MAX_P || mu == nullptr) return hipErrorInvalidValue;
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_alpha_bfgs<P>, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, mu, ldn,
                                          Xt, ldx, N, G, alpha_hat, min_disp, max_disp, prior_var, cr_reg, prior_reg,
                                          alpha, conv, nfev, grid_count, grid_list))
    return hipGetLastError();
}

hipError_t launch_alpha(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                        int ldx, int N, int G, int P_, const double* alpha_hat, double min_disp,
                        double max_disp, double prior_var, int cr_reg, int prior_reg, double* alpha,
                        uint8_t* conv, int32_t* nfev, int32_t* grid_count, int32_t* grid_list,
                        double* nll_const, int const_mode, const AlphaExtras* extras, int32_t* queue) {
    if (G <= 0) return hipSuccess;
    if (nll_const == nullptr) const_mode = DSQ_CONST_COMPUTE;
    AlphaExtras ex{};
    if (extras != nullptr) ex = *extras;
    ex.list = nullptr;
    // linear-model mu_hat designs with <= 4 cells: the genes of ex.rows run four to a wavefront
    // (dsq_k_alpha_rows.hip), the rest of them (ex.waves) below
    bool parked = false;
    const int G_rows = ex.n_rows;
    ex.cell_mu_of = ex.cells.cell_of;  // (ex.cells is cleared below for designs of <= 4 cells)
    ex.cell_mu_C = ex.cells.C;
    const bool have_lists = ex.rows != nullptr && queue != nullptr && ex.n_rows + ex.n_waves == G;
    const bool rows_reg = have_lists && ex.cell_mu == nullptr &&
                          alpha_rows_eligible(N, P_, ex.cells.C, ex.coef != nullptr, cr_reg);
    const bool rows_lds = have_lists && !rows_reg && cr_reg != 0 && (ex.coef != nullptr || ex.cell_mu != nullptr) &&
                          alpha_rowsc_tail(N, P_, ex.cells.C) > 0;
    // mixed designs (dsq_k_alpha_mix.hip): the genes of ex.rows, counts and mu_hat streamed from their slot-ordered copies
    const bool rows_mix = have_lists && ex.mix != nullptr && cr_reg != 0 && ex.mix_ys != nullptr && ex.mix_mu != nullptr &&
                          alpha_mix_enabled();
    if (rows_mix) {
        const bool park = ex.resume_state != nullptr && ex.resume_count != nullptr && ex.resume_list != nullptr &&
                          nll_const != nullptr && ex.eval_cap > 0 && !no_two_phase();
        hipError_t e = launch_alpha_mix(st, ex.mix_ys, ex.mix_mu, *ex.mix, ex.rows, ex.n_rows, nullptr, queue, alpha_hat,
                                        min_disp, max_disp, prior_var, prior_reg, alpha, conv, nfev, grid_count, grid_list,
                                        nll_const, const_mode, park ? ex.eval_cap : 0, 0, ex.resume_state, ex.resume_count,
                                        ex.resume_list);
        if (e != hipSuccess) return e;
        if (ex.mid_hook != nullptr) { ex.mid_hook(ex.mid_arg); ex.mid_hook = nullptr; }
        if (ex.conv_late != nullptr) conv = ex.conv_late;  // (everything from here on is "late")
        if (park) {
            // continuation of the parked fits: the same kernel, one gene per wavefront again, launched for a capacity
            // (the count is on the device); queue + 2: its own gene counter (run_alpha zeroes both)
            e = launch_alpha_mix(st, ex.mix_ys, ex.mix_mu, *ex.mix, ex.resume_list, ex.n_rows, ex.resume_count, queue + 2,
                                 alpha_hat, min_disp, max_disp, prior_var, prior_reg, alpha, conv, nfev, grid_count,
                                 grid_list, nll_const, const_mode, 0, 1, ex.resume_state, ex.resume_count, ex.resume_list);
            if (e != hipSuccess) return e;
        }
        if (ex.n_waves <= 0) return hipSuccess;
        ex.list = ex.waves;  // genes with a count beyond the 16-bit staging: the general kernel below
        G = ex.n_waves;
    }
    if (!rows_mix && (rows_reg || rows_lds)) {
        const bool park = ex.resume_state != nullptr && ex.resume_count != nullptr && ex.resume_list != nullptr &&
                          nll_const != nullptr && ex.eval_cap > 0 && !no_two_phase();
        hipError_t e;
        if (rows_reg)
            e = launch_alpha_rows(st, y, ldn, N, ex.rows, ex.n_rows, queue, ex.coef, ex.sf, ex.cells, P_, ex.min_mu,
                                  alpha_hat, min_disp, max_disp, prior_var, cr_reg, prior_reg, alpha, conv, nfev,
                                  grid_count, grid_list, nll_const, const_mode, park ? ex.eval_cap : 0, ex.resume_state,
                                  ex.resume_count, ex.resume_list);
        else
            e = launch_alpha_rows_c(st, y, ldn, N, ex.rows, ex.n_rows, queue, ex.coef, ex.cell_mu, ex.sf, ex.cells, P_,
                                    ex.min_mu, alpha_hat, min_disp, max_disp, prior_var, prior_reg, alpha, conv, nfev,
                                    grid_count, grid_list, nll_const, const_mode, park ? ex.eval_cap : 0,
                                    ex.resume_state, ex.resume_count, ex.resume_list);
        if (e != hipSuccess) return e;
        if (ex.mid_hook != nullptr) { ex.mid_hook(ex.mid_arg); ex.mid_hook = nullptr; }
        if (ex.conv_late != nullptr) conv = ex.conv_late;  // (everything from here on is "late")
        parked = park;
        if (parked && rows_reg && alpha_wg_eligible(N)) {
            // the parked fits continue one per WORKGROUP (k_alpha_wg): ~3 us per evaluation instead of ~15 with the 64
            // lanes of k_alpha - the launch lasts as long as its longest fit (up to 26 more evaluations)
            const hipError_t e2 = launch_alpha_wg(st, y, ldn, N, ex.resume_list, ex.resume_count, ex.n_rows, ex.coef,
                                                  ex.sf, ex.cells, P_, ex.min_mu, alpha_hat, prior_var, prior_reg, alpha,
                                                  conv, nfev, grid_count, grid_list, nll_const, ex.resume_state);
            if (e2 != hipSuccess) return e2;
            parked = false;
        }
        if (ex.n_waves <= 0 && !parked) return hipSuccess;
        ex.list = ex.waves;
        G = ex.n_waves;
    }
    // designs beyond the register path's width - or, on request, any design without cell structure from
    // DSQ_WIDE_MIN_P columns on - run the LDS / matrix-core kernels (dsq_k_wide.hip)
    if (ex.cells.C <= kSmallCells) ex.cells = CellDesign{};  // the dispersion kernels use cells from 5 upwards
    if (P_ > DSQ_REG_MAX_P || (P_ >= wide_min_p() && (ex.cells.C == 0 || wide_with_cells()))) {
        if (mu == nullptr) return hipErrorInvalidValue;  // that path reads a materialised mu_hat
        return launch_wide_alpha(st, y, mu, ldn, Xt, ldx, N, G, P_, alpha_hat, min_disp, max_disp, prior_var, cr_reg,
                                 prior_reg, alpha, conv, nfev, nll_const, const_mode,
                                 ex.cells.C > 0 ? &ex.cells : nullptr);
    }
    const dim3 block(kBlock);
    const int npad = (N + 63) & ~63;
    const size_t smem = (size_t)kWavesPerBlock * (npad + npad / 2) * sizeof(double);
    const bool stage = smem <= 80 * 1024;  // >= 2 workgroups per CU keep their rows in LDS
    if (!stage && (ex.coef != nullptr || ex.cell_mu != nullptr)) return hipErrorInvalidValue;  // mu_hat on the fly needs the staged variant
    const bool cell = stage && ex.cells.C > kSmallCells && P_ >= 3 && cr_reg != 0;
#define DSQ_ALPHA_LAUNCH(KERNEL, SMEM)                                                                            \
    do {                                                                                                          \
        if ((SMEM) > 48 * 1024) {                                                                                 \
            (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SMEM)); \
            (void)hipGetLastError();                                                                              \
        }                                                                                                         \
        hipLaunchKernelGGL(KERNEL, grid, block, (SMEM), st, y, mu, ldn, Xt, ldx, N, G, alpha_hat, min_disp,      \
                           max_disp, prior_var, cr_reg, prior_reg, alpha, conv, nfev, grid_count, grid_list,     \
                           nll_const, const_mode, ex);                                                           \
    } while (0)
    // pass 0: the genes of this kernel (all of them, or the `waves` list beside the row kernel); pass 1: the genes
    // the row kernel parked (AlphaExtras), continued from their optimiser states
    ex.eval_cap = 0;
    for (int pass = 0; pass < (parked ? 2 : 1); ++pass) {
        if (pass == 1) {
            ex.resume = 1;
            ex.list = ex.resume_list;
            ex.n_dev = ex.resume_count;
            G = G_rows;  // capacity: every row-kernel gene could have been parked; blocks beyond the count exit at once
        }
        if (G <= 0) continue;
        const dim3 grid(genes_to_blocks(G));
        if (cell) {
            const size_t smem_c = smem + (size_t)ex.cells.C * (P_ * (P_ + 1) / 2 + P_) * sizeof(double);
            DSQ_DISPATCH_P(P_, {
                if constexpr (P >= 3) DSQ_ALPHA_LAUNCH((k_alpha<P, true, true>), smem_c);
            })
        } else if (stage) {
            DSQ_DISPATCH_P(P_, DSQ_ALPHA_LAUNCH((k_alpha<P, true, false>), smem))
        } else {
            DSQ_DISPATCH_P(P_, DSQ_ALPHA_LAUNCH((k_alpha<P, false, false>), (size_t)0))
        }
        if (ex.mid_hook != nullptr) {
            ex.mid_hook(ex.mid_arg); ex.mid_hook = nullptr;
            if (ex.conv_late != nullptr) conv = ex.conv_late;  // (a second pass of this loop is "late")
        }
    }
#undef DSQ_ALPHA_LAUNCH
    return hipGetLastError();
}

size_t alpha_resume_bytes(int G) { return (size_t)G * sizeof(Lbfgsb1d); }
bool no_two_phase() {
    static const bool off = getenv("DSQ_NO_TWO_PHASE") != nullptr;
    return off;
}

// does the dispersion fit of such a design take the run-time-P (LDS) kernels?  (launch_alpha's routing rule)
bool alpha_is_wide(int P_, int n_cells) {
    if (n_cells <= kSmallCells) n_cells = 0;
    return P_ > DSQ_REG_MAX_P || (P_ >= wide_min_p() && (n_cells == 0 || wide_with_cells()));
}

// must mu_hat be handed over as a matrix?  (the LDS path reads one; rows too long for the staged kernel cannot rebuild it
// from coefficients while staging) - launch_alpha's own routing rule, for callers that would otherwise mirror it
bool alpha_needs_mu(int N, int P_, int n_cells) {
    if (alpha_is_wide(P_, n_cells)) return true;
    const int npad = (N + 63) & ~63;
    return (size_t)kWavesPerBlock * (npad + npad / 2) * sizeof(double) > 80 * 1024;
}

// work: n_grid * (3 + kGridLen) doubles of device scratch
hipError_t launch_alpha_grid(hipStream_t st, const int32_t* y, const double* mu, int ldn, const double* Xt,
                             int ldx, int N, int P_, double min_disp, double max_disp, double* alpha,
                             const int32_t* grid_list, int n_grid, double* work, const int32_t* n_dev,
                             bool mu_compact) {
    if (n_grid <= 0) return hipSuccess;
    if (P_ > DSQ_REG_MAX_P) {
        if (n_dev != nullptr || mu_compact) return hipErrorInvalidValue;  // the LDS path takes its count from the host
        return launch_wide_alpha_grid(st, y, mu, ldn, Xt, ldx, N, P_, min_disp, max_disp, alpha, grid_list, n_grid);
    }
    double* lohi = work;
    double* ll = work + 2 * (size_t)n_grid;
    int32_t* done = (int32_t*)(ll + (size_t)n_grid * kGridLen);
    const dim3 ge(genes_to_blocks(n_grid * kGridLen)), block(kBlock);
    hipError_t e = hipMemsetAsync(done, 0, (size_t)n_grid * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    for (int stage = 0; stage < 2; ++stage) {
        DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_alpha_grid_eval<P>, ge, block, 0, st, y, mu, ldn, Xt, ldx, N,
                                              grid_list, n_grid, mu_compact ? 1 : 0, stage, log(min_disp),
                                              log(max_disp), lohi, ll, done, alpha, n_dev))
    }
    return hipGetLastError();
}

}  // namespace dsq

#if defined(DSQ_PHASE_TIMING)
// developer build only (tools/phase_probe.py): read / reset the per-phase cycle totals of k_alpha
extern "C" int dsq_debug_phase_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dsq::g_phase_total), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(dsq::g_phase_total), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
