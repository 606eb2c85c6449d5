// dsq_k_irls.hip — NB-GLM IRLS kernel (gfx950): one gene per wavefront.
// Algorithmic HBM traffic per gene and launch: 4N bytes of counts read (re-reads of the
// ~4 IRLS sweeps are L1/L2 hits), 8N (mu) + 8N (hat diagonal) written, O(P) scalars.
#include <type_traits>

#include "dsq_dispatch.h"
#include "dsq_irls.h"
#include "dsq_launch.h"

namespace dsq {

#if defined(DSQ_PHASE_TIMING)
__device__ unsigned long long g_phase_total_irls[16];
#endif

// wide designs: ask for 2 waves/SIMD (a few spilled accumulators cost less than running one
// wave per SIMD with nothing to overlap its fp64 dependency chains)
#ifndef DSQ_IRLS_WAVES_P2
#define DSQ_IRLS_WAVES_P2 4
#endif
constexpr int irls_min_waves(int p) { return p <= 2 ? DSQ_IRLS_WAVES_P2 : (p <= 5 ? 1 : 2); }

// fused outputs of one gene (lane 0 writes)
template <int P>
__device__ __forceinline__ void epilogue_begin(LfcEpilogue& E, const IrlsExtras& ex, int g, int ldn) {
    if (ex.flags != nullptr) {
        E.flags = ex.flags; E.robust_disp = ex.robust_disp[g]; E.cutoff = ex.cutoff;
        E.cooks_row = ex.cooks ? ex.cooks + (size_t)g * ldn : nullptr;
    }
    if (ex.ridge != nullptr) { E.ridge = ex.ridge; E.contrast = ex.contrast; E.lfc_null = ex.lfc_null; E.alt = ex.alt; }
}
__device__ __forceinline__ void epilogue_store(const LfcEpilogue& E, const IrlsExtras& ex, int g) {
    if (ex.flags != nullptr) {
        ex.any_all[g] = (uint8_t)E.cooks.any_gt_all;
        ex.any_use[g] = (uint8_t)E.cooks.any_gt_use;
        ex.any_use_nr[g] = (uint8_t)E.cooks.any_gt_use_nr;
        ex.few_above[g] = (uint8_t)E.cooks.few_above;
    }
    if (ex.ridge != nullptr) { ex.pvals[g] = E.wald.p; ex.stats[g] = E.wald.stat; ex.se[g] = E.wald.se; }
}

template <int P, int CELL>
__global__ __launch_bounds__(kBlock, CELL == 1 ? cell_min_waves(P) : irls_min_waves(P)) void k_irls(const int32_t* __restrict__ y, int ldn,
                                                 const double* __restrict__ sf, const double* __restrict__ lsf,
                                                 const double* __restrict__ Xt,
                                                 const double* __restrict__ pinvXt, int ldx, int N,
                                                 int G, int full_rank, const double* __restrict__ disp,
                                                 double min_mu, double beta_tol, double min_beta,
                                                 double max_beta, int maxiter, double* __restrict__ beta,
                                                 double* __restrict__ mu, double* __restrict__ hat,
                                                 uint8_t* __restrict__ conv, int32_t* __restrict__ iters,
                                                 int32_t* __restrict__ fb_count,
                                                 int32_t* __restrict__ fb_list, IrlsExtras ex, int stage) {
    __shared__ typename std::conditional<CELL == 1, CellWork<P>, char>::type cellw[kWavesPerBlock];
    extern __shared__ __attribute__((aligned(16))) double irls_lds[];
    const int w = threadIdx.x >> 6;
    const int g = blockIdx.x * kWavesPerBlock + w;
    if (ex.part != nullptr) {  // a fit in two launches: a workgroup none of whose genes is in this launch's part leaves at once
        bool any = false;
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; ++k) {
            const int gk = blockIdx.x * kWavesPerBlock + k;
            any = any || (gk < G && ex.part[gk] == (uint8_t)ex.part_want);
        }
        if (!any) return;
    }
    log_tab_fill();  // the table of flog_t (dsq_math.h); the barrier below covers it
    double* lds_next = irls_lds;
    if (CELL) {  // the cells' tables once per workgroup into LDS (read by every entry-parallel rebuild)
        constexpr int T = Tri<P>::N;
        double* sXX = lds_next;
        double* sXc = sXX + ex.cells.C * T;
        lds_next = sXc + ex.cells.C * P;
        for (int i = threadIdx.x; i < ex.cells.C * T; i += kBlock) sXX[i] = ex.cells.XX[i];
        for (int i = threadIdx.x; i < ex.cells.C * P; i += kBlock) sXc[i] = ex.cells.Xc[i];
        ex.cells.XX = sXX;
        ex.cells.Xc = sXc;
    }
    // The ~5 sweeps of a fit re-read the gene's counts and the per-sample vectors shared by all genes (size factors,
    // their logs, cell indices); from L2 every such read parks the wave (SQ counters: 35 % - 63 % of the wave cycles
    // of this kernel were s_waitcnt).  stage: they are copied into LDS once - the shared vectors per workgroup, the
    // counts per wave - and the sweeps run from there.
    const int32_t* yrow = y + (size_t)(g < G ? g : G - 1) * ldn;
    if (stage) {
        const int npad = (N + 15) & ~15;
        double* s_sf = lds_next;
        double* s_lsf = s_sf + npad;
        int32_t* s_cell = (int32_t*)(s_lsf + npad);
        int32_t* s_y = s_cell + npad + (size_t)w * npad;
        for (int n = threadIdx.x; n < N; n += kBlock) {
            s_sf[n] = sf[n];
            s_lsf[n] = lsf != nullptr ? lsf[n] : 0.0;
            if (CELL) s_cell[n] = ex.cells.cell_of[n];
        }
        for (int n = threadIdx.x & 63; n < N; n += 64) s_y[n] = yrow[n];
        sf = s_sf;
        if (lsf != nullptr) lsf = s_lsf;
        if (CELL) ex.cells.cell_of = s_cell;
        yrow = s_y;
    }
    __syncthreads();
    if (g >= G) return;
    if (ex.part != nullptr && ex.part[g] != (uint8_t)ex.part_want) return;
#if defined(DSQ_PHASE_TIMING)
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < kPhases; ++k) g_ph_acc[w][k] = 0;
        g_ph_last[w] = clock64();
        g_ph_cur[w] = 0;
    }
#endif
    IrlsArgs A;
    A.y = yrow; A.sf = sf; A.lsf = lsf; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta;
    A.max_beta = max_beta; A.maxiter = maxiter; A.full_rank = full_rank != 0;
    if (CELL) { A.cells = &ex.cells; A.cell_ws = CELL == 1 ? (void*)&cellw[w] : nullptr; }
    LfcEpilogue E;
    epilogue_begin<P>(E, ex, g, ldn);
    double b[P];
    const IrlsOut o = irls_gene<DeviceWave, P, CELL>(A, b, mu ? mu + (size_t)g * ldn : nullptr,
                                                     hat ? hat + (size_t)g * ldn : nullptr, &E);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)o.converged;
        if (iters != nullptr) iters[g] = o.iters;
        if (o.fallback) fb_list[atomicAdd(fb_count, 1)] = g;
        else epilogue_store(E, ex, g);
    }
#if defined(DSQ_PHASE_TIMING)
    DSQ_PHASE(0);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < kPhases; ++k) atomicAdd(&g_phase_total_irls[k], (unsigned long long)g_ph_acc[w][k]);
#endif
}

// The same fit with SIXTEEN lanes per gene (RowWave, dsq_wave.h): four genes per wavefront, sixteen per workgroup.
// For designs with more than kSmallCells cells and p >= 5 the kernel above spends most of its cycles in the
// wave-redundant p x p algebra and the per-cell tables between short sample loops (profiles/r02_phase_c4.txt: sample
// loops 14 % of the cycles); here those instructions serve four genes at once, the sample loops take four times as
// many (light) trips.  Rows diverge when their genes need different numbers of sweeps.  Shared per-sample vectors and
// the cells' tables are staged in LDS; the count rows are read from global memory (L2: 16 rows per workgroup would not
// leave LDS for two workgroups per CU).
constexpr int kRowGenes = kBlock / 16;
template <int P>
__global__ __launch_bounds__(kBlock, 2) void k_irls_row(const int32_t* __restrict__ y, int ldn,
                                                        const double* __restrict__ sf, const double* __restrict__ lsf,
                                                        const double* __restrict__ Xt,
                                                        const double* __restrict__ pinvXt, int ldx, int N, int G,
                                                        int full_rank, const double* __restrict__ disp, double min_mu,
                                                        double beta_tol, double min_beta, double max_beta, int maxiter,
                                                        double* __restrict__ beta, double* __restrict__ mu,
                                                        double* __restrict__ hat, uint8_t* __restrict__ conv,
                                                        int32_t* __restrict__ iters, int32_t* __restrict__ fb_count,
                                                        int32_t* __restrict__ fb_list, IrlsExtras ex) {
    __shared__ CellWork<P> cellw[kRowGenes];
    extern __shared__ __attribute__((aligned(16))) double irls_lds[];
    constexpr int T = Tri<P>::N;
    const int row = threadIdx.x >> 4;
    const int slot = blockIdx.x * kRowGenes + row;
    int g = slot < G ? (ex.order != nullptr ? ex.order[slot] : slot) : G;
    if (ex.part != nullptr && g < G && ex.part[g] != (uint8_t)ex.part_want) g = G;  // not in this launch's part
    if (ex.part != nullptr && __syncthreads_count(g < G) == 0) return;
    log_tab_fill();
    double* sXX = irls_lds;
    double* sXc = sXX + ex.cells.C * T;
    const int npad = (N + 15) & ~15;
    double* s_sf = sXc + ex.cells.C * P;
    double* s_lsf = s_sf + npad;
    double* s_pinvc = s_lsf + npad;
    int32_t* s_cell = (int32_t*)(s_pinvc + ex.cells.C * P);
    int32_t* s_rep = s_cell + npad;  // one sample of each cell
    for (int i = threadIdx.x; i < ex.cells.C * T; i += kBlock) sXX[i] = ex.cells.XX[i];
    for (int i = threadIdx.x; i < ex.cells.C * P; i += kBlock) sXc[i] = ex.cells.Xc[i];
    // the cell's FIRST sample represents it: the columns of pinv(X) of two samples with the same design row agree to
    // rounding only, and "whichever thread wrote last" made the start values - hence the last bits of the fit - differ
    // from launch to launch
    if (full_rank) {
        for (int i = threadIdx.x; i < ex.cells.C; i += kBlock) s_rep[i] = 0x7fffffff;
        __syncthreads();
    }
    for (int n = threadIdx.x; n < N; n += kBlock) {
        s_sf[n] = sf[n];
        s_lsf[n] = lsf != nullptr ? lsf[n] : 0.0;
        const int c = ex.cells.cell_of[n];
        s_cell[n] = c;
        if (full_rank) atomicMin(&s_rep[c], n);
    }
    if (full_rank) {
        __syncthreads();
        for (int i = threadIdx.x; i < ex.cells.C * P; i += kBlock) s_pinvc[i] = pinvXt[(i % P) * ldx + s_rep[i / P]];
    }
    ex.cells.XX = sXX;
    ex.cells.Xc = sXc;
    ex.cells.cell_of = s_cell;
    __syncthreads();
    if (g >= G) return;
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = s_sf; A.lsf = lsf != nullptr ? s_lsf : nullptr; A.Xt = Xt; A.pinvXt = pinvXt;
    A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta;
    A.max_beta = max_beta; A.maxiter = maxiter; A.full_rank = full_rank != 0;
    A.cells = &ex.cells;
    A.cell_ws = (void*)&cellw[row];
    A.pinvc = full_rank ? s_pinvc : nullptr;
#if defined(DSQ_PHASE_TIMING)  // the timeline of the first row of each wavefront
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < kPhases; ++k) g_ph_acc[threadIdx.x >> 6][k] = 0;
        g_ph_last[threadIdx.x >> 6] = clock64();
        g_ph_cur[threadIdx.x >> 6] = 0;
    }
#endif
    LfcEpilogue E;
    epilogue_begin<P>(E, ex, g, ldn);
    double b[P];
    const IrlsOut o = irls_gene<RowWave, P, 1>(A, b, mu ? mu + (size_t)g * ldn : nullptr,
                                               hat ? hat + (size_t)g * ldn : nullptr, &E);
    if ((threadIdx.x & 15) == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)o.converged;
        if (iters != nullptr) iters[g] = o.iters;
        if (o.fallback) fb_list[atomicAdd(fb_count, 1)] = g;
        else epilogue_store(E, ex, g);
    }
#if defined(DSQ_PHASE_TIMING)
    DSQ_PHASE(0);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < kPhases; ++k)
            atomicAdd(&g_phase_total_irls[k], (unsigned long long)g_ph_acc[threadIdx.x >> 6][k]);
#endif
}

// Genes by decreasing predicted number of sweeps (a counting sort over all genes).  The four genes of a k_irls_row
// wavefront iterate until the slowest has converged: in input order that costs 40 % more sweeps than the genes need
// (mean 4.5, mean of the maximum of four 6.3 on the c4 benchmark shape); ordered by the dispersion 13 %, ordered by the
// iteration counts of an earlier fit of the same genes (the mu_hat fit before the LFC fit) 1.4 %.  The order is GLOBAL and
// decreasing on purpose: the long fits start first and the launch ends with short ones (sorting chunks of 1024 genes
// independently evened out the wavefronts just as well and lost a quarter of the launch to its tail).  Two kernels:
// per-workgroup histograms added into a global one, then every workgroup scans the global histogram itself, reserves its
// share of each class with one atomic per class and places its genes (the order inside a class does not matter).
constexpr int kOrderBins = 512;
constexpr int kOrderChunk = 1024;
__device__ __forceinline__ int irls_order_key(const double* __restrict__ disp, const int32_t* __restrict__ hint, int g) {
    int k;
    if (hint != nullptr) {
        k = hint[g];
    } else {  // exponent and three mantissa bits of the dispersion: eight classes per octave
        const unsigned long long b = (unsigned long long)__double_as_longlong(disp[g]);
        const int e = (int)((b >> 52) & 0x7ff) - 1023 + 40;
        k = (b >> 63) ? 0 : (e < 0 ? 0 : (e > 63 ? 63 : e)) * 8 + (int)((b >> 49) & 7);
    }
    k = k < 0 ? 0 : (k >= kOrderBins ? kOrderBins - 1 : k);
    return kOrderBins - 1 - k;  // decreasing
}
// work: [0, kOrderBins) global histogram, [kOrderBins, 2 kOrderBins) per-class cursors - zeroed before the launch
__global__ __launch_bounds__(kOrderChunk) void k_irls_order_hist(const double* __restrict__ disp,
                                                                 const int32_t* __restrict__ hint, int G,
                                                                 int32_t* __restrict__ work) {
    __shared__ int bins[kOrderBins];
    const int g = blockIdx.x * kOrderChunk + threadIdx.x;
    for (int i = threadIdx.x; i < kOrderBins; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    if (g < G) atomicAdd(&bins[irls_order_key(disp, hint, g)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < kOrderBins; i += blockDim.x)
        if (bins[i] != 0) atomicAdd(&work[i], bins[i]);
}
__global__ __launch_bounds__(kOrderChunk) void k_irls_order_place(const double* __restrict__ disp,
                                                                  const int32_t* __restrict__ hint, int G,
                                                                  int32_t* __restrict__ work, int32_t* __restrict__ order) {
    __shared__ int bins[kOrderBins];   // this workgroup's class counts, then its running positions
    __shared__ int scan[kOrderBins];
    const int g = blockIdx.x * kOrderChunk + threadIdx.x;
    const int key = g < G ? irls_order_key(disp, hint, g) : -1;
    for (int i = threadIdx.x; i < kOrderBins; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    if (key >= 0) atomicAdd(&bins[key], 1);
    __syncthreads();
    // exclusive scan of the GLOBAL histogram (Hillis-Steele on the first kOrderBins threads)
    int v = threadIdx.x < kOrderBins ? work[threadIdx.x] : 0;
    const int own = v;
    for (int d = 1; d < kOrderBins; d <<= 1) {
        if (threadIdx.x < kOrderBins) scan[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x < kOrderBins && threadIdx.x >= d) v += scan[threadIdx.x - d];
        __syncthreads();
    }
    if (threadIdx.x < kOrderBins) {
        const int mine = bins[threadIdx.x];
        bins[threadIdx.x] = (v - own) + (mine != 0 ? atomicAdd(&work[kOrderBins + threadIdx.x], mine) : 0);
    }
    __syncthreads();
    if (key >= 0) order[atomicAdd(&bins[key], 1)] = g;
}

hipError_t launch_irls_order(hipStream_t st, const double* disp, const int32_t* hint_iters, int G, int32_t* order,
                             int32_t* work) {
    if (G <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(work, 0, 2 * kOrderBins * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    const dim3 grid((G + kOrderChunk - 1) / kOrderChunk), block(kOrderChunk);
    hipLaunchKernelGGL(k_irls_order_hist, grid, block, 0, st, disp, hint_iters, G, work);
    hipLaunchKernelGGL(k_irls_order_place, grid, block, 0, st, disp, hint_iters, G, work, order);
    return hipGetLastError();
}
int irls_order_work_ints() { return 2 * kOrderBins; }

template <int P>
__global__ __launch_bounds__(kBlock) void k_irls_rescue(const int32_t* __restrict__ y, int ldn,
                                                        const double* __restrict__ sf, const double* __restrict__ lsf,
                                                        const double* __restrict__ Xt,
                                                        const double* __restrict__ pinvXt, int ldx,
                                                        int N, int full_rank,
                                                        const double* __restrict__ disp, double min_mu,
                                                        double beta_tol, double min_beta, double max_beta,
                                                        int maxiter, double* __restrict__ beta,
                                                        double* __restrict__ mu, double* __restrict__ hat,
                                                        uint8_t* __restrict__ conv,
                                                        int32_t* __restrict__ iters,
                                                        const int32_t* __restrict__ fb_list, int n_fb,
                                                        IrlsExtras ex, const int32_t* __restrict__ n_dev) {
    if (n_dev != nullptr) {  // capacity launch: the number of diverged genes lives on the device
        n_fb = min(n_fb, *n_dev);
        if ((int)(blockIdx.x * kWavesPerBlock) >= n_fb) return;
    }
    log_tab_fill();  // the table of flog_t (dsq_math.h)
    __syncthreads();
    int k = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const bool live = k < n_fb;
    if (!live) return;
    const int g = fb_list[k];
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = lsf; A.Xt = Xt; A.pinvXt = pinvXt; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = beta_tol; A.min_beta = min_beta;
    A.max_beta = max_beta; A.maxiter = maxiter; A.full_rank = full_rank != 0;
    __shared__ IrlsRescueWork<P> work[kWavesPerBlock];
    LfcEpilogue E;
    epilogue_begin<P>(E, ex, g, ldn);
    // slot-ordered Cook's layer (mixed designs): this kernel writes sample order - into the scratch row of list entry k
    if (ex.cooks_ld != 0 && E.cooks_row != nullptr) E.cooks_row = ex.cooks_tmp + (size_t)k * ldn;
    double b[P];
    const IrlsOut o = irls_rescue_gene<DeviceWave, P>(A, work[threadIdx.x >> 6], b,
                                                      mu ? mu + (size_t)g * ldn : nullptr,
                                                      hat ? hat + (size_t)g * ldn : nullptr, &E, ex.optimizer);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) beta[(size_t)g * P + j] = b[j];
        conv[g] = (uint8_t)o.converged;
        if (iters != nullptr) iters[g] = o.iters;
        epilogue_store(E, ex, g);
    }
}

// grid_search.grid_fit_beta (grid_search.py:145-221) for every gene of a (small) batch: the same routine the
// rescue kernel falls back to, as a stand-alone entry point (tests pin it against the reference's output)
__global__ __launch_bounds__(kBlock) void k_grid_beta(const int32_t* __restrict__ y, int ldn,
                                                      const double* __restrict__ sf, const double* __restrict__ Xt,
                                                      int ldx, int N, int G, const double* __restrict__ disp,
                                                      double min_mu, double min_beta, double max_beta, int grid_length,
                                                      double* __restrict__ beta) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = nullptr; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = 0.0; A.min_beta = min_beta; A.max_beta = max_beta;
    A.maxiter = 0; A.full_rank = false;
    const double a = 1.0 / A.disp;
    double c = 0.0;  // sum lgamma(y + a) - lgamma(y + 1) - N lgamma(a), as irls_init_exact
    for (int n = DeviceWave::lane(); n < N; n += 64) {
        const double yv = (double)A.y[n];
        c += lgamma_pos(yv + a) - lgamma_pos(yv + 1.0);
    }
    const double cst = DeviceWave::sum(c) - N * lgamma_pos(a);
    double b[2];
    grid_fit_beta2<DeviceWave>(A, a, cst, b, grid_length);
    if ((threadIdx.x & 63) == 0) { beta[2 * g] = b[0]; beta[2 * g + 1] = b[1]; }
}

hipError_t launch_grid_beta(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt, int ldx,
                            int N, int G, const double* disp, double min_mu, double min_beta, double max_beta,
                            int grid_length, double* beta) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_grid_beta, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, ldn, sf, Xt, ldx, N, G, disp,
                       min_mu, min_beta, max_beta, grid_length, beta);
    return hipGetLastError();
}

// N x G layers of a finished fit on demand (dds.layers["_mu_LFC"], ["_hat_diagonals"]): mu = sf exp(X beta)
// (unclamped, utils.py:435-437) and the hat diagonal at the clamped mu (utils.py:427-433) from the stored beta
template <int P>
__global__ __launch_bounds__(kBlock) void k_irls_layers(const int32_t* __restrict__ y, int ldn,
                                                        const double* __restrict__ sf, const double* __restrict__ Xt,
                                                        int ldx, int N, int G, const double* __restrict__ disp,
                                                        const double* __restrict__ beta, double min_mu,
                                                        double* __restrict__ mu, double* __restrict__ hat) {
    log_tab_fill();  // the table of flog_t (dsq_math.h)
    __syncthreads();
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    IrlsArgs A;
    A.y = y + (size_t)g * ldn; A.sf = sf; A.lsf = nullptr; A.Xt = Xt; A.pinvXt = nullptr; A.ldx = ldx; A.N = N;
    A.disp = disp[g]; A.min_mu = min_mu; A.beta_tol = 0.0; A.min_beta = 0.0; A.max_beta = 0.0; A.maxiter = 0;
    A.full_rank = false;
    double b[P], M[Tri<P>::N], r[P], S;
#pragma unroll
    for (int j = 0; j < P; ++j) b[j] = beta[(size_t)g * P + j];
    irls_sweep<DeviceWave, P>(A, b, 1.0 / A.disp, S, M, r);
    irls_finish<DeviceWave, P>(A, b, M, mu ? mu + (size_t)g * ldn : nullptr, hat ? hat + (size_t)g * ldn : nullptr);
}

hipError_t launch_irls_layers(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* Xt, int ldx,
                              int N, int G, int P_, const double* disp, const double* beta, double min_mu, double* mu,
                              double* hat) {
    if (G <= 0) return hipSuccess;
    if (P_ > DSQ_REG_MAX_P)
        return launch_wide_irls_layers(st, y, ldn, sf, Xt, ldx, N, G, P_, disp, beta, min_mu, mu, hat);
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_irls_layers<P>, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, ldn, sf,
                                          Xt, ldx, N, G, disp, beta, min_mu, mu, hat))
    return hipGetLastError();
}

constexpr int kRowMinP = 3;  // instantiated from here on; taken from row_min_p() on
static bool row_wave_enabled() {
    static const bool v = getenv("DSQ_NO_ROW_WAVE") == nullptr;  // A/B switch
    return v;
}
static int row_min_p() {  // narrower designs: the sample loops dominate, one gene per wavefront stays ahead
    static const int v = getenv("DSQ_ROW_MIN_P") ? atoi(getenv("DSQ_ROW_MIN_P")) : 5;
    return v;
}
static size_t row_lds_bytes(int C, int P, int N) {
    const int npad = (N + 15) & ~15;
    return (size_t)C * (P * (P + 1) / 2 + 2 * P) * sizeof(double) + (size_t)npad * 20 + (size_t)kMaxCells * 4;
}

// batches below this many genes leave the sixteen-lane kernel to the one-gene-per-wavefront kernels (DSQ_IRLS_ROW_MIN_G)
static int irls_row_min_genes() {  // (read per launch: tests pin both kernels to the reference KATs in one process)
    const char* e = getenv("DSQ_IRLS_ROW_MIN_G");
    return e ? atoi(e) : 1024;
}

bool irls_takes_parts(int N, int P_, int n_cells, const MixDesign* mix, int full_rank) {
    (void)N;
    if (mix != nullptr && irls_takes_mix(mix, full_rank)) return true;
    return !(P_ > DSQ_REG_MAX_P || (P_ >= wide_min_p() && (n_cells == 0 || wide_with_cells())));
}

bool irls_takes_rows(int N, int P_, int n_cells) {
    if (P_ > DSQ_REG_MAX_P || (P_ >= wide_min_p() && (n_cells == 0 || wide_with_cells()))) return false;
    return n_cells > kSmallCells && P_ >= kRowMinP && P_ >= row_min_p() && row_wave_enabled() &&
           row_lds_bytes(n_cells, P_, N) <= 40 * 1024;
}

// ---- mixed designs: one translation unit per number of continuous covariates (dsq_k_irls_mix.hip)
#define DSQ_MIXI_DECL(Q_)                                                                                             \
    bool irls_mix_fits_q##Q_(int Ns, int P);                                                                          \
    void irls_mix_grid_q##Q_(int Ns, int P, int G, int* blocks, int* nw);                                             \
    hipError_t launch_irls_mix_q##Q_(hipStream_t, const int32_t*, int, const uint16_t*, const uint8_t*, const MixDesign&, \
                                     const double*, int, int32_t*, const double*, double, double, double, int, double*, \
                                     double*, double*, uint8_t*, int32_t*, int32_t*, int32_t*, const IrlsExtras&, void*, \
                                     size_t);
DSQ_MIXI_DECL(1)
DSQ_MIXI_DECL(2)
DSQ_MIXI_DECL(3)
#undef DSQ_MIXI_DECL

bool irls_takes_mix(const MixDesign* mix, int full_rank) {
    static const bool off = getenv("DSQ_NO_IRLS_MIX") != nullptr;  // A/B switch: such designs on the general kernel
    if (off || mix == nullptr || !full_rank || mix->Ginv == nullptr || !alpha_mix_enabled()) return false;
    switch (mix->Q) {
        case 1: return irls_mix_fits_q1(mix->Ns, mix->P);
        case 2: return irls_mix_fits_q2(mix->Ns, mix->P);
        case 3: return irls_mix_fits_q3(mix->Ns, mix->P);
        default: return false;
    }
}

size_t irls_mix_work_bytes(const MixDesign& D, int G, int n_layers) {
    int blocks = 0, nw = 0;
    if (D.Q == 1) irls_mix_grid_q1(D.Ns, D.P, G, &blocks, &nw);
    else if (D.Q == 2) irls_mix_grid_q2(D.Ns, D.P, G, &blocks, &nw);
    else if (D.Q == 3) irls_mix_grid_q3(D.Ns, D.P, G, &blocks, &nw);
    (void)blocks; (void)nw; (void)n_layers;  // (no per-wavefront rows: the layers are written in place)
    // slot-ordered size factors, their logs, Cook's flags + the last sweep's X^T W X of every gene for k_mix_epilogue
    return (((size_t)D.Ns * 17 + 63) & ~(size_t)63) + (size_t)(G > 0 ? G : 0) * (kMixMaxP * (kMixMaxP + 1) / 2) * sizeof(double) + 64;
}

static hipError_t launch_irls_mix(hipStream_t st, const int32_t* y, int ldn, const MixDesign& D, const double* sf, int G,
                                  int32_t* queue, const double* disp, double min_mu, double beta_tol, double max_beta,
                                  int maxiter, double* beta, double* mu, double* hat, uint8_t* conv, int32_t* iters,
                                  int32_t* fb_count, int32_t* fb_list, const IrlsExtras& ex, void* work,
                                  size_t work_bytes) {
    switch (D.Q) {
        case 1: return launch_irls_mix_q1(st, y, ldn, ex.mix_ys, ex.mix_big, D, sf, G, queue, disp, min_mu, beta_tol, max_beta, maxiter, beta, mu,
                                          hat, conv, iters, fb_count, fb_list, ex, work, work_bytes);
        case 2: return launch_irls_mix_q2(st, y, ldn, ex.mix_ys, ex.mix_big, D, sf, G, queue, disp, min_mu, beta_tol, max_beta, maxiter, beta, mu,
                                          hat, conv, iters, fb_count, fb_list, ex, work, work_bytes);
        case 3: return launch_irls_mix_q3(st, y, ldn, ex.mix_ys, ex.mix_big, D, sf, G, queue, disp, min_mu, beta_tol, max_beta, maxiter, beta, mu,
                                          hat, conv, iters, fb_count, fb_list, ex, work, work_bytes);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_irls(hipStream_t st, const int32_t* y, int ldn, const double* sf, const double* lsf,
                       const double* Xt,
                       const double* pinvXt, int ldx, int N, int G, int P_, int full_rank,
                       const double* disp, double min_mu, double beta_tol, double min_beta,
                       double max_beta, int maxiter, double* beta, double* mu, double* hat,
                       uint8_t* conv, int32_t* iters, int32_t* fb_count, int32_t* fb_list,
                       const IrlsExtras* extras) {
    if (G <= 0) return hipSuccess;
    IrlsExtras ex{};
    if (extras != nullptr) ex = *extras;
    // mixed designs (dsq_k_irls_mix.hip): one gene per wavefront in slot order; the same outputs and fallback list
    if (ex.mix != nullptr && ex.mix_work != nullptr && ex.mix_queue != nullptr && ex.mix_ys != nullptr &&
        ex.mix_big != nullptr && irls_takes_mix(ex.mix, full_rank))
        return launch_irls_mix(st, y, ldn, *ex.mix, sf, G, ex.mix_queue, disp, min_mu, beta_tol, max_beta, maxiter, beta,
                               mu, hat, conv, iters, fb_count, fb_list, ex, ex.mix_work, ex.mix_work_bytes);
    if (P_ > DSQ_REG_MAX_P || (P_ >= wide_min_p() && (ex.cells.C == 0 || wide_with_cells()))) {
        if (ex.cells.C > 0 && ex.cells.C <= kSmallCells) ex.cells = CellDesign{};  // (the wide kernels take 5..64 cells)
        return launch_wide_irls(st, y, ldn, sf, lsf, Xt, pinvXt, ldx, N, G, P_, full_rank, disp, min_mu, beta_tol,
                                min_beta, max_beta, maxiter, beta, mu, hat, conv, iters, fb_count, fb_list, &ex);
    }
    const dim3 grid(genes_to_blocks(G)), block(kBlock);
    // LDS staging of the per-sample vectors (sf, log sf: 16 B; cell index: 4 B) and the waves' count rows (4 B each)
    const int npad = (N + 15) & ~15;
    const size_t stage_bytes = (size_t)npad * (16 + 4 + 4 * kWavesPerBlock);
    static const bool allow_stage = getenv("DSQ_IRLS_NO_STAGE") == nullptr;
    if (ex.cells.C > 0 && ex.cells.C <= kSmallCells && P_ <= 4) {
        // <= 4 distinct design rows (two-group designs): per-cell exponentials and weight sums in registers
        DSQ_DISPATCH_P(P_, {
            if constexpr (P <= 4) {
                const size_t tables = (size_t)ex.cells.C * (Tri<P>::N + P) * sizeof(double);
                const int stage = allow_stage && tables + stage_bytes <= 48 * 1024;
                if (ex.cells.C <= 2 && P <= 2)
                    hipLaunchKernelGGL((k_irls<(P <= 2 ? P : 1), 3>), grid, block, tables + (stage ? stage_bytes : 0),
                                       st, y, ldn, sf, lsf, Xt, pinvXt, ldx, N, G, full_rank, disp, min_mu, beta_tol,
                                       min_beta, max_beta, maxiter, beta, mu, hat, conv, iters, fb_count, fb_list, ex,
                                       stage);
                else
                    hipLaunchKernelGGL((k_irls<P, 2>), grid, block, tables + (stage ? stage_bytes : 0), st, y, ldn,
                                       sf, lsf, Xt, pinvXt, ldx, N, G, full_rank, disp, min_mu, beta_tol, min_beta,
                                       max_beta, maxiter, beta, mu, hat, conv, iters, fb_count, fb_list, ex, stage);
            }
        })
    } else if (irls_takes_rows(N, P_, ex.cells.C) && G >= irls_row_min_genes()) {
        // wide categorical designs: sixteen lanes per gene (k_irls_row) - a throughput kernel: a row walks its gene's samples
        // sixteen at a time, so a single fit takes ~4x as long as on a whole wavefront; small batches (the outlier refit's
        // few dozen genes: 175 + 199 us of c4's 7.1 ms step) take the one-gene-per-wavefront cell kernel below
        const dim3 grid_r((G + kRowGenes - 1) / kRowGenes);
        DSQ_DISPATCH_P(P_, {
            if constexpr (P >= kRowMinP)
                hipLaunchKernelGGL((k_irls_row<P>), grid_r, block, row_lds_bytes(ex.cells.C, P, N), st, y, ldn, sf, lsf,
                                   Xt, pinvXt, ldx, N, G, full_rank, disp, min_mu, beta_tol, min_beta, max_beta, maxiter,
                                   beta, mu, hat, conv, iters, fb_count, fb_list, ex);
        })
    } else if (ex.cells.C > kSmallCells && P_ >= 3) {
        DSQ_DISPATCH_P(P_, {
            if constexpr (P >= 3) {
                const size_t tables = (size_t)ex.cells.C * (Tri<P>::N + P) * sizeof(double);
                const int stage = allow_stage && tables + stage_bytes <= 48 * 1024;
                hipLaunchKernelGGL((k_irls<P, 1>), grid, block, tables + (stage ? stage_bytes : 0), st, y, ldn, sf,
                                   lsf, Xt, pinvXt, ldx, N, G, full_rank, disp, min_mu, beta_tol, min_beta, max_beta,
                                   maxiter, beta, mu, hat, conv, iters, fb_count, fb_list, ex, stage);
            }
        })
    } else {
        const int stage = allow_stage && stage_bytes <= 48 * 1024;
        DSQ_DISPATCH_P(P_, hipLaunchKernelGGL((k_irls<P, 0>), grid, block, stage ? stage_bytes : 0, st, y, ldn, sf,
                                              lsf, Xt, pinvXt, ldx, N, G, full_rank, disp, min_mu, beta_tol, min_beta,
                                              max_beta, maxiter, beta, mu, hat, conv, iters, fb_count, fb_list, ex,
                                              stage))
    }
    return hipGetLastError();
}

// Cook's rows of rescued genes: sample-order scratch rows -> the slot-ordered layer of a mixed design (dsq_mix.h)
__global__ __launch_bounds__(256) void k_cooks_rows_to_slots(const double* __restrict__ tmp, int ldn,
                                                             const int32_t* __restrict__ fb_list, int n_fb,
                                                             const int32_t* __restrict__ n_dev,
                                                             const int32_t* __restrict__ slot_of, int N,
                                                             double* __restrict__ cooks, int cooks_ld) {
    if (n_dev != nullptr) n_fb = min(n_fb, *n_dev);
    const int k = blockIdx.x;
    if (k >= n_fb) return;
    const double* src = tmp + (size_t)k * ldn;
    double* dst = cooks + (size_t)fb_list[k] * cooks_ld;
    for (int n = threadIdx.x; n < N; n += 256) dst[slot_of[n]] = src[n];
}

hipError_t launch_irls_rescue(hipStream_t st, const int32_t* y, int ldn, const double* sf,
                              const double* lsf, const double* Xt, const double* pinvXt, int ldx, int N, int P_,
                              int full_rank, const double* disp, double min_mu, double beta_tol,
                              double min_beta, double max_beta, int maxiter, double* beta, double* mu,
                              double* hat, uint8_t* conv, int32_t* iters, const int32_t* fb_list,
                              int n_fb, const IrlsExtras* extras, const int32_t* n_dev) {
    if (n_fb <= 0) return hipSuccess;
    IrlsExtras ex{};
    if (extras != nullptr) ex = *extras;
    const bool wide = P_ > DSQ_REG_MAX_P || (P_ >= wide_min_p() && (ex.cells.C == 0 || wide_with_cells()));
    if (wide && n_dev != nullptr) return hipErrorInvalidValue;  // the LDS path takes its count from the host
    const bool to_slots = ex.cooks_ld != 0 && ex.cooks != nullptr && ex.flags != nullptr;
    if (to_slots && (ex.cooks_tmp == nullptr || ex.mix == nullptr)) return hipErrorInvalidValue;
    // (ex.mix is a host pointer to the descriptor; its members are device pointers)
    const int32_t* slot_of = to_slots ? ex.mix->slot_of : nullptr;
    if (wide) {
        hipError_t e = launch_wide_irls_rescue(st, y, ldn, sf, lsf, Xt, pinvXt, ldx, N, P_, full_rank, disp, min_mu,
                                               beta_tol, min_beta, max_beta, maxiter, beta, mu, hat, conv, iters, fb_list,
                                               n_fb, &ex);
        if (e != hipSuccess) return e;
    } else {
        const dim3 grid(genes_to_blocks(n_fb)), block(kBlock);
        DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_irls_rescue<P>, grid, block, 0, st, y, ldn, sf, lsf, Xt, pinvXt,
                                              ldx, N, full_rank, disp, min_mu, beta_tol, min_beta, max_beta,
                                              maxiter, beta, mu, hat, conv, iters, fb_list, n_fb, ex, n_dev))
    }
    if (to_slots)
        hipLaunchKernelGGL(k_cooks_rows_to_slots, dim3(n_fb), dim3(256), 0, st, ex.cooks_tmp, ldn, fb_list, n_fb, n_dev,
                           slot_of, N, ex.cooks, ex.cooks_ld);
    return hipGetLastError();
}

// does a design of this shape take the run-time-P (LDS) kernels?  (they take their second-pass counts from the host)
bool irls_is_wide(int P_, int n_cells) {
    return P_ > DSQ_REG_MAX_P || (P_ >= wide_min_p() && (n_cells == 0 || wide_with_cells()));
}

}  // namespace dsq

#if defined(DSQ_PHASE_TIMING)
// developer build only (tools/phase_probe.py): per-phase cycle totals of k_irls
extern "C" int dsq_debug_phase_read_irls(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dsq::g_phase_total_irls), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(dsq::g_phase_total_irls), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
