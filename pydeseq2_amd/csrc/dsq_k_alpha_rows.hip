// dsq_k_alpha_rows.hip — dispersion MLE / MAP (utils.py:441-564) with FOUR genes per wavefront, for the designs
// whose mu_hat is "size factor x per-cell value": the linear-model mu_hat of dds.py:747-756 (number of design cells
// == p; here p <= 4, i.e. the two-group comparison of BASELINE configs[1..2] and small factorial designs).
//
// Why.  k_alpha (one gene per 64-lane wavefront) is fp64-issue bound, and at N = 1000 about 1100 of the ~2700
// instructions of an evaluation are wave-uniform: exp / log of alpha, the count memo, the butterfly reductions, the
// p x p algebra and the L-BFGS-B state machine run once per gene in all 64 lanes (at N = 200 that is three quarters of
// the evaluation).  Here a gene owns a 16-lane ROW: the same uniform instructions serve four genes, the sample loop
// takes four times as many (lighter) trips.  Three things make that fit:
//   * no count memo.  sum_n [lgamma(a) - lgamma(y_n + a)] = - sum_{i >= 0} T_i log(a + i) with the TAIL COUNTS
//     T_i = #{n : y_n > i}, which depend on the counts only: they are built once per gene (LDS histogram + suffix sum)
//     and an evaluation takes one logarithm and one reciprocal per distinct i < max count, spread over the row's lanes;
//     the digamma differences are - sum_i T_i / (a + i).  The sample loop no longer touches gamma functions at all.
//     (Counts >= kRowTail are handled per sample with the Stirling series; genes with many of them stay with k_alpha.)
//   * mu_hat is not staged: mu_n = max(sf_n q_c, min_mu) with q_c = x_c . coef, the same expression and operation order
//     as k_mom_lin_mu / k_alpha's staging (all samples of a cell share x_c, so q_c is that very value).  A gene's LDS
//     footprint is its counts as uint16 + tail counts + optimiser state: 3.5 KB at N = 1000 (k_alpha: 12 KB).
//   * rows finish at different evaluation counts, so a row that is done fetches its next gene from a device-side
//     queue at once (persistent wavefronts); staging a gene is done by all 64 lanes for the row that needs it.
// X^T W X = sum_c (sum_{n in c} w_n) x_c x_c^T from per-cell sums kept in registers (<= 4 cells).
#include <cstdio>
#include <type_traits>

#include "dsq_alpha_rows.h"
#include "dsq_dispatch.h"

namespace dsq {

#if defined(DSQ_ROWS_PHASES)
__device__ unsigned long long g_rows_phase[8];
#define ROWS_PH(k)                                                      \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const long long t_ = clock64();                                 \
        ph_acc[ph_cur] += t_ - ph_last; ph_last = t_; ph_cur = (k);     \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#else
#define ROWS_PH(k) ((void)0)
#endif

template <int P>
__global__ __launch_bounds__(kRowBlock, 2) void k_alpha_rows(
    const int32_t* __restrict__ y, int ldn, int N, const int32_t* __restrict__ list, int n_list,
    int32_t* __restrict__ queue, const double* __restrict__ coef, const double* __restrict__ sf,
    const int32_t* __restrict__ cell_of, const double* __restrict__ Xc, const double* __restrict__ XXc, double min_mu,
    const double* __restrict__ alpha_hat, double min_disp, double max_disp, double prior_var, int cr_reg, int prior_reg,
    double* __restrict__ alpha_out, uint8_t* __restrict__ conv, int32_t* __restrict__ nfev,
    int32_t* __restrict__ grid_count, int32_t* __restrict__ grid_list, double* __restrict__ nll_const, int const_mode,
    int eval_cap, Lbfgsb1d* __restrict__ park_state, int32_t* __restrict__ park_count, int32_t* __restrict__ park_list) {
    constexpr int C = P;  // linear-model mu_hat: as many design cells as columns
    constexpr int T = Tri<P>::N;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int npad = (N + 63) & ~63;  // rows are walked four trips at a time: padded to 4 x 16 samples
    double* const sf_s = dyn;                                   // [npad], 0 beyond N
    uint8_t* const cell_s = (uint8_t*)(sf_s + npad);            // [npad]
    char* const slots0 = (char*)(cell_s + npad);
    const size_t slot_bytes = row_slot_bytes(npad);
    unsigned int* const hist0 = (unsigned int*)(slots0 + slot_bytes * kRowSlots * kRowWaves);  // [waves][kRowTail]

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = lane >> 4, rl = lane & 15;
    log_tab_fill();
    for (int n = threadIdx.x; n < npad; n += kRowBlock) {
        sf_s[n] = n < N ? sf[n] : 0.0;
        cell_s[n] = (uint8_t)(n < N ? cell_of[n] : 0);
    }
    __syncthreads();

    auto slot_of = [&](int r) { return (RowGene*)(slots0 + slot_bytes * (size_t)(w * kRowSlots + r)); };
    RowGene* const S = slot_of(row);
    uint16_t* const cnt = (uint16_t*)((char*)S + sizeof(RowGene));
    uint16_t* const tail = cnt + npad;
    unsigned int* const hist = hist0 + (size_t)w * kRowTail;
    if (rl == 0) { S->g = -1; S->n_tail = 0; S->n_big = 0; }
    DeviceWave::sync();

    const double lo = log(min_disp), hi = log(max_disp);
    double xx[C][T];  // x_i x_j of the cells' rows
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int k = 0; k < T; ++k) xx[c][k] = XXc[c * T + k];

#if defined(DSQ_ROWS_PHASES)
    long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = clock64();
    int ph_cur = 0;
#endif
    bool queue_open = true;
    for (;;) {
        // ---------------------------------------------------------------- refill: rows without a gene fetch one
        ROWS_PH(1);
        int mine = S->g;
        if (queue_open) {
            int want = -1;
            if (mine < 0 && rl == 0) {
                const int k = atomicAdd(queue, 1);
                want = k < n_list ? (list != nullptr ? list[k] : k) : -2;  // -2: the queue is empty
            }
            unsigned long long todo = __ballot(want >= 0);
            if (__any(want == -2)) queue_open = false;
            while (todo) {  // wave-uniform: all 64 lanes stage the gene of one row
                const int src = __ffsll((long long)todo) - 1;  // the row's first lane
                todo &= todo - 1;
                const int g = __shfl(want, src, 64);
                const int r = src >> 4;
                RowGene* const Sr = slot_of(r);
                uint16_t* const cr = (uint16_t*)((char*)Sr + sizeof(RowGene));
                uint16_t* const tr = cr + npad;
                for (int i = lane; i < kRowTail; i += 64) hist[i] = 0u;
                double q[C];
                {
                    double b[P];
#pragma unroll
                    for (int j = 0; j < P; ++j) b[j] = coef[(size_t)g * P + j];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        double yh = 0.0;
#pragma unroll
                        for (int j = 0; j < P; ++j) yh += Xc[c * P + j] * b[j];
                        q[c] = yh;
                    }
                }
                DeviceWave::sync();
                const int32_t* yg = y + (size_t)g * ldn;
                const bool want_cst = const_mode != DSQ_CONST_LOAD;
                KSum cs;
                int maxc = 0, nbig = 0;
                // the row of counts comes from HBM: eight loads per lane in flight at a time (one at a time, their
                // latency - with all four rows of the wavefront waiting - was a third of the kernel)
                constexpr int CH = 8;
                for (int base = 0; base < npad; base += 64 * CH) {
                    int v8[CH];
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int n = base + 64 * k + lane;
                        v8[k] = n < N ? yg[n] : 0;
                    }
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int n = base + 64 * k + lane;
                        if (base + 64 * k >= npad) break;  // (wave-uniform)
                        const bool in = n < N;
                        const int v = v8[k];
                        cr[n] = (uint16_t)v;
                        maxc = v > maxc ? v : maxc;
                        const bool isbig = v >= kRowTail;
                        if (in && v > 0 && !isbig) atomicAdd(&hist[v], 1u);
                        // samples beyond the tail table: counted here, evaluated one by one (second sweep of an evaluation)
                        const unsigned long long bm = __ballot(isbig);
                        if (isbig) {
                            if (want_cst) {  // lgamma(y + 1) - lgamma(kRowTail + 1): the part the tail counts do not cover
                                const double z = (double)v + 1.0, zM = (double)kRowTail + 1.0;
                                cs.add(((z - 0.5) * flog(z) - z + stirling_tail(frcp(z))) -
                                       ((zM - 0.5) * flog(zM) - zM + stirling_tail(frcp(zM))));
                            }
                        }
                        nbig += __popcll(bm);
                        if (want_cst && in) {
                            const int cl = cell_s[n];
                            double qq = q[0];
#pragma unroll
                            for (int c = 1; c < C; ++c) qq = (cl == c) ? q[c] : qq;
                            const double mu = dmax(sf_s[n] * qq, min_mu);
                            cs.add(-((double)v * flog_t(mu)));
                        }
                    }
                }
                maxc = DeviceWave::maxi(maxc);
                DeviceWave::sync();
                // tail counts T_i = #{y > i} = (samples beyond the table) + sum_{k > i} hist[k]; lane l owns the
                // entries [8 l, 8 l + 8).  sum_n lgamma(y_n + 1) = sum_{i >= 0} T_i log(i + 1) comes out of the same walk.
                {
                    constexpr int BPL = kRowTail / 64;
                    int h[BPL], tot = 0;
#pragma unroll
                    for (int k = 0; k < BPL; ++k) { h[k] = (int)hist[lane * BPL + k]; tot += h[k]; }
                    const int below = DeviceWave::excl_scan_i(tot);          // entries of lower lanes
                    const int all = DeviceWave::sumi(tot);
                    int above = all - below - tot + nbig;                    // counts > every entry of this lane
#pragma unroll
                    for (int k = BPL - 1; k >= 0; --k) {
                        tr[lane * BPL + k] = (uint16_t)above;                // T_i for i = lane * BPL + k
                        if (want_cst && above > 0) cs.add((double)above * flog_t((double)(lane * BPL + k + 1)));
                        above += h[k];
                    }
                }
                double cst = 0.0;
                if (want_cst) cst = DeviceWave::sum_comp(cs);
                else cst = nll_const[g];
                if (const_mode == DSQ_CONST_STORE && lane == 0) nll_const[g] = cst;
                if (lane == 0) {
                    Sr->g = g;
                    Sr->cst = cst;
                    const int mt = maxc < kRowTail ? maxc : kRowTail;
                    Sr->n_tail = (mt + 2 * kRowLanes - 1) & ~(2 * kRowLanes - 1);
                    Sr->n_big = nbig;
#pragma unroll
                    for (int c = 0; c < C; ++c) Sr->q[c] = q[c];
                    const double lah = log(alpha_hat[g]);
                    Sr->la_hat = lah;
                    Sr->m.start(lah, lo, hi);
                }
                DeviceWave::sync();
            }
            mine = S->g;
        }
        if (!__any(mine >= 0)) break;
        const bool active = mine >= 0;
        ROWS_PH(2);

        // ---------------------------------------------------------------- one evaluation per row
        const double la = active ? S->m.x : 0.0;
        const double alpha = exp(la);
        const double a = frcp(alpha);
        const double lal = flog_t(alpha);
        KSum accf;
        double accg = 0.0;
        {   // gamma-function terms from the tail counts
            const int ntl = active ? S->n_tail : 0;
            for (int i = rl; i < ntl; i += kRowLanes) {
                const double t = a + (double)i;
                const double ti = (double)tail[i];
                accf.add(-(ti * flog_t(t)));
                accg -= ti * frcp(t);
            }
            const int nb = active ? S->n_big : 0;
            if (__any(nb > 0)) {
                // counts >= kRowTail (high-count genes): lgamma / digamma differences beyond the table, sample by sample:
                //   lgamma(a) - lgamma(y + a) = [tail counts: i < kRowTail] + lgamma(kRowTail + a) - lgamma(y + a)
                // a second sweep over the row's counts, taken only by wavefronts that hold such a gene
                double lgM, psiM;
                stirling_big((double)kRowTail + a, lgM, psiM);
                if (nb > 0) {
                    for (int n = rl; n < npad; n += kRowLanes) {
                        const int yi = cnt[n];
                        if (yi >= kRowTail) {
                            double lgz, psiz;
                            stirling_big((double)yi + a, lgz, psiz);
                            accf.add(lgM - lgz);
                            accg += psiM - psiz;
                        }
                    }
                }
            }
        }
        ROWS_PH(3);
        double q[C], wc[C], dwc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { q[c] = S->q[c]; wc[c] = 0.0; dwc[c] = 0.0; }
        // Four samples per lane and iteration, written stage by stage over the four: a sample is one chain of ~45
        // dependent fp64 operations behind two LDS round trips, and two resident waves per SIMD cannot cover that;
        // four independent chains in one basic block let the scheduler interleave them (the rolled loop issued a
        // quarter of the time).  Padding samples (size factor 0) contribute exactly zero.
        constexpr int U = 4;
        KSum af[U];
        double ag[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ag[u] = 0.0;
        // per-cell sums of w and dw: the total and the cells 1 .. C-1 (cell 0 = total - the others); a sample adds into
        // its cell through a 0 / 1 factor (one fused multiply-add instead of a compare and two selects per cell)
        double wt = 0.0, dwt = 0.0;
        for (int n0 = rl; n0 < npad; n0 += kRowLanes * U) {
            int yi[U], cl[U];
            double sfn[U], m[U], r1[U], L1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                yi[u] = cnt[n0 + kRowLanes * u];
                sfn[u] = sf_s[n0 + kRowLanes * u];
                cl[u] = cell_s[n0 + kRowLanes * u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double qq = q[0];
#pragma unroll
                for (int c = 1; c < C; ++c) qq = (cl[u] == c) ? q[c] : qq;
                const double mm = dmax(sfn[u] * qq, min_mu);
                m[u] = (sfn[u] == 0.0) ? 0.0 : mm;  // padding beyond N: every term below is exactly zero
            }
#pragma unroll
            for (int u = 0; u < U; ++u) r1[u] = frcp(1.0 + m[u] * alpha);
            {   // flog1p_t(m alpha, r1) (dsq_math.h) for the four samples, the four table reads issued together
                int kk[U];
                double cc[U], rc[U], tt[U], mant[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double ma = m[u] * alpha;
                    const double wv = 1.0 + ma;
                    cc[u] = (ma - (wv - 1.0)) * r1[u];
                    detail::log_split(wv, kk[u], rc[u], tt[u], mant[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double r = fma(mant[u], rc[u], -1.0);
                    const double pl = detail::log1p_tail(r);
                    const double dk = (double)kk[u];
                    L1[u] = fma(dk, detail::kLn2Hi, tt[u] + (r + (pl + fma(dk, detail::kLn2Lo, cc[u]))));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double yv = (double)yi[u];
                af[u].add(yv * (L1[u] - lal) + a * L1[u]);
                ag[u] += L1[u] + (yv - m[u]) * alpha * r1[u];
                const double wv = m[u] * r1[u];
                const double dwv = -(wv * wv);
                wt += wv;
                dwt += dwv;
#pragma unroll
                for (int c = 1; c < C; ++c) {
                    const double on = (cl[u] == c) ? 1.0 : 0.0;
                    wc[c] = fma(wv, on, wc[c]);
                    dwc[c] = fma(dwv, on, dwc[c]);
                }
            }
        }
        {
            double rest = 0.0, drest = 0.0;
#pragma unroll
            for (int c = 1; c < C; ++c) { rest += wc[c]; drest += dwc[c]; }
            wc[0] = wt - rest;
            dwc[0] = dwt - drest;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { accf.merge(af[u].s, af[u].c); accg += ag[u]; }
        const bool cr = true;  // (the launcher sends fits without the Cox-Reid term to k_alpha)
        ROWS_PH(4);
        const double sumf = RowWave::sum_comp(accf);
        accg = RowWave::sum(accg);
        double f = sumf + S->cst;
        double gr = alpha * (-(a * a * accg));
        if (cr) {
            double M[T], dM[T];
#pragma unroll
            for (int k = 0; k < T; ++k) { M[k] = 0.0; dM[k] = 0.0; }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double ws = RowWave::sum(wc[c]), dws = RowWave::sum(dwc[c]);
#pragma unroll
                for (int k = 0; k < T; ++k) { M[k] += ws * xx[c][k]; dM[k] += dws * xx[c][k]; }
            }
            chol<P>(M);
            f += 0.5 * chol_logdet<P>(M);
            double inv[T];
            chol_inverse<P>(M, inv);
            gr += 0.5 * sym_frob<P>(inv, dM) * alpha;
        }
        if (prior_reg != 0) {
            const double dl = la - S->la_hat;
            f += dl * dl / (2.0 * prior_var);
            gr += dl / prior_var;
        }
        // ---------------------------------------------------------------- optimiser step, results of finished genes
        ROWS_PH(5);
        if (active) {
            S->m.feed(f, gr);
            if (S->m.done) {
                if (rl == 0) {
                    const int g = mine;
                    alpha_out[g] = exp(S->m.x);
                    conv[g] = (uint8_t)(S->m.success ? 1 : 0);
                    if (nfev != nullptr) nfev[g] = S->m.nfev;
                    if (!S->m.success) grid_list[atomicAdd(grid_count, 1)] = g;
                    S->g = -1;
                }
            } else if (eval_cap > 0 && S->m.nfev >= eval_cap) {
                // A fit whose line search ends in rounding noise takes 20-34 evaluations (0.1-0.3 % of the genes; the
                // median is 5, the 99th percentile 7).  A row is slow per evaluation, so one such gene that starts late
                // keeps a wavefront alive for as long as the whole launch otherwise takes (measured: the average
                // wavefront was busy 42 % of the kernel's duration).  Such genes are PARKED - the optimiser's state
                // goes to global memory - and k_alpha continues them after this kernel, all at once, with 64 lanes
                // each (launch_alpha).  The sequence of iterates is unchanged.
                constexpr int kDw = (int)(sizeof(Lbfgsb1d) / 4);
                uint32_t* dst = (uint32_t*)(park_state + mine);
                const uint32_t* src = (const uint32_t*)&S->m;
                for (int i = rl; i < kDw; i += kRowLanes) dst[i] = src[i];
                if (rl == 0) {
                    park_list[atomicAdd(park_count, 1)] = mine;
                    S->g = -1;
                }
            }
        }
        DeviceWave::sync();
    }
#if defined(DSQ_ROWS_PHASES)
    ROWS_PH(0);
    if (lane == 0) {
        unsigned long long life = 0;
        for (int k = 0; k < 6; ++k) { atomicAdd(&g_rows_phase[k], (unsigned long long)ph_acc[k]); life += ph_acc[k]; }
        atomicMax(&g_rows_phase[6], life);            // longest wave lifetime
        atomicAdd(&g_rows_phase[7], 1ull);            // waves
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Continuation of the parked genes: ONE GENE PER WORKGROUP.  The few fits that outlast the evaluation cap of the row
// kernel need 1 .. 26 more evaluations, strictly one after the other: the launch lasts as long as the longest of them,
// and with 64 lanes per gene (k_alpha) that was 0.4 ms per dispersion stage at N = 1000 - a fifth of the stage for 1 % of
// the genes.  Here every thread of a workgroup owns at most kWgSpt samples in registers (N = 1000: 1024 threads, one
// sample each), so an evaluation is ~100 instructions of per-sample work, a two-level reduction through LDS and the
// optimiser step: about 3 us instead of 15.  Same objective as the row kernel (tail counts, per-cell mu_hat); every
// wavefront of the workgroup keeps its own copy of the optimiser state and steps it with the same totals.
constexpr int kWgSpt = 4;  // samples per thread at most

template <int P, int TPB>
__global__ __launch_bounds__(TPB) void k_alpha_wg(
    const int32_t* __restrict__ y, int ldn, int N, const int32_t* __restrict__ list, const int32_t* __restrict__ n_dev,
    const double* __restrict__ coef, const double* __restrict__ sf, const int32_t* __restrict__ cell_of,
    const double* __restrict__ Xc, const double* __restrict__ XXc, double min_mu, const double* __restrict__ alpha_hat,
    double prior_var, int prior_reg, double* __restrict__ alpha_out, uint8_t* __restrict__ conv,
    int32_t* __restrict__ nfev, int32_t* __restrict__ grid_count, int32_t* __restrict__ grid_list,
    const double* __restrict__ nll_const, const Lbfgsb1d* __restrict__ park_state) {
    constexpr int C = P;
    constexpr int T = Tri<P>::N;
    constexpr int NV = 3 + 2 * C;  // reduced values: f (sum, compensation), g, per-cell w and dw
    const int n_parked = *n_dev;
    if ((int)blockIdx.x >= n_parked) return;
    __shared__ unsigned int hist[kRowTail];
    __shared__ int wave_tot[kRowTail / 64];
    __shared__ double part[16][NV];
    __shared__ Lbfgsb1d mach[16];
    __shared__ int s_nbig;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
    log_tab_fill();
    double xx[C][T];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int k = 0; k < T; ++k) xx[c][k] = XXc[c * T + k];
    // (launched for a fixed number of workgroups: a launch with one - mostly empty - workgroup per gene that COULD have
    // been parked spent 0.3 ms dispatching them)
    for (int item = blockIdx.x; item < n_parked; item += gridDim.x) {
    const int g = list[item];
    __syncthreads();  // the previous gene's tables are no longer in use
    for (int i = tid; i < kRowTail; i += nthreads) hist[i] = 0u;
    if (tid == 0) s_nbig = 0;
    {   // this wavefront's copy of the parked optimiser state
        constexpr int kDw = (int)(sizeof(Lbfgsb1d) / 4);
        const uint32_t* src = (const uint32_t*)(park_state + g);
        uint32_t* dst = (uint32_t*)&mach[w];
        for (int i = lane; i < kDw; i += 64) dst[i] = src[i];
    }
    double q[C];
    {
        double b[P];
#pragma unroll
        for (int j = 0; j < P; ++j) b[j] = coef[(size_t)g * P + j];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double yh = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) yh += Xc[c * P + j] * b[j];
            q[c] = yh;
        }
    }
    __syncthreads();
    // this thread's samples
    const int32_t* yg = y + (size_t)g * ldn;
    int yi[kWgSpt], cl[kWgSpt];
    double mu[kWgSpt];
    int nbig = 0;
#pragma unroll
    for (int k = 0; k < kWgSpt; ++k) {
        const int n = tid + k * nthreads;
        const bool in = n < N;
        yi[k] = in ? yg[n] : 0;
        cl[k] = in ? cell_of[n] : 0;
        double qq = q[0];
#pragma unroll
        for (int c = 1; c < C; ++c) qq = (cl[k] == c) ? q[c] : qq;
        mu[k] = in ? dmax(sf[n] * qq, min_mu) : 0.0;
        if (yi[k] >= kRowTail) nbig += 1;
        else if (yi[k] > 0) atomicAdd(&hist[yi[k]], 1u);
    }
    nbig = DeviceWave::sumi(nbig);
    if (lane == 0 && nbig > 0) atomicAdd(&s_nbig, nbig);
    __syncthreads();
    // tail counts of this thread's entries i = tid + k * TPB: T_i = #{y > i}
    constexpr int TPT = (kRowTail + TPB - 1) / TPB;  // entries per thread
    double my_tail[TPT];
    {
        int suf[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int i = tid + k * TPB;
            suf[k] = 0;
            if (i < kRowTail) {  // (whole wavefronts: TPB and kRowTail are multiples of 64)
                const int h = (int)hist[i];
                int v = h;  // inclusive suffix sum inside the chunk of 64 entries
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_down(v, d, 64);
                    if (lane + d < 64) v += o;
                }
                suf[k] = v - h;  // entries above this one in the same chunk
                if (lane == 0) wave_tot[i >> 6] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int i = tid + k * TPB;
            my_tail[k] = 0.0;
            if (i < kRowTail) {
                int above = s_nbig + suf[k];
                for (int cc = (i >> 6) + 1; cc < kRowTail / 64; ++cc) above += wave_tot[cc];
                my_tail[k] = (double)above;
            }
        }
    }
    const double cst = nll_const[g];
    const double la_hat = log(alpha_hat[g]);
    // the optimiser state in registers for the loop (stepping it through an LDS reference serialises on the LDS
    // latency of every member access: ~10 us per evaluation instead of ~3)
    Lbfgsb1d m = mach[w];
    DeviceWave::sync();
    while (!m.done) {
        const double la = m.x;
        const double alpha = exp(la);
        const double a = frcp(alpha);
        const double lal = flog_t(alpha);
        KSum accf;
        double accg = 0.0, wc[C], dwc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { wc[c] = 0.0; dwc[c] = 0.0; }
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            if (my_tail[k] > 0.0) {
                const double t = a + (double)(tid + k * TPB);
                accf.add(-(my_tail[k] * flog_t(t)));
                accg -= my_tail[k] * frcp(t);
            }
        }
        double lgM = 0.0, psiM = 0.0;
        if (s_nbig > 0) stirling_big((double)kRowTail + a, lgM, psiM);
#pragma unroll
        for (int k = 0; k < kWgSpt; ++k) {
            if (tid + k * nthreads < N) {
                const double mm = mu[k], yv = (double)yi[k];
                const double ma = mm * alpha;
                const double r1 = frcp(1.0 + ma);
                const double L1 = flog1p_t(ma, r1);
                accf.add(yv * (L1 - lal) + a * L1);
                accg += L1 + (yv - mm) * alpha * r1;
                const double wv = mm * r1, dwv = -(wv * wv);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    wc[c] += (cl[k] == c) ? wv : 0.0;
                    dwc[c] += (cl[k] == c) ? dwv : 0.0;
                }
                if (yi[k] >= kRowTail) {
                    double lgz, psiz;
                    stirling_big(yv + a, lgz, psiz);
                    accf.add(lgM - lgz);
                    accg += psiM - psiz;
                }
            }
        }
        // level 1: inside the wavefront; level 2: every wavefront adds the partials of all of them in the same order
        {
            KSum k1 = accf;
            const double f1 = DeviceWave::sum_comp(k1);
            const double g1 = DeviceWave::sum(accg);
            double v[NV];
            v[0] = f1; v[1] = 0.0; v[2] = g1;
#pragma unroll
            for (int c = 0; c < C; ++c) { v[3 + c] = DeviceWave::sum(wc[c]); v[3 + C + c] = DeviceWave::sum(dwc[c]); }
            if (lane == 0)
#pragma unroll
                for (int i = 0; i < NV; ++i) part[w][i] = v[i];
        }
        __syncthreads();
        KSum fs;
        double gs = 0.0, ws[C], dws[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { ws[c] = 0.0; dws[c] = 0.0; }
        for (int ww = 0; ww < nwaves; ++ww) {
            fs.add(part[ww][0]);
            gs += part[ww][2];
#pragma unroll
            for (int c = 0; c < C; ++c) { ws[c] += part[ww][3 + c]; dws[c] += part[ww][3 + C + c]; }
        }
        __syncthreads();
        double f = fs.value() + cst;
        double gr = alpha * (-(a * a * gs));
        {
            double M[T], dM[T];
#pragma unroll
            for (int k = 0; k < T; ++k) { M[k] = 0.0; dM[k] = 0.0; }
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int k = 0; k < T; ++k) { M[k] += ws[c] * xx[c][k]; dM[k] += dws[c] * xx[c][k]; }
            chol<P>(M);
            f += 0.5 * chol_logdet<P>(M);
            double inv[T];
            chol_inverse<P>(M, inv);
            gr += 0.5 * sym_frob<P>(inv, dM) * alpha;
        }
        if (prior_reg != 0) {
            const double dl = la - la_hat;
            f += dl * dl / (2.0 * prior_var);
            gr += dl / prior_var;
        }
        m.feed(f, gr);
        DeviceWave::sync();
    }
    if (tid == 0) {
        alpha_out[g] = exp(m.x);
        conv[g] = (uint8_t)(m.success ? 1 : 0);
        if (nfev != nullptr) nfev[g] = m.nfev;
        if (!m.success) grid_list[atomicAdd(grid_count, 1)] = g;
    }
    }  // next parked gene of this workgroup
}

static bool wg_narrow_enabled() {
    static const bool v = getenv("DSQ_WG_512") == nullptr;  // A/B switch: 512 threads also for rows of <= 1024 samples
    return v;
}
bool alpha_wg_eligible(int N) { return N <= 1024 * kWgSpt && getenv("DSQ_NO_ALPHA_WG") == nullptr; }

// the parked genes of the row kernel (count on the device), one workgroup each; capacity = n_cap workgroups
hipError_t launch_alpha_wg(hipStream_t st, const int32_t* y, int ldn, int N, const int32_t* list, const int32_t* n_dev,
                           int n_cap, const double* coef, const double* sf, const CellDesign& cells, int P_,
                           double min_mu, const double* alpha_hat, double prior_var, int prior_reg, double* alpha,
                           uint8_t* conv, int32_t* nfev, int32_t* grid_count, int32_t* grid_list,
                           const double* nll_const, const void* park_state) {
    if (n_cap <= 0) return hipSuccess;
    // 256 threads with four samples each (rows of <= 1024 samples): one wavefront per SIMD - most of an evaluation is the
    // wave-uniform part (reductions, algebra, optimiser step), which two resident wavefronts per SIMD only interleave.
    // 512 threads (up to 2048 samples) still keep the optimiser state, the tables and the samples in 256 registers;
    // with 1024 threads (longer rows) the budget is 128 and a third of it spills
    const bool wide = N > 512 * kWgSpt, narrow = N <= 256 * kWgSpt && wg_narrow_enabled();
#define DSQ_WG_LAUNCH(PP)                                                                                            \
    do {                                                                                                             \
        if (narrow)                                                                                                  \
            hipLaunchKernelGGL((k_alpha_wg<PP, 256>), dim3(n_cap < 768 ? n_cap : 768), dim3(256), 0, st, y, ldn, N, list, \
                               n_dev, coef, sf, cells.cell_of, cells.Xc, cells.XX, min_mu, alpha_hat, prior_var,     \
                               prior_reg, alpha, conv, nfev, grid_count, grid_list, nll_const,                       \
                               (const Lbfgsb1d*)park_state);                                                         \
        else if (wide)                                                                                               \
            hipLaunchKernelGGL((k_alpha_wg<PP, 1024>), dim3(n_cap < 768 ? n_cap : 768), dim3(1024), 0, st, y, ldn, N, list, \
                               n_dev, coef, sf, cells.cell_of, cells.Xc, cells.XX, min_mu, alpha_hat, prior_var,     \
                               prior_reg, alpha, conv, nfev, grid_count, grid_list, nll_const,                       \
                               (const Lbfgsb1d*)park_state);                                                         \
        else                                                                                                         \
            hipLaunchKernelGGL((k_alpha_wg<PP, 512>), dim3(n_cap < 768 ? n_cap : 768), dim3(512), 0, st, y, ldn, N, list, \
                               n_dev, coef, sf, cells.cell_of, cells.Xc, cells.XX, min_mu, alpha_hat, prior_var,     \
                               prior_reg, alpha, conv, nfev, grid_count, grid_list, nll_const,                       \
                               (const Lbfgsb1d*)park_state);                                                         \
    } while (0)
    switch (P_) {
        case 1: DSQ_WG_LAUNCH(1); break;
        case 2: DSQ_WG_LAUNCH(2); break;
        case 3: DSQ_WG_LAUNCH(3); break;
        case 4: DSQ_WG_LAUNCH(4); break;
        default: return hipErrorInvalidValue;
    }
#undef DSQ_WG_LAUNCH
    return hipGetLastError();
}

size_t alpha_rows_smem(int N) {
    const int npad = (N + 63) & ~63;
    return (size_t)npad * 8 + (size_t)npad + row_slot_bytes(npad) * kRowSlots * kRowWaves +
           (size_t)kRowWaves * kRowTail * 4;
}

// can this (design, N) run on the row kernel?  (linear-model mu_hat with <= 4 cells == columns, rows short enough for
// two workgroups per CU, Cox-Reid term on)
bool alpha_rows_eligible(int N, int P_, int n_cells, bool has_coef, int cr_reg) {
    static const bool off = getenv("DSQ_NO_ALPHA_ROWS") != nullptr;
    return !off && has_coef && cr_reg != 0 && P_ >= 1 && P_ <= 4 && n_cells == P_ && N <= 65535 &&
           alpha_rows_smem(N) <= 78 * 1024;
}

hipError_t launch_alpha_rows(hipStream_t st, const int32_t* y, int ldn, int N, const int32_t* list, int n_list,
                             int32_t* queue, const double* coef, const double* sf, const CellDesign& cells, int P_,
                             double min_mu, const double* alpha_hat, double min_disp, double max_disp, double prior_var,
                             int cr_reg, int prior_reg, double* alpha, uint8_t* conv, int32_t* nfev, int32_t* grid_count,
                             int32_t* grid_list, double* nll_const, int const_mode, int eval_cap, void* park_state,
                             int32_t* park_count, int32_t* park_list) {
    if (n_list <= 0) return hipSuccess;
    const int n_cu = current_device_cus();
    if (n_cu <= 0) return hipGetLastError();
    const size_t smem = alpha_rows_smem(N);
    const int per_block = kRowSlots * kRowWaves;
    int blocks = (n_list + per_block - 1) / per_block;
    if (blocks > 2 * n_cu) blocks = 2 * n_cu;  // persistent: two workgroups per CU, genes come from the queue
    if (nll_const == nullptr) const_mode = DSQ_CONST_COMPUTE;
#define DSQ_ROWS_LAUNCH(PP)                                                                                          \
    do {                                                                                                             \
        if (smem > 48 * 1024) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)k_alpha_rows<PP>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                      (int)smem);                                                                    \
            (void)hipGetLastError();                                                                                 \
        }                                                                                                            \
        if (getenv("DSQ_DEBUG_ROWS")) {                                                                              \
            int nb = -1;                                                                                             \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_alpha_rows<PP>, kRowBlock, smem); \
            fprintf(stderr, "[k_alpha_rows<%d>] smem %zu blocks %d n_list %d occupancy %d blocks/CU\n", PP, smem,    \
                    blocks, n_list, nb);                                                                             \
        }                                                                                                            \
        hipLaunchKernelGGL(k_alpha_rows<PP>, dim3(blocks), dim3(kRowBlock), smem, st, y, ldn, N, list, n_list, queue, \
                           coef, sf, cells.cell_of, cells.Xc, cells.XX, min_mu, alpha_hat, min_disp, max_disp,       \
                           prior_var, cr_reg, prior_reg, alpha, conv, nfev, grid_count, grid_list, nll_const,        \
                           const_mode, eval_cap, (Lbfgsb1d*)park_state, park_count, park_list);                     \
    } while (0)
    switch (P_) {
        case 1: DSQ_ROWS_LAUNCH(1); break;
        case 2: DSQ_ROWS_LAUNCH(2); break;
        case 3: DSQ_ROWS_LAUNCH(3); break;
        case 4: DSQ_ROWS_LAUNCH(4); break;
        default: return hipErrorInvalidValue;
    }
#undef DSQ_ROWS_LAUNCH
    return hipGetLastError();
}

// per gene: -1 when the row / mixed kernels cannot take it (a count that does not fit their 16-bit staging: the slot-ordered
// copies of the mixed designs keep 0xFFFF for padding and 0xFFFE for "saturated", k_mix_counts_to_slots, so the largest
// count such a gene may hold is 65 533 - the same rule here, or a gene with a count of 65 534 / 65 535 would be fitted on
// its saturated copy), else the number of its
// samples with a count >= kRowTail (they cost a second sweep per evaluation: the host queues such genes together, first).
// Depends on the counts only: evaluated once per data set.
__global__ __launch_bounds__(kBlock) void k_count_big(const int32_t* __restrict__ y, int ldn, int N, int G,
                                                      int32_t* __restrict__ out) {
    const int g = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (g >= G) return;
    const int32_t* yg = y + (size_t)g * ldn;
    int nb = 0, mx = 0;
    for (int n = threadIdx.x & 63; n < N; n += 64) {
        const int v = yg[n];
        nb += v >= kRowTail ? 1 : 0;
        mx = v > mx ? v : mx;
    }
    nb = DeviceWave::sumi(nb);
    mx = DeviceWave::maxi(mx);
    if ((threadIdx.x & 63) == 0) out[g] = mx >= 0xFFFE ? -1 : nb;
}

hipError_t launch_count_big(hipStream_t st, const int32_t* y, int ldn, int N, int G, int32_t* out) {
    if (G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_count_big, dim3(genes_to_blocks(G)), dim3(kBlock), 0, st, y, ldn, N, G, out);
    return hipGetLastError();
}

}  // namespace dsq

#if defined(DSQ_ROWS_PHASES)
extern "C" int dsq_debug_rows_phase_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dsq::g_rows_phase), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(dsq::g_rows_phase), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
