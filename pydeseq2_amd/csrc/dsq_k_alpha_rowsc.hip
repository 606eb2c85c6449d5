// dsq_k_alpha_rowsc.hip — dispersion MLE / MAP (utils.py:441-564) with FOUR genes per wavefront for designs whose rows
// take up to 32 distinct values ("design cells": every purely categorical design, BASELINE configs[3] = 30 cells, p = 8).
//
// Same scheme as dsq_k_alpha_rows.hip (16-lane rows, persistent wavefronts with a device-side gene queue, gamma-function
// terms from per-gene tail counts, four samples per lane interleaved, parking of the fits that outlast the evaluation
// cap), with what a design of many cells and columns changes:
//   * mu_hat_n = size factor x per-cell value for BOTH mu_hat routes of dds.py:747-765: the linear model's
//     max(sf_n (x_c . coef), min_mu), and the IRLS route's UNclamped sf_n exp(x_c . beta) - the caller hands over the
//     per-cell exponentials (k_cell_mu), so the N x G mu_hat matrix of that route is neither written nor read;
//   * per-cell sums of w and dw in the row's LDS record (ds_add_f64; at most 16 lanes add at a time), the matrices
//     X^T W X and X^T dW X rebuilt entry-parallel by the row's lanes from the cells' outer products;
//   * the p x p algebra without the inverse: log det from the Cholesky factor, and
//         tr(M^-1 dM) = sum_k l_k dM l_k^T   (l_k = row k of L^-1)
//     so that at p = 8 about 80 doubles are live instead of the 144 of factor + inverse + both matrices (k_alpha<8>
//     spills: 0.74 GB of scratch writes per launch).
#include <cstdio>

#include "dsq_alpha_rows.h"
#include "dsq_dispatch.h"

namespace dsq {

constexpr int kRcCells = 32;  // design cells at most

struct RowGeneC {  // per-slot record in LDS (many-cell designs)
    Lbfgsb1d m;
    double cst, la_hat;
    int g;       // gene index, -1: the slot is empty
    int n_tail;  // tail-count entries in use, rounded up to the row width
    int n_big;   // samples with a count >= the table size
    int pad_;
    double cellv[kRcCells];   // mu_hat / size factor of the cells
    double acc[2][kRcCells];  // per-cell sums of w and dw of the current evaluation
};

DSQ_HD size_t rowc_slot_bytes(int npad, int ntail, int P) {
    return (sizeof(RowGeneC) + (size_t)npad * 2 + (size_t)ntail * 2 + (size_t)P * (P + 1) * 4 + 15) & ~(size_t)15;
}

template <int P>
__global__ __launch_bounds__(kRowBlock, 2) void k_alpha_rows_c(
    const int32_t* __restrict__ y, int ldn, int N, const int32_t* __restrict__ list, int n_list,
    int32_t* __restrict__ queue, const double* __restrict__ coef, const double* __restrict__ cell_mu,
    const double* __restrict__ sf, const int32_t* __restrict__ cell_of, const double* __restrict__ Xc,
    const double* __restrict__ XXc, int C, int ntail, double min_mu, const double* __restrict__ alpha_hat,
    double min_disp, double max_disp, double prior_var, int prior_reg, double* __restrict__ alpha_out,
    uint8_t* __restrict__ conv, int32_t* __restrict__ nfev, int32_t* __restrict__ grid_count,
    int32_t* __restrict__ grid_list, double* __restrict__ nll_const, int const_mode, int eval_cap,
    Lbfgsb1d* __restrict__ park_state, int32_t* __restrict__ park_count, int32_t* __restrict__ park_list) {
    constexpr int T = Tri<P>::N;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int npad = (N + 63) & ~63;
    const bool linear = coef != nullptr;  // mu_hat = max(sf q_c, min_mu); else sf * cell_mu (unclamped, dds.py:757-771)
    double* const sf_s = dyn;                                       // [npad], 0 beyond N
    double* const xx_s = sf_s + npad;                               // [C][T] outer products of the cells' rows
    double* const xc_s = xx_s + (size_t)kRcCells * T;               // [C][P] the cells' design rows
    uint8_t* const cell_s = (uint8_t*)(xc_s + (size_t)kRcCells * P);  // [npad]
    char* const slots0 = (char*)(cell_s + npad);
    const size_t slot_bytes = rowc_slot_bytes(npad, ntail, P);
    unsigned int* const hist0 = (unsigned int*)(slots0 + slot_bytes * kRowSlots * kRowWaves);  // [waves][ntail]

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = lane >> 4, rl = lane & 15;
    log_tab_fill();
    for (int n = threadIdx.x; n < npad; n += kRowBlock) {
        sf_s[n] = n < N ? sf[n] : 0.0;
        cell_s[n] = (uint8_t)(n < N ? cell_of[n] : 0);
    }
    for (int i = threadIdx.x; i < C * T; i += kRowBlock) xx_s[i] = XXc[i];
    for (int i = threadIdx.x; i < C * P; i += kRowBlock) xc_s[i] = Xc[i];
    __syncthreads();

    auto slot_of = [&](int r) { return (RowGeneC*)(slots0 + slot_bytes * (size_t)(w * kRowSlots + r)); };
    RowGeneC* const S = slot_of(row);
    uint16_t* const cnt = (uint16_t*)((char*)S + sizeof(RowGeneC));
    uint16_t* const tail = cnt + npad;
    double* const ent = (double*)(((uintptr_t)(tail + ntail) + 7) & ~(uintptr_t)7);  // [T] X^T W X, then its factor
    unsigned int* const hist = hist0 + (size_t)w * ntail;
    if (rl == 0) { S->g = -1; S->n_tail = 0; S->n_big = 0; }
    DeviceWave::sync();

    const double lo = log(min_disp), hi = log(max_disp);
    bool queue_open = true;
    for (;;) {
        // ---------------------------------------------------------------- refill: rows without a gene fetch one
        int mine = S->g;
        if (queue_open) {
            int want = -1;
            if (mine < 0 && rl == 0) {
                const int k = atomicAdd(queue, 1);
                want = k < n_list ? (list != nullptr ? list[k] : k) : -2;
            }
            unsigned long long todo = __ballot(want >= 0);
            if (__any(want == -2)) queue_open = false;
            while (todo) {  // wave-uniform: all 64 lanes stage the gene of one row
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int g = __shfl(want, src, 64);
                RowGeneC* const Sr = slot_of(src >> 4);
                uint16_t* const cr = (uint16_t*)((char*)Sr + sizeof(RowGeneC));
                uint16_t* const tr = cr + npad;
                for (int i = lane; i < ntail; i += 64) hist[i] = 0u;
                // per-cell mu_hat / size factor: lane c < C
                if (lane < kRcCells) {
                    double v = 0.0;
                    if (lane < C) {
                        if (linear) {
                            double yh = 0.0;
#pragma unroll
                            for (int j = 0; j < P; ++j) yh += Xc[lane * P + j] * coef[(size_t)g * P + j];
                            v = yh;
                        } else {
                            v = cell_mu[(size_t)g * C + lane];
                        }
                    }
                    Sr->cellv[lane] = v;
                }
                DeviceWave::sync();
                const int32_t* yg = y + (size_t)g * ldn;
                const bool want_cst = const_mode != DSQ_CONST_LOAD;
                KSum cs;
                int maxc = 0, nbig = 0;
                constexpr int CH = 8;  // count loads per lane in flight
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
                        const bool isbig = v >= ntail;
                        if (in && v > 0 && !isbig) atomicAdd(&hist[v], 1u);
                        const unsigned long long bm = __ballot(isbig);
                        if (isbig && want_cst) {  // lgamma(y + 1) - lgamma(ntail + 1): what the tail counts do not cover
                            const double z = (double)v + 1.0, zM = (double)ntail + 1.0;
                            cs.add(((z - 0.5) * flog(z) - z + stirling_tail(frcp(z))) -
                                   ((zM - 0.5) * flog(zM) - zM + stirling_tail(frcp(zM))));
                        }
                        nbig += __popcll(bm);
                        if (want_cst && in) {
                            double mu = sf_s[n] * Sr->cellv[cell_s[n]];
                            if (linear) mu = dmax(mu, min_mu);
                            cs.add(-((double)v * flog_t(mu)));
                        }
                    }
                }
                maxc = DeviceWave::maxi(maxc);
                DeviceWave::sync();
                {   // tail counts T_i = #{y > i}; sum_n lgamma(y_n + 1) = sum_i T_i log(i + 1) from the same walk
                    const int bpl = ntail / 64;  // 4 or 8 entries per lane
                    int h[8], tot = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { h[k] = k < bpl ? (int)hist[lane * bpl + k] : 0; tot += h[k]; }
                    const int below = DeviceWave::excl_scan_i(tot);
                    const int all = DeviceWave::sumi(tot);
                    int above = all - below - tot + nbig;
#pragma unroll
                    for (int k = 7; k >= 0; --k) {
                        if (k < bpl) {
                            tr[lane * bpl + k] = (uint16_t)above;
                            if (want_cst && above > 0) cs.add((double)above * flog_t((double)(lane * bpl + k + 1)));
                            above += h[k];
                        }
                    }
                }
                double cst = 0.0;
                if (want_cst) cst = DeviceWave::sum_comp(cs);
                else cst = nll_const[g];
                if (const_mode == DSQ_CONST_STORE && lane == 0) nll_const[g] = cst;
                if (lane == 0) {
                    Sr->g = g;
                    Sr->cst = cst;
                    const int mt = maxc < ntail ? maxc : ntail;
                    Sr->n_tail = (mt + kRowLanes - 1) & ~(kRowLanes - 1);
                    Sr->n_big = nbig;
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

        // ---------------------------------------------------------------- one evaluation per row
        const double la = active ? S->m.x : 0.0;
        const double alpha = exp(la);
        const double a = frcp(alpha);
        const double lal = flog_t(alpha);
        KSum accf;
        double accg = 0.0;
        for (int c = rl; c < kRcCells; c += kRowLanes) { S->acc[0][c] = 0.0; S->acc[1][c] = 0.0; }
        {
            const int ntl = active ? S->n_tail : 0;
            for (int i = rl; i < ntl; i += kRowLanes) {
                const double t = a + (double)i;
                const double ti = (double)tail[i];
                accf.add(-(ti * flog_t(t)));
                accg -= ti * frcp(t);
            }
            const int nb = active ? S->n_big : 0;
            if (__any(nb > 0)) {  // counts beyond the table (high-count genes): Stirling, sample by sample
                double lgM, psiM;
                stirling_big((double)ntail + a, lgM, psiM);
                if (nb > 0) {
                    for (int n = rl; n < npad; n += kRowLanes) {
                        const int yi = cnt[n];
                        if (yi >= ntail) {
                            double lgz, psiz;
                            stirling_big((double)yi + a, lgz, psiz);
                            accf.add(lgM - lgz);
                            accg += psiM - psiz;
                        }
                    }
                }
            }
        }
        DeviceWave::sync();  // the zeroed per-cell sums are visible before the first sample adds to them
        {
            constexpr int U = 4;
            KSum af[U];
            double ag[U];
#pragma unroll
            for (int u = 0; u < U; ++u) ag[u] = 0.0;
            for (int n0 = rl; n0 < npad; n0 += kRowLanes * U) {
                int yi[U], cl[U];
                double m[U], r1[U], L1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    yi[u] = cnt[n0 + kRowLanes * u];
                    cl[u] = cell_s[n0 + kRowLanes * u];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double sfn = sf_s[n0 + kRowLanes * u];
                    const double mm = sfn * S->cellv[cl[u]];
                    // padding beyond N has size factor 0: every term below is exactly zero (the clamp must not lift it)
                    m[u] = linear ? ((sfn == 0.0) ? 0.0 : dmax(mm, min_mu)) : mm;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) r1[u] = frcp(1.0 + m[u] * alpha);
                {
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
                    RowWave::cell_add(&S->acc[0][cl[u]], wv);
                    RowWave::cell_add(&S->acc[1][cl[u]], -(wv * wv));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { accf.merge(af[u].s, af[u].c); accg += ag[u]; }
        }
        const double sumf = RowWave::sum_comp(accf);
        accg = RowWave::sum(accg);
        double f = sumf + S->cst;
        double gr = alpha * (-(a * a * accg));
        DeviceWave::sync();  // all adds of the row have landed
        // Cox-Reid term and its derivative without a p x p matrix in any lane:
        //   M = X^T W X = sum_c w_c x_c x_c^T entry by entry (a lane owns entries e, e + 16, ... and walks the cells once),
        //   its Cholesky factor with lane i keeping row i (row broadcasts, as row_chol_solve of the IRLS kernel),
        //   log det M from the pivots, and  tr(M^-1 X^T dW X) = sum_c dw_c x_c^T M^-1 x_c = sum_c dw_c |L^-1 x_c|^2
        //   cell-parallel by forward substitution against the factor in LDS - X^T dW X is never formed.
        {
            constexpr int NE = (T + kRowLanes - 1) / kRowLanes;
            int em[NE];
            double vm[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) { em[k] = rl + k * kRowLanes; em[k] = em[k] < T ? em[k] : T - 1; vm[k] = 0.0; }
#pragma unroll 4
            for (int c = 0; c < C; ++c) {
                const double a0 = S->acc[0][c];
#pragma unroll
                for (int k = 0; k < NE; ++k) vm[k] += xx_s[c * T + em[k]] * a0;
            }
#pragma unroll
            for (int k = 0; k < NE; ++k)
                if (rl + k * kRowLanes < T) ent[rl + k * kRowLanes] = vm[k];
        }
        DeviceWave::sync();
        {
            const int ri = rl < P ? rl : P - 1;
            double arow[P], rinv[P];
#pragma unroll
            for (int j = 0; j < P; ++j) arow[j] = ent[tri(ri, j <= ri ? j : ri)];
            static_for<0, P>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const double r = frsq(RowWave::row_bcast<j>(arow[j]));
                rinv[j] = r;
                arow[j] *= r;
                static_for<j + 1, P>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    arow[k] -= arow[j] * RowWave::row_bcast<k>(arow[j]);
                });
            });
            {   // 0.5 log det M = sum_j log L_jj = -log prod_j (1 / L_jj)   (one logarithm, as chol_logdet)
                double p1 = 1.0, p2 = 1.0;
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    if (j < (P + 1) / 2) p1 *= rinv[j];
                    else p2 *= rinv[j];
                }
                const double pr = p1 * p2;
                double hl;
                if (pr > 1e-140 && pr < 1e140) hl = -flog(pr);
                else if (!(p1 > 0.0 && p1 < INFINITY && p2 > 0.0 && p2 < INFINITY)) hl = NAN;  // NaN / zero pivot
                else hl = -(flog(p1) + flog(p2));
                f += hl;
            }
            DeviceWave::sync();  // every lane has read its row of M: the factor may take its place
            if (rl < P) {
                const int base = rl * (rl + 1) / 2;
#pragma unroll
                for (int j = 0; j < P; ++j)
                    if (j <= rl) ent[base + j] = arow[j];
            }
            DeviceWave::sync();
            double trp = 0.0;
            for (int c = rl; c < C; c += kRowLanes) {
                double t[P], q = 0.0;
#pragma unroll
                for (int i = 0; i < P; ++i) t[i] = xc_s[c * P + i];
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    t[k] *= rinv[k];
                    q += t[k] * t[k];
#pragma unroll
                    for (int i = k + 1; i < P; ++i) t[i] -= ent[tri(i, k)] * t[k];
                }
                trp += S->acc[1][c] * q;
            }
            gr += 0.5 * RowWave::sum(trp) * alpha;
            DeviceWave::sync();  // ent is rewritten by the next evaluation
        }
        if (prior_reg != 0) {
            const double dl = la - S->la_hat;
            f += dl * dl / (2.0 * prior_var);
            gr += dl / prior_var;
        }
        // ---------------------------------------------------------------- optimiser step, results of finished genes
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
            } else if (eval_cap > 0 && S->m.nfev >= eval_cap) {  // park (dsq_k_alpha_rows.hip): k_alpha continues the fit
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
}

// LDS of one workgroup for (N, P, table size)
static size_t rowsc_smem(int N, int P, int ntail) {
    const int npad = (N + 63) & ~63;
    return (size_t)npad * 8 + (size_t)kRcCells * (P * (P + 1) / 2 + P) * 8 + (size_t)npad +
           rowc_slot_bytes(npad, ntail, P) * kRowSlots * kRowWaves + (size_t)kRowWaves * ntail * 4 + 64;
}

// tail-count table size that lets two workgroups share a CU's LDS (0: the design / sample count does not fit)
int alpha_rowsc_tail(int N, int P_, int n_cells) {
    static const bool off = getenv("DSQ_NO_ALPHA_ROWSC") != nullptr;
    if (off || P_ < 1 || P_ > 8 || n_cells < 1 || n_cells > kRcCells || N > 65535) return 0;
    for (int nt : {512, 256})
        if (rowsc_smem(N, P_, nt) <= 78 * 1024) return nt;
    return 0;
}

hipError_t launch_alpha_rows_c(hipStream_t st, const int32_t* y, int ldn, int N, const int32_t* list, int n_list,
                               int32_t* queue, const double* coef, const double* cell_mu, const double* sf,
                               const CellDesign& cells, int P_, double min_mu, const double* alpha_hat, double min_disp,
                               double max_disp, double prior_var, int prior_reg, double* alpha, uint8_t* conv,
                               int32_t* nfev, int32_t* grid_count, int32_t* grid_list, double* nll_const,
                               int const_mode, int eval_cap, void* park_state, int32_t* park_count,
                               int32_t* park_list) {
    if (n_list <= 0) return hipSuccess;
    const int ntail = alpha_rowsc_tail(N, P_, cells.C);
    if (ntail == 0 || (coef == nullptr) == (cell_mu == nullptr)) return hipErrorInvalidValue;
    const int n_cu = current_device_cus();
    if (n_cu <= 0) return hipGetLastError();
    const size_t smem = rowsc_smem(N, P_, ntail);
    const int per_block = kRowSlots * kRowWaves;
    int blocks = (n_list + per_block - 1) / per_block;
    if (blocks > 2 * n_cu) blocks = 2 * n_cu;
    if (nll_const == nullptr) const_mode = DSQ_CONST_COMPUTE;
    DSQ_DISPATCH_P(P_, {
        if constexpr (P <= 8) {
            if (smem > 48 * 1024) {
                (void)hipFuncSetAttribute((const void*)k_alpha_rows_c<P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)smem);
                (void)hipGetLastError();
            }
            if (getenv("DSQ_DEBUG_ROWS")) {
                int nb = -1;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_alpha_rows_c<P>, kRowBlock, smem);
                fprintf(stderr, "[k_alpha_rows_c<%d>] smem %zu ntail %d blocks %d n_list %d occupancy %d blocks/CU\n", P,
                        smem, ntail, blocks, n_list, nb);
            }
            hipLaunchKernelGGL(k_alpha_rows_c<P>, dim3(blocks), dim3(kRowBlock), smem, st, y, ldn, N, list, n_list,
                               queue, coef, cell_mu, sf, cells.cell_of, cells.Xc, cells.XX, cells.C, ntail, min_mu,
                               alpha_hat, min_disp, max_disp, prior_var, prior_reg, alpha, conv, nfev, grid_count,
                               grid_list, nll_const, const_mode, eval_cap, (Lbfgsb1d*)park_state, park_count, park_list);
        }
    })
    return hipGetLastError();
}

// per-cell mu_hat / size factor of the IRLS route: cell_mu[g][c] = exp(x_c . beta_g)   (dds.py:757-771: the UNclamped
// sf * exp(X beta) that irls_solver returns, utils.py:435-437, is sf_n times this value for every sample n of cell c)
template <int P>
__global__ void k_cell_mu(const double* __restrict__ beta, const double* __restrict__ Xc, int C, int G,
                          double* __restrict__ cell_mu) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * C) return;
    const int g = t / C, c = t % C;
    double eta = 0.0;
#pragma unroll
    for (int j = 0; j < P; ++j) eta += Xc[c * P + j] * beta[(size_t)g * P + j];
    cell_mu[t] = exp(eta);
}

hipError_t launch_cell_mu(hipStream_t st, const double* beta, const double* Xc, int C, int G, int P_, double* cell_mu) {
    if (G <= 0 || C <= 0) return hipSuccess;
    const int total = G * C;
    DSQ_DISPATCH_P(P_, hipLaunchKernelGGL(k_cell_mu<P>, dim3((total + 255) / 256), dim3(256), 0, st, beta, Xc, C, G,
                                          cell_mu))
    return hipGetLastError();
}

// rows of mu_hat = sf * cell_mu[cell] for a gene list (grid-search pass of designs whose mu_hat is not materialised)
__global__ __launch_bounds__(kBlock) void k_mu_from_cells(const double* __restrict__ cell_mu, int C,
                                                          const double* __restrict__ sf,
                                                          const int32_t* __restrict__ cell_of, int N,
                                                          const int32_t* __restrict__ list, int n_list,
                                                          double* __restrict__ dst, int ldn,
                                                          int32_t* __restrict__ idx_out,
                                                          const int32_t* __restrict__ n_dev) {
    const int k = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (n_dev != nullptr) n_list = min(n_list, *n_dev);
    if (k >= n_list) return;
    const int g = list[k];
    for (int n = threadIdx.x & 63; n < N; n += 64) dst[(size_t)k * ldn + n] = sf[n] * cell_mu[(size_t)g * C + cell_of[n]];
    if ((threadIdx.x & 63) == 0) idx_out[k] = k;
}

hipError_t launch_mu_from_cells(hipStream_t st, const double* cell_mu, int C, const double* sf, const int32_t* cell_of,
                                int N, const int32_t* list, int n_list, double* dst, int ldn, int32_t* idx_out,
                                const int32_t* n_dev) {
    if (n_list <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_mu_from_cells, dim3(genes_to_blocks(n_list)), dim3(kBlock), 0, st, cell_mu, C, sf, cell_of, N,
                       list, n_list, dst, ldn, idx_out, n_dev);
    return hipGetLastError();
}

}  // namespace dsq
