// dsq_k_alpha_mix.hip — dispersion MLE / MAP (utils.py:441-564) for MIXED designs: categorical columns with few distinct
// rows + up to three continuous covariates (dsq_mix.h; BASELINE configs[4]: 60 000 x 5000, p = 8).
//
// One gene per wavefront, persistent wavefronts with a device-side gene queue.  What differs from k_alpha (the general
// one-gene-per-wavefront kernel these designs used to run, 36 + 36 accumulator FMAs and 8 design loads per sample and
// evaluation at p = 8, rows re-read from L2 / HBM by every evaluation):
//   * samples are walked in SLOT order (sorted by design cell, cells padded to whole trips): the cell of a trip is
//     wave-uniform, X^T W X and X^T dW X come from 2 (1 + Q + Q (Q + 1) / 2) = 20 register accumulators per lane that are
//     folded into the matrices when the cell changes (one multi-value butterfly per cell and evaluation; lane e owns
//     matrix entry e);
//   * the gene's counts and its mu_hat row come from SLOT-ORDERED copies (round 5: ys [G][Ns] uint16 written once per count
//     matrix, mu_s [G][Ns] fp64 written once per fit by k_mix_mu_slots from the IRLS coefficients - dds.py:757-771: the
//     UNclamped sf * exp(X beta) - or by k_mix_f64_to_slots from the plug-in caller's matrix): staging is a contiguous
//     copy of 2 B per slot into LDS plus one streaming pass over the mu_hat row for the NLL constant, no gather through
//     the slot permutation, no exponential; every evaluation streams the mu_hat row again (L2 / MALL);
//   * gamma-function terms from per-gene tail counts (dsq_k_alpha_rows.hip): no lgamma / digamma per sample;
//   * two trips per loop iteration, the next iteration's loads issued ahead of the arithmetic (software pipelining);
//   * fits that outlast the evaluation cap are parked and continued by a second launch of this kernel (resume != 0),
//     so that a launch is not as long as its slowest 34-evaluation fit.
// Compiled once per number of continuous covariates (-DDSQ_MIX_Q=1|2|3): three translation units build in parallel.
#include <cstdio>

#include "dsq_alpha_rows.h"
#include "dsq_mix.h"

#ifndef DSQ_MIX_Q
#error "compile with -DDSQ_MIX_Q=1, 2 or 3"
#endif

namespace dsq {

// developer build (make mixph, tools/mix_phase_probe.py): per-phase cycle accounting of k_alpha_mix
#if defined(DSQ_MIX_PHASES) && DSQ_MIX_Q == 3
__device__ unsigned long long g_mix_phase[12];
#endif
#if defined(DSQ_MIX_PHASES)
#if DSQ_MIX_Q != 3
extern __device__ unsigned long long g_mix_phase[12];
#endif
#define MIX_PH(k)                                                       \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const long long t_ = clock64();                                 \
        ph_acc[ph_cur] += t_ - ph_last; ph_last = t_; ph_cur = (k);     \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#else
#define MIX_PH(k) ((void)0)
#endif

struct MixWaveLds {  // wave-private LDS record (followed by the gene's counts, uint16 [Ns])
    Lbfgsb1d m;
    double ent[2 * (kMixMaxP * (kMixMaxP + 1) / 2)]; // matrix entries on their way from the owning lane to all lanes
    unsigned int hist[kMixTail];
    uint16_t tail[kMixTail];
};
static_assert(sizeof(MixWaveLds) % 8 == 0, "the counts follow the record");

DSQ_HD size_t mix_wave_bytes(int Ns) { return (sizeof(MixWaveLds) + (size_t)Ns * 2 + 15) & ~(size_t)15; }
// the one-gene-per-workgroup continuation (WG): two (double-buffered) sets of per-wavefront partial sums - the matrix
// entries of X^T W X and X^T dW X, the loss and the gradient sum - and the gene index of the workgroup
constexpr int kMixRedStride = 2 * (kMixMaxP * (kMixMaxP + 1) / 2) + 8;
constexpr int kMixRedWaves = 4;
constexpr size_t kMixRedBytes = (size_t)2 * kMixRedWaves * kMixRedStride * 8 + 16;
DSQ_HD size_t mix_shared_bytes(int Ns, int P) {
    return (size_t)kMixMaxCells * P * 8 + (size_t)(((Ns >> 6) + 15) & ~15) + kMixRedBytes;
}

// WG = false: one gene per wavefront (the full-size launch).  WG = true: one gene per WORKGROUP - the continuation of the
// parked fits (the 1 % of the genes whose line search takes up to 34 evaluations): with a wavefront per gene that launch was
// as long as its slowest fit, 26 more evaluations of ~55 us each at 5000 samples = 1.4-1.7 ms for 75 genes, longer than the
// full-size launch before it (profiles/r05_timeline_c5_shard.txt).  The wavefronts of a workgroup take every nw-th loop
// iteration of the same gene (an iteration lies inside one design cell), reduce their partial sums through LDS in a fixed
// order and all step their own copy of the optimiser with the same totals (as k_alpha_wg does for the row kernel).
template <int P, int Q, bool WG = false>
__global__ __launch_bounds__(256, 2) void k_alpha_mix(
    const uint16_t* __restrict__ ys, const double* __restrict__ mu_s, const MixDesign D, unsigned cont_mask,
    const int32_t* __restrict__ list, int n_list, const int32_t* __restrict__ n_dev, int32_t* __restrict__ queue,
    const double* __restrict__ alpha_hat, double min_disp, double max_disp, double prior_var, int prior_reg,
    double* __restrict__ alpha_out, uint8_t* __restrict__ conv, int32_t* __restrict__ nfev,
    int32_t* __restrict__ grid_count, int32_t* __restrict__ grid_list, double* __restrict__ nll_const, int const_mode,
    int eval_cap, int resume, Lbfgsb1d* __restrict__ park_state, int32_t* __restrict__ park_count,
    int32_t* __restrict__ park_list) {
    constexpr int T = Tri<P>::N;
    constexpr int NS = 2 * (1 + Q);       // per-cell sums: w, w z_q | dw, dw z_q
    constexpr int QQ = Q * (Q + 1) / 2;   // continuous x continuous block
    constexpr int U = kMixU;
    static_assert(Q >= 1 && Q <= kMixMaxQ && P >= Q && P <= kMixMaxP, "design shape");
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int Ns = D.Ns, ntrips = Ns >> 6, C = D.C;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wq = WG ? w : 0, nq = WG ? (int)(blockDim.x >> 6) : 1;  // this wavefront's share of a gene's loop iterations
    double* const xc_s = dyn;                                       // [C][P] (continuous columns 0)
    uint8_t* const tc_s = (uint8_t*)(xc_s + kMixMaxCells * P);      // [ntrips]
    double* const red_s = (double*)((char*)dyn + mix_shared_bytes(Ns, P) - kMixRedBytes);
    int* const gene_s = (int*)(red_s + 2 * kMixRedWaves * kMixRedStride);
    int red_parity = 0;
    char* const wbase = (char*)dyn + mix_shared_bytes(Ns, P) + mix_wave_bytes(Ns) * (size_t)w;
    MixWaveLds* const L = (MixWaveLds*)wbase;
    uint16_t* const y16 = (uint16_t*)(wbase + sizeof(MixWaveLds));
    if (n_dev != nullptr) n_list = min(n_list, *n_dev);  // launched for a capacity, the count is on the device

    log_tab_fill();
    for (int i = threadIdx.x; i < C * P; i += blockDim.x) xc_s[i] = D.Xc[i];
    for (int i = threadIdx.x; i < ntrips; i += blockDim.x) tc_s[i] = D.trip_cell[i];
    __syncthreads();

    // lane e owns entry e = tri(ei, ej) of the two matrices: categorical x categorical (kind 0), categorical x continuous
    // (kind 1: column xa categorical, covariate qz) or continuous x continuous (kind 2: entry zzk of the packed block)
    int ei = 0, ej = 0;
    {
        const int e = lane < T ? lane : T - 1;
        while ((ei + 1) * (ei + 2) / 2 <= e) ++ei;
        ej = e - ei * (ei + 1) / 2;
    }
    const bool ci = ((cont_mask >> ei) & 1u) != 0, cj = ((cont_mask >> ej) & 1u) != 0;
    const int qi = __popc(cont_mask & ((1u << ei) - 1u)), qj = __popc(cont_mask & ((1u << ej) - 1u));
    const int kind = (ci ? 1 : 0) + (cj ? 1 : 0);
    const int xa = ci ? ej : ei, xb = ej, qz = ci ? qi : qj;
    const int zzk = tri(qi > qj ? qi : qj, qi > qj ? qj : qi);
    // v[k] for a lane-varying k WITHOUT taking the address of a register array (a pointer to it - even into an inlined
    // lambda - keeps the whole array in scratch memory: the first build re-loaded and re-stored the per-cell sums through
    // scratch in every loop iteration)
    auto pick3 = [](int k, double v0, double v1, double v2) { return k == 1 ? v1 : (k == 2 ? v2 : v0); };
    auto pick6 = [](int k, double v0, double v1, double v2, double v3, double v4, double v5) {
        double r = v0;
        r = k == 1 ? v1 : r;
        r = k == 2 ? v2 : r;
        r = k == 3 ? v3 : r;
        r = k == 4 ? v4 : r;
        r = k == 5 ? v5 : r;
        return r;
    };
    constexpr int q1 = Q > 1 ? 1 : 0, q2 = Q > 2 ? 2 : 0;  // (clamped indices: the unused operands of a pick)

    const double lo = log(min_disp), hi = log(max_disp);
#if defined(DSQ_MIX_PHASES)
    long long ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_last = clock64();
    int ph_cur = 0;
#endif
    for (;;) {
        MIX_PH(0);
        int k = 0;
        if constexpr (WG) {  // one gene for the whole workgroup
            if (threadIdx.x == 0) *gene_s = atomicAdd(queue, 1);
            __syncthreads();
            k = *gene_s;
            __syncthreads();
        } else {
            if (lane == 0) k = atomicAdd(queue, 1);
            k = __builtin_amdgcn_readfirstlane(k);
        }
        if (k >= n_list) break;
        const int g = list != nullptr ? list[k] : k;
        // ------------------------------------------------------------------------------------------ stage the gene
        MIX_PH(2);
        for (int i = lane; i < kMixTail; i += 64) L->hist[i] = 0u;
        DeviceWave::sync();
        const uint16_t* const yg = ys + (size_t)g * Ns;
        const double* const mus = mu_s + (size_t)g * Ns;  // mu_hat in slot order, 0 in padding slots
        const bool want_cst = !(const_mode == DSQ_CONST_LOAD || resume != 0);
        KSum cs;
        int maxc = 0, nbig = 0;
        constexpr int CH = 4;  // contiguous loads per lane in flight (Ns is a multiple of 256)
        for (int base = 0; base < Ns; base += 64 * CH) {
            int v4[CH];
            double mv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int s = base + 64 * c + lane;
                v4[c] = yg[s];
                mv[c] = want_cst ? mus[s] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int s = base + 64 * c + lane;
                const bool valid = v4[c] != 0xFFFF;  // (padding slot of the slot-ordered copy)
                const int v = valid ? v4[c] : 0;
                y16[s] = (uint16_t)v;
                maxc = v > maxc ? v : maxc;
                const bool isbig = v >= kMixTail;
                if (v > 0 && !isbig) atomicAdd(&L->hist[v], 1u);
                const unsigned long long bm = __ballot(isbig);
                if (isbig && want_cst) {  // lgamma(y + 1) - lgamma(kMixTail + 1): what the tail counts do not cover
                    const double z = (double)v + 1.0, zM = (double)kMixTail + 1.0;
                    cs.add(((z - 0.5) * flog(z) - z + stirling_tail(frcp(z))) -
                           ((zM - 0.5) * flog(zM) - zM + stirling_tail(frcp(zM))));
                }
                nbig += __popcll(bm);
                if (want_cst && valid) cs.add(-((double)v * flog_t(mv[c])));
            }
        }
        MIX_PH(3);
        maxc = DeviceWave::maxi(maxc);
        DeviceWave::sync();
        {   // tail counts T_i = #{y > i}; sum_n lgamma(y_n + 1) = sum_i T_i log(i + 1) from the same walk
            constexpr int BPL = kMixTail / 64;
            int h[BPL], tot = 0;
#pragma unroll
            for (int i = 0; i < BPL; ++i) { h[i] = (int)L->hist[lane * BPL + i]; tot += h[i]; }
            const int below = DeviceWave::excl_scan_i(tot);
            const int all = DeviceWave::sumi(tot);
            int above = all - below - tot + nbig;
#pragma unroll
            for (int i = BPL - 1; i >= 0; --i) {
                L->tail[lane * BPL + i] = (uint16_t)above;
                if (want_cst && above > 0) cs.add((double)above * flog_t((double)(lane * BPL + i + 1)));
                above += h[i];
            }
        }
        double cst;
        if (want_cst) cst = DeviceWave::sum_comp(cs);
        else cst = nll_const[g];
        cst = DeviceWave::uniform(cst);
        if (const_mode == DSQ_CONST_STORE && resume == 0 && lane == 0) nll_const[g] = cst;
        const int n_tail = ((maxc < kMixTail ? maxc : kMixTail) + 63) & ~63;
        const double la_hat = DeviceWave::uniform(log(alpha_hat[g]));
        if (resume != 0) {
            constexpr int kDw = (int)(sizeof(Lbfgsb1d) / 4);
            const uint32_t* src = (const uint32_t*)(park_state + g);
            uint32_t* dst = (uint32_t*)&L->m;
            for (int i = lane; i < kDw; i += 64) dst[i] = src[i];
        } else {
            L->m.start(la_hat, lo, hi);
        }
        DeviceWave::sync();

        // ------------------------------------------------------------------------------------------ the fit
        int budget = (eval_cap > 0 && resume == 0) ? eval_cap : 0x7fffffff;
        while (!L->m.done && budget > 0) {
            --budget;
            MIX_PH(4);
            const double la = DeviceWave::uniform(L->m.x);
            const double alpha = DeviceWave::uniform(exp(la));
            const double a = DeviceWave::uniform(frcp(alpha));
            const double lal = DeviceWave::uniform(flog_t(alpha));  // log of the ROUNDED alpha (see alpha_eval_body)
            KSum accf;
            double accg = 0.0;
            for (int i = lane + 64 * wq; i < n_tail; i += 64 * nq) {  // gamma-function terms from the tail counts
                const double t = a + (double)i;
                const double ti = (double)L->tail[i];
                accf.add(-(ti * flog_t(t)));
                accg -= ti * frcp(t);
            }
            if (nbig > 0) {  // counts beyond the table (high-count genes): Stirling, sample by sample
                double lgM, psiM;
                stirling_big((double)kMixTail + a, lgM, psiM);
                for (int s = lane + 64 * wq; s < Ns; s += 64 * nq) {
                    const int yi = y16[s];
                    if (yi >= kMixTail) {
                        double lgz, psiz;
                        stirling_big((double)yi + a, lgz, psiz);
                        accf.add(lgM - lgz);
                        accg += psiM - psiz;
                    }
                }
            }
            double sc[NS], zz[2 * QQ], Me = 0.0, dMe = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) sc[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 2 * QQ; ++i) zz[i] = 0.0;
            // the per-cell sums of the cell that ends here go into the matrix entries (wave-uniform branch)
            auto fold = [&](int c) {
                DeviceWave::template sum_n<NS>(sc);
                const double va = xc_s[c * P + xa], vb = xc_s[c * P + xb];
                const double s1 = pick3(qz, sc[1], sc[1 + q1], sc[1 + q2]);
                const double ds1 = pick3(qz, sc[2 + Q], sc[2 + Q + q1], sc[2 + Q + q2]);
                const double cc = va * vb;
                Me += kind == 0 ? cc * sc[0] : (kind == 1 ? va * s1 : 0.0);
                dMe += kind == 0 ? cc * sc[1 + Q] : (kind == 1 ? va * ds1 : 0.0);
#pragma unroll
                for (int i = 0; i < NS; ++i) sc[i] = 0.0;
            };
            const int t_first = U * wq < ntrips ? U * wq : 0;  // (a wavefront without an iteration: the loop below is empty)
            int cur = __builtin_amdgcn_readfirstlane((int)tc_s[t_first]);
            KSum af[U];
            double ag[U];
#pragma unroll
            for (int u = 0; u < U; ++u) ag[u] = 0.0;
            // Software-pipelined: the loads of iteration i + 1 (mu_hat row and covariates from L2, counts from LDS) are
            // issued before the arithmetic of iteration i (~80 fp64 instructions per sample), so that their latency is
            // not exposed with two wavefronts per SIMD.  A loop iteration lies inside ONE design cell (padding rule of
            // dsq_mix.h): the cell is checked once per iteration.
            int yn[U];
            double mn[U], zn[U][Q];
            auto issue = [&](int t0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s = (t0 + u) * 64 + lane;
                    yn[u] = y16[s];
                    mn[u] = mus[s];
#pragma unroll
                    for (int q = 0; q < Q; ++q) zn[u][q] = D.Zs[(size_t)q * Ns + s];
                }
            };
            MIX_PH(5);
            issue(t_first);
            for (int t0 = U * wq; t0 < ntrips; t0 += U * nq) {
                int yi[U];
                double m[U], z[U][Q], r1[U], L1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    yi[u] = yn[u];
                    m[u] = mn[u];
#pragma unroll
                    for (int q = 0; q < Q; ++q) z[u][q] = zn[u][q];
                }
                issue(t0 + U * nq < ntrips ? t0 + U * nq : t0);  // (the last iteration re-reads its own slots: no branch)
                const int cell = __builtin_amdgcn_readfirstlane((int)tc_s[t0]);
                if (cell != cur) {
                    fold(cur);
                    cur = cell;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) r1[u] = frcp(1.0 + m[u] * alpha);
                {   // flog1p_t(m alpha, r1) (dsq_math.h) for the samples of the iteration, the table reads issued together
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
                    sc[0] += wv;
                    sc[1 + Q] += dwv;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const double wz = wv * z[u][q], dwz = dwv * z[u][q];
                        sc[1 + q] += wz;
                        sc[2 + Q + q] += dwz;
#pragma unroll
                        for (int q2 = 0; q2 <= q; ++q2) {
                            zz[tri(q, q2)] = fma(wz, z[u][q2], zz[tri(q, q2)]);
                            zz[QQ + tri(q, q2)] = fma(dwz, z[u][q2], zz[QQ + tri(q, q2)]);
                        }
                    }
                }
            }
            MIX_PH(6);
            fold(cur);
            DeviceWave::template sum_n<2 * QQ>(zz);
            if (kind == 2) {
                constexpr int L_ = QQ - 1;  // last entry of the packed block
                Me = pick6(zzk, zz[0], zz[1 < L_ ? 1 : L_], zz[2 < L_ ? 2 : L_], zz[3 < L_ ? 3 : L_], zz[4 < L_ ? 4 : L_], zz[L_]);
                dMe = pick6(zzk, zz[QQ], zz[QQ + (1 < L_ ? 1 : L_)], zz[QQ + (2 < L_ ? 2 : L_)], zz[QQ + (3 < L_ ? 3 : L_)],
                            zz[QQ + (4 < L_ ? 4 : L_)], zz[QQ + L_]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { accf.merge(af[u].s, af[u].c); accg += ag[u]; }
            double sumf = DeviceWave::sum_comp(accf);
            accg = DeviceWave::sum(accg);
            if constexpr (WG) {
                // partial sums of the workgroup's wavefronts -> totals, summed in wavefront order by everybody (so that
                // every wavefront steps its optimiser copy with the same numbers); double-buffered: one barrier per evaluation
                double* const red = red_s + (size_t)red_parity * kMixRedWaves * kMixRedStride;
                double* const mine = red + w * kMixRedStride;
                if (lane < T) { mine[lane] = Me; mine[T + lane] = dMe; }
                if (lane == 0) { mine[2 * T] = sumf; mine[2 * T + 1] = accg; }
                __syncthreads();
                double m2 = 0.0, d2 = 0.0, f2 = 0.0, g2 = 0.0;
                for (int q = 0; q < nq; ++q) {
                    const double* r = red + q * kMixRedStride;
                    if (lane < T) { m2 += r[lane]; d2 += r[T + lane]; }
                    f2 += r[2 * T];
                    g2 += r[2 * T + 1];
                }
                Me = m2; dMe = d2; sumf = f2; accg = g2;
                red_parity ^= 1;
            }
            double f = sumf + cst;
            double gr = alpha * (-(a * a * accg));
            if (lane < T) { L->ent[lane] = Me; L->ent[T + lane] = dMe; }
            DeviceWave::sync();
            MIX_PH(7);
            {   // Cox-Reid term 0.5 log det M and its derivative 0.5 alpha tr(M^-1 dM): the factor, its inverse factor and
                // then dM are the register peak (two packed matrices, not four: tr(M^-1 dM) = sum_k l_k dM l_k^T, l_k the
                // rows of L^-1 - no inverse matrix)
                double M[T], li[T];
#pragma unroll
                for (int i = 0; i < T; ++i) M[i] = L->ent[i];
                chol<P>(M);
                f += 0.5 * chol_logdet<P>(M);
                tri_inverse<P>(M, li);
#pragma unroll
                for (int i = 0; i < T; ++i) M[i] = L->ent[T + i];
                DeviceWave::sync();  // ent is rewritten by the next evaluation
                gr += 0.5 * trace_inv_times<P>(li, M) * alpha;
            }
            if (prior_reg != 0) {
                const double dl = la - la_hat;
                f += dl * dl / (2.0 * prior_var);
                gr += dl / prior_var;
            }
            MIX_PH(8);
            L->m.feed(f, gr);
            DeviceWave::sync();
        }
        MIX_PH(9);
        // ------------------------------------------------------------------------------------------ result / parking
        if (WG && w != 0) {
            // (the workgroup's other wavefronts hold the same result)
        } else if (!L->m.done) {  // out of this launch's evaluation budget: the continuation launch resumes the gene
            constexpr int kDw = (int)(sizeof(Lbfgsb1d) / 4);
            uint32_t* dst = (uint32_t*)(park_state + g);
            const uint32_t* src = (const uint32_t*)&L->m;
            for (int i = lane; i < kDw; i += 64) dst[i] = src[i];
            if (lane == 0) park_list[atomicAdd(park_count, 1)] = g;
        } else if (lane == 0) {
            alpha_out[g] = exp(L->m.x);
            conv[g] = (uint8_t)(L->m.success ? 1 : 0);
            if (nfev != nullptr) nfev[g] = L->m.nfev;
            if (!L->m.success) grid_list[atomicAdd(grid_count, 1)] = g;
        }
        DeviceWave::sync();
    }
#if defined(DSQ_MIX_PHASES)
    MIX_PH(0);
    if (lane == 0) {
        unsigned long long life = 0;
        for (int i = 0; i < 10; ++i) { atomicAdd(&g_mix_phase[i], (unsigned long long)ph_acc[i]); life += ph_acc[i]; }
        atomicMax(&g_mix_phase[10], life);
        atomicAdd(&g_mix_phase[11], 1ull);
    }
#endif
}

// wavefronts per workgroup for rows of Ns slots: four when two workgroups still share a CU's LDS, else fewer (0: too long)
static int mix_waves_per_block(int Ns, int P) {
    for (int nw : {4, 2, 1}) {
        const size_t smem = mix_shared_bytes(Ns, P) + mix_wave_bytes(Ns) * nw + 64;
        if (2 * smem <= 156 * 1024) return nw;
    }
    return mix_shared_bytes(Ns, P) + mix_wave_bytes(Ns) + 64 <= 64 * 1024 ? 1 : 0;
}

#define DSQ_MIX_CAT_(a, b) a##b
#define DSQ_MIX_CAT(a, b) DSQ_MIX_CAT_(a, b)

// grid of the persistent launch for n_list genes: (workgroups, wavefronts per workgroup); workgroups = 0: not eligible
void DSQ_MIX_CAT(alpha_mix_grid_q, DSQ_MIX_Q)(int Ns, int P, int n_list, int* blocks, int* nw_out) {
    *blocks = 0; *nw_out = 0;
    const int nw = mix_waves_per_block(Ns, P);
    const int n_cu = current_device_cus();
    if (nw == 0 || n_cu <= 0 || n_list <= 0) return;
    const size_t smem = mix_shared_bytes(Ns, P) + mix_wave_bytes(Ns) * nw + 64;
    int per_cu = (int)((156 * 1024) / smem);
    if (per_cu * nw > 8) per_cu = 8 / nw;  // 256 VGPRs: two wavefronts per SIMD
    if (per_cu < 1) per_cu = 1;
    int b = (n_list + nw - 1) / nw;
    if (b > per_cu * n_cu) b = per_cu * n_cu;
    *blocks = b; *nw_out = nw;
}

hipError_t DSQ_MIX_CAT(launch_alpha_mix_q, DSQ_MIX_Q)(
    hipStream_t st, const uint16_t* ys, const double* mu_s, const MixDesign& D, const int32_t* list, int n_list,
    const int32_t* n_dev, int32_t* queue, const double* alpha_hat, double min_disp, double max_disp, double prior_var,
    int prior_reg, double* alpha, uint8_t* conv, int32_t* nfev, int32_t* grid_count, int32_t* grid_list,
    double* nll_const, int const_mode, int eval_cap, int resume, void* park_state, int32_t* park_count,
    int32_t* park_list) {
    constexpr int Q = DSQ_MIX_Q;
    if (n_list <= 0) return hipSuccess;
    if (D.Q != Q || D.P < Q || D.P > kMixMaxP || ys == nullptr || mu_s == nullptr) return hipErrorInvalidValue;
    int blocks = 0, nw = 0;
    DSQ_MIX_CAT(alpha_mix_grid_q, DSQ_MIX_Q)(D.Ns, D.P, n_list, &blocks, &nw);
    if (blocks == 0) return hipErrorInvalidValue;
    const size_t smem = mix_shared_bytes(D.Ns, D.P) + mix_wave_bytes(D.Ns) * nw + 64;
    unsigned cont_mask = 0;
    for (int q = 0; q < Q; ++q) cont_mask |= 1u << D.zcol[q];
    if (nll_const == nullptr) const_mode = DSQ_CONST_COMPUTE;
    // the continuation of the parked fits: one gene per workgroup (DSQ_MIX_WG_CONT=0: a wavefront per gene, as the main launch)
    static const bool wg_cont = !(getenv("DSQ_MIX_WG_CONT") && atoi(getenv("DSQ_MIX_WG_CONT")) == 0);
    const bool wg = resume != 0 && wg_cont && nw > 1;
    if (wg) {  // a workgroup per listed gene (the list's length is a capacity: the count is on the device)
        const int n_cu = current_device_cus();
        int per_cu = (int)((156 * 1024) / smem);
        if (per_cu * nw > 8) per_cu = 8 / nw;
        if (per_cu < 1) per_cu = 1;
        blocks = n_list < per_cu * n_cu ? n_list : per_cu * n_cu;
    }
#define DSQ_MIX_LAUNCH_K(PP, WG_)                                                                                       \
    do {                                                                                                                \
        if (smem > 48 * 1024) {                                                                                         \
            (void)hipFuncSetAttribute((const void*)k_alpha_mix<PP, Q, WG_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)smem);                                                                       \
            (void)hipGetLastError();                                                                                    \
        }                                                                                                               \
        if (getenv("DSQ_DEBUG_ROWS")) {                                                                                 \
            int nb = -1;                                                                                                \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_alpha_mix<PP, Q, WG_>, 64 * nw, smem); \
            fprintf(stderr, "[k_alpha_mix<%d,%d,%d>] smem %zu blocks %d x %d waves, n_list %d, occupancy %d blocks/CU\n", \
                    PP, Q, (int)WG_, smem, blocks, nw, n_list, nb);                                                      \
        }                                                                                                               \
        hipLaunchKernelGGL((k_alpha_mix<PP, Q, WG_>), dim3(blocks), dim3(64 * nw), smem, st, ys, mu_s, D, cont_mask, list, \
                           n_list, n_dev, queue, alpha_hat, min_disp, max_disp, prior_var, prior_reg, alpha, conv, nfev, \
                           grid_count, grid_list, nll_const, const_mode, eval_cap, resume, (Lbfgsb1d*)park_state,       \
                           park_count, park_list);                                                                      \
    } while (0)
#define DSQ_MIX_LAUNCH(PP)                                                                                              \
    do {                                                                                                                \
        if constexpr (PP >= Q) {                                                                                        \
            if (wg) DSQ_MIX_LAUNCH_K(PP, true);                                                                         \
            else DSQ_MIX_LAUNCH_K(PP, false);                                                                           \
        }                                                                                                               \
    } while (0)
    switch (D.P) {
        case 1: DSQ_MIX_LAUNCH(1); break;
        case 2: DSQ_MIX_LAUNCH(2); break;
        case 3: DSQ_MIX_LAUNCH(3); break;
        case 4: DSQ_MIX_LAUNCH(4); break;
        case 5: DSQ_MIX_LAUNCH(5); break;
        case 6: DSQ_MIX_LAUNCH(6); break;
        case 7: DSQ_MIX_LAUNCH(7); break;
        case 8: DSQ_MIX_LAUNCH(8); break;
        default: return hipErrorInvalidValue;
    }
#undef DSQ_MIX_LAUNCH_K
#undef DSQ_MIX_LAUNCH
    return hipGetLastError();
}

}  // namespace dsq

#if defined(DSQ_MIX_PHASES) && DSQ_MIX_Q == 3
extern "C" int dsq_debug_mix_phase_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dsq::g_mix_phase), 12 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[12] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(dsq::g_mix_phase), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
