// dsq_k_irls_mix.hip — NB-GLM IRLS (utils.py:273-438 irls_solver) with the fused Cook's / Wald epilogue (dds.py:986-1040,
// 1066-1110; ds.py:303-360, utils.py:718-811) for MIXED designs: categorical columns with few distinct rows + up to three
// continuous covariates (dsq_mix.h; BASELINE configs[4]).  The companion of dsq_k_alpha_mix.hip.
//
// One gene per wavefront, persistent wavefronts with a device-side gene queue, samples walked in slot order (sorted by
// design cell).  Against k_irls<P, 0> (the general kernel these designs used to run):
//   * per sample and sweep 1 + Q + Q (Q + 1) / 2 + 1 + Q = 14 accumulations at Q = 3 (X^T W X and X^T W z from per-cell
//     sums, cell x covariate sums and a small covariate block) instead of 36 + 8, no design loads: the linear predictor
//     is eta_c + z . beta_z with eta_c looked up per loop iteration;
//   * counts staged once per gene in LDS (uint16) from their slot-ordered copy (round 5: a contiguous row, no gather
//     through the slot permutation; a gene with a count beyond 16 bits - flagged in `big` - gathers from its int32 row),
//     log size factors and covariates streamed from L2 with the next iteration's loads issued ahead of the arithmetic;
//   * start values from per-cell sums of log(y / sf + 0.1) and (X^T X)^-1 (the reference's QR solve, utils.py:349-353,
//     is the same least-squares solution);
//   * the mu-independent part of the deviance from per-gene tail counts (no lgamma per sample);
//   * hat diagonal h_n = w_n x_n^T (X^T W X + ridge)^-1 x_n as A_c + 2 b_c . z + z^T D z with per-cell A_c, b_c.
// Diverged genes (|beta| > max_beta or maxiter) go to the general rescue kernel through the fallback list, as k_irls does.
// Compiled once per number of continuous covariates (-DDSQ_MIX_Q=1|2|3).
#include <cstdio>
#include <type_traits>

#include "dsq_alpha_rows.h"
#include "dsq_irls.h"
#include "dsq_launch.h"
#include "dsq_mix.h"

#ifndef DSQ_MIX_Q
#error "compile with -DDSQ_MIX_Q=1, 2 or 3"
#endif

// the exponential of the sweeps (one per sample and sweep): the table-driven fexp_t (dsq_math.h, <= 1 ulp, ~22 instructions
// + one LDS read; its 1 KB table is filled per workgroup).  Round 5 measured it at +37 spilled registers inside the kernel
// that still held the epilogue and left the library's in; with the epilogue in a kernel of its own (round 6) the sweeps
// have the registers: c5 shard 7.80 -> 7.67 ms per step.  -DDSQ_MIX_LIBEXP selects the library's for the A/B.
#if defined(DSQ_MIX_LIBEXP)
#define DSQ_MIX_EXP(x) exp(x)
#else
#define DSQ_MIX_FEXP 1
#define DSQ_MIX_EXP(x) fexp_t(x)
#endif

namespace dsq {

constexpr int kMixPad = 0xFFFF;  // count stored for a padding slot (real counts of the genes listed for this family are <= 65533: k_count_big)

struct MixIrlsLds {  // wave-private LDS record (followed by the gene's counts, uint16 [Ns])
    double cellv[kMixMaxCells];                  // x_c . beta of the categorical part, per cell
    double ent[kMixMaxP * (kMixMaxP + 1) / 2 + kMixMaxP];  // X^T W X and X^T W z on their way to all lanes
    unsigned int hist[kMixTail];
    uint16_t tail[kMixTail];
};
static_assert(sizeof(MixIrlsLds) % 8 == 0, "the counts follow the record");

DSQ_HD size_t mixi_wave_bytes(int Ns) { return (sizeof(MixIrlsLds) + (size_t)Ns * 2 + 15) & ~(size_t)15; }
DSQ_HD size_t mixi_shared_bytes(int Ns, int P) {
    return (size_t)kMixMaxCells * P * 8 + (size_t)P * P * 8 + (size_t)(((Ns >> 6) + 15) & ~15);
}

// slot-ordered copies of the per-sample vectors of one fit: size factors (1 in padding slots), their logs, Cook's flags
template <int Q_>  // (a template only so that the three translation units do not define one symbol three times)
__global__ void k_mix_prep(const double* __restrict__ sf, const uint8_t* __restrict__ flags,
                           const int32_t* __restrict__ perm, int Ns, double* __restrict__ sfs,
                           double* __restrict__ lsfs, uint8_t* __restrict__ flags_s) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= Ns) return;
    const int p = perm[s];
    const double v = p >= 0 ? sf[p] : 1.0;
    sfs[s] = v;
    lsfs[s] = log(v);
    if (flags_s != nullptr) flags_s[s] = (p >= 0 && flags != nullptr) ? flags[p] : (uint8_t)0;
}

template <int P, int Q>
__global__ __launch_bounds__(256, 2) void k_irls_mix(
    const int32_t* __restrict__ y, int ldn, const uint16_t* __restrict__ ys, const uint8_t* __restrict__ ys_big,
    const MixDesign D, unsigned cont_mask, const double* __restrict__ sfs,
    const double* __restrict__ lsfs, const uint8_t* __restrict__ flags_s, int G, int32_t* __restrict__ queue,
    const double* __restrict__ disp, double min_mu, double beta_tol, double max_beta, int maxiter,
    double* __restrict__ beta_out, double* __restrict__ m_out,
    uint8_t* __restrict__ conv, int32_t* __restrict__ iters, int32_t* __restrict__ fb_count,
    int32_t* __restrict__ fb_list, const uint8_t* __restrict__ part, int part_want) {
    constexpr int T = Tri<P>::N;
    constexpr int QQ = Q * (Q + 1) / 2;
    constexpr int U = kMixU;
    static_assert(Q >= 1 && Q <= kMixMaxQ && P >= Q && P <= kMixMaxP, "design shape");
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int Ns = D.Ns, ntrips = Ns >> 6, C = D.C, N = D.N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* const xc_s = dyn;                                      // [C][P] (continuous columns 0)
    double* const gi_s = xc_s + kMixMaxCells * P;                  // [P][P] (X^T X)^-1
    uint8_t* const tc_s = (uint8_t*)(gi_s + P * P);                // [ntrips]
    char* const wbase = (char*)dyn + mixi_shared_bytes(Ns, P) + mixi_wave_bytes(Ns) * (size_t)w;
    MixIrlsLds* const L = (MixIrlsLds*)wbase;
    uint16_t* const y16 = (uint16_t*)(wbase + sizeof(MixIrlsLds));

    log_tab_fill();
#if defined(DSQ_MIX_FEXP)
    exp_tab_fill();
#endif
    for (int i = threadIdx.x; i < C * P; i += blockDim.x) xc_s[i] = D.Xc[i];
    for (int i = threadIdx.x; i < P * P; i += blockDim.x) gi_s[i] = D.Ginv[i];
    for (int i = threadIdx.x; i < ntrips; i += blockDim.x) tc_s[i] = D.trip_cell[i];
    __syncthreads();

    // lane e owns entry e = tri(ei, ej) of X^T W X (see k_alpha_mix); lane j < P also owns entry j of X^T W z
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
    const int rj = lane < P ? lane : P - 1;                              // this lane's entry of a P-vector
    const bool rj_cont = ((cont_mask >> rj) & 1u) != 0;
    const int rj_q = __popc(cont_mask & ((1u << rj) - 1u));
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
    constexpr int q1 = Q > 1 ? 1 : 0, q2 = Q > 2 ? 2 : 0;
    constexpr int L_ = QQ - 1;
    // b[col] for a run-time column index without dynamic register indexing (which would put the array into scratch)
    auto at_col = [](const double (&b)[P], int col) {
        double v = b[0];
#pragma unroll
        for (int j = 1; j < P; ++j) v = (j == col) ? b[j] : v;
        return v;
    };
    const double lmin = log(min_mu);

    for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(queue, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= G) break;
        // (a fit in two launches, IrlsExtras::part: the other launch's genes are left alone)
        if (part != nullptr && __builtin_amdgcn_readfirstlane((int)part[g]) != part_want) continue;
        const double dsp = DeviceWave::uniform(disp[g]);
        const double a = DeviceWave::uniform(1.0 / dsp);
        const int32_t* const yg = y + (size_t)g * ldn;
        const uint16_t* const ysg = ys + (size_t)g * Ns;
        // a count beyond the 16-bit staging: staging, sweeps and epilogue gather this gene's counts from its int32 row
#if defined(DSQ_IRLS_MIX_GATHER)  // developer A/B: staging through the permutation gather, as before round 5
        bool big_gene = false;
        constexpr bool kGatherAlways = true;
#else
        const bool big_gene = __builtin_amdgcn_readfirstlane((int)ys_big[g]) != 0;
        constexpr bool kGatherAlways = false;
#endif
        // ---------------------------------------------------------------- stage the counts; start values; deviance constant
        for (int i = lane; i < kMixTail; i += 64) L->hist[i] = 0u;
        DeviceWave::sync();
        double beta[P];
        double cst;
        {
            double rv = 0.0, sly = 0.0, zl[Q];  // X^T log(y / sf + 0.1): entry rj | the current cell's sum | covariates
#pragma unroll
            for (int q = 0; q < Q; ++q) zl[q] = 0.0;
            int nbig = 0, maxc = 0;
            double cbig = 0.0;
            int cur = __builtin_amdgcn_readfirstlane((int)tc_s[0]);
            auto fold0 = [&](int c) {
                const double S = DeviceWave::sum(sly);
                rv += xc_s[c * P + rj] * S;
                sly = 0.0;
            };
            // one loop iteration = U trips of one design cell: valid[c] / v[c] the slot's count, sv its size factor, zq its covariates
            auto body = [&](int base, const bool (&vl)[U], const int (&v2)[U], const double (&sv)[U], const double (&zq)[U][Q]) {
                const int cell = __builtin_amdgcn_readfirstlane((int)tc_s[base >> 6]);
                if (cell != cur) {
                    fold0(cur);
                    cur = cell;
                }
#pragma unroll
                for (int c = 0; c < U; ++c) {
                    const int s = base + 64 * c + lane;
                    const bool valid = vl[c];
                    const int v = valid ? v2[c] : 0;
                    y16[s] = (uint16_t)(valid ? v : kMixPad);
                    maxc = v > maxc ? v : maxc;
                    const bool isbig = v >= kMixTail;
                    if (v > 0 && !isbig) atomicAdd(&L->hist[v], 1u);
                    nbig += __popcll(__ballot(isbig));
                    if (isbig) {  // [lgamma(y + a) - lgamma(T + a)] - [lgamma(y + 1) - lgamma(T + 1)] beyond the tail table
                        double l1, p1, l2, p2, l3, p3, l4, p4;
                        stirling_big((double)v + a, l1, p1);
                        stirling_big((double)kMixTail + a, l2, p2);
                        stirling_big((double)v + 1.0, l3, p3);
                        stirling_big((double)kMixTail + 1.0, l4, p4);
                        cbig += (l1 - l2) - (l3 - l4);
                    }
                    const double ly = valid ? flog_t((double)v * frcp(sv[c]) + 0.1) : 0.0;
                    sly += ly;
#pragma unroll
                    for (int q = 0; q < Q; ++q) zl[q] = fma(zq[c][q], ly, zl[q]);
                }
            };
            if (kGatherAlways || big_gene) {  // (wave-uniform, rare: a count beyond 16 bits) the int32 row through the permutation
                for (int base = 0; base < Ns; base += 64 * U) {
                    int pp[U], v2[U];
                    bool vl[U];
                    double sv[U], zq[U][Q];
#pragma unroll
                    for (int c = 0; c < U; ++c) pp[c] = D.perm[base + 64 * c + lane];
#pragma unroll
                    for (int c = 0; c < U; ++c) {
                        const int s = base + 64 * c + lane;
                        v2[c] = yg[pp[c] >= 0 ? pp[c] : 0];
                        vl[c] = pp[c] >= 0;
                        sv[c] = sfs[s];
#pragma unroll
                        for (int q = 0; q < Q; ++q) zq[c][q] = D.Zs[(size_t)q * Ns + s];
                    }
                    body(base, vl, v2, sv, zq);
                }
            } else {
                // the slot-ordered uint16 row: contiguous, the next iteration's loads issued ahead of this one's arithmetic
                int vn[U];
                double sn[U], zn0[U][Q];
                auto issue0 = [&](int base) {
#pragma unroll
                    for (int c = 0; c < U; ++c) {
                        const int s = base + 64 * c + lane;
                        vn[c] = ysg[s];
                        sn[c] = sfs[s];
#pragma unroll
                        for (int q = 0; q < Q; ++q) zn0[c][q] = D.Zs[(size_t)q * Ns + s];
                    }
                };
                issue0(0);
                for (int base = 0; base < Ns; base += 64 * U) {
                    int v2[U];
                    bool vl[U];
                    double sv[U], zq[U][Q];
#pragma unroll
                    for (int c = 0; c < U; ++c) {
                        v2[c] = vn[c];
                        vl[c] = vn[c] != kMixPad;
                        sv[c] = sn[c];
#pragma unroll
                        for (int q = 0; q < Q; ++q) zq[c][q] = zn0[c][q];
                    }
                    issue0(base + 64 * U < Ns ? base + 64 * U : base);  // (the last iteration re-reads its own slots: no branch)
                    body(base, vl, v2, sv, zq);
                }
            }
            fold0(cur);
            DeviceWave::template sum_n<Q>(zl);
            if (rj_cont) rv = pick3(rj_q, zl[0], zl[q1], zl[q2]);
            maxc = DeviceWave::maxi(maxc);
#if defined(DSQ_IRLS_MIX_GATHER)
            big_gene = maxc >= kMixPad;
#endif
            if (lane < P) L->ent[lane] = rv;
            DeviceWave::sync();
#pragma unroll
            for (int j = 0; j < P; ++j) {
                double b = 0.0;
#pragma unroll
                for (int k = 0; k < P; ++k) b += gi_s[j * P + k] * L->ent[k];
                beta[j] = DeviceWave::uniform(b);
            }
            // sum_n [lgamma(y + a) - lgamma(a) - log y!] = sum_i T_i [log(a + i) - log(i + 1)] over the tail counts
            // T_i = #{y > i} (+ the Stirling differences of the counts beyond the table): the mu-independent part of the
            // NB log-likelihood; it only enters the denominator |dev| + 0.1 of the stopping test (utils.py:418-421)
            constexpr int BPL = kMixTail / 64;
            int h[BPL], tot = 0;
#pragma unroll
            for (int i = 0; i < BPL; ++i) { h[i] = (int)L->hist[lane * BPL + i]; tot += h[i]; }
            const int below = DeviceWave::excl_scan_i(tot);
            const int all = DeviceWave::sumi(tot);
            int above = all - below - tot + nbig;
            double c1 = cbig;
            const int ntl = maxc < kMixTail ? maxc : kMixTail;
#pragma unroll
            for (int i = BPL - 1; i >= 0; --i) {
                const int idx = lane * BPL + i;
                if (above > 0 && idx < ntl) c1 += (double)above * (flog_t(a + (double)idx) - flog_t((double)(idx + 1)));
                above += h[i];
            }
            cst = DeviceWave::uniform(DeviceWave::sum(c1));
            DeviceWave::sync();
        }
        const double nlogterm = (double)N * a * log(dsp);

        // ---------------------------------------------------------------- one sweep: S, M = X^T W X, r = X^T W z at beta
        double M[T], r[P], S = 0.0;
        auto sweep = [&]() {
            if (lane < kMixMaxCells) {
                double e = 0.0;
                if (lane < C) {
#pragma unroll
                    for (int j = 0; j < P; ++j) e += xc_s[lane * P + j] * beta[j];
                }
                L->cellv[lane] = e;
            }
            double bz[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) bz[q] = at_col(beta, D.zcol[q]);
            DeviceWave::sync();
            double sc[2 + Q], zz[QQ], zr[Q], Me = 0.0, re = 0.0, sdev = 0.0;  // per-cell: w, w z_q, w zwork
#pragma unroll
            for (int i = 0; i < 2 + Q; ++i) sc[i] = 0.0;
#pragma unroll
            for (int i = 0; i < QQ; ++i) zz[i] = 0.0;
#pragma unroll
            for (int q = 0; q < Q; ++q) zr[q] = 0.0;
            auto fold = [&](int c) {
                DeviceWave::template sum_n<2 + Q>(sc);
                const double va = xc_s[c * P + xa], vb = xc_s[c * P + xb];
                const double s1 = pick3(qz, sc[1], sc[1 + q1], sc[1 + q2]);
                Me += kind == 0 ? (va * vb) * sc[0] : (kind == 1 ? va * s1 : 0.0);
                re += xc_s[c * P + rj] * sc[1 + Q];
#pragma unroll
                for (int i = 0; i < 2 + Q; ++i) sc[i] = 0.0;
            };
            int cur = __builtin_amdgcn_readfirstlane((int)tc_s[0]);
            double etac = DeviceWave::uniform(L->cellv[cur]);
            int yn[U];
            double ln[U], zn[U][Q];
            auto issue = [&](int t0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s = (t0 + u) * 64 + lane;
                    if (big_gene) {
                        const int pq = D.perm[s];
                        yn[u] = pq >= 0 ? yg[pq] : -1;
                    } else {
                        const int v = y16[s];
                        yn[u] = v == kMixPad ? -1 : v;
                    }
                    ln[u] = lsfs[s];
#pragma unroll
                    for (int q = 0; q < Q; ++q) zn[u][q] = D.Zs[(size_t)q * Ns + s];
                }
            };
            issue(0);
            for (int t0 = 0; t0 < ntrips; t0 += U) {
                int yi[U];
                double lsf[U], z[U][Q];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    yi[u] = yn[u];
                    lsf[u] = ln[u];
#pragma unroll
                    for (int q = 0; q < Q; ++q) z[u][q] = zn[u][q];
                }
                issue(t0 + U < ntrips ? t0 + U : t0);
                const int cell = __builtin_amdgcn_readfirstlane((int)tc_s[t0]);
                if (cell != cur) {
                    fold(cur);
                    cur = cell;
                    etac = DeviceWave::uniform(L->cellv[cur]);
                }
                double eta0[U], e[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    double t = etac;
#pragma unroll
                    for (int q = 0; q < Q; ++q) t = fma(z[u][q], bz[q], t);
                    eta0[u] = t;
                    e[u] = DSQ_MIX_EXP(t + lsf[u]);  // mu = sf exp(eta) = exp(eta + log sf)
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool valid = yi[u] >= 0;
                    const double yv = valid ? (double)yi[u] : 0.0;
                    const bool clamped = !(e[u] > min_mu);
                    const double mu = clamped ? min_mu : e[u];
                    const double lmu = clamped ? lmin : eta0[u] + lsf[u];
                    const double sd = (yv + a) * flog_t(a + mu) - yv * lmu;
                    const double rd = frcp(1.0 + mu * dsp);
                    double wv = mu * rd;
                    double wz = (mu * (clamped ? lmin - lsf[u] : eta0[u]) + (yv - mu)) * rd;
                    wv = valid ? wv : 0.0;
                    wz = valid ? wz : 0.0;
                    sdev += valid ? sd : 0.0;
                    sc[0] += wv;
                    sc[1 + Q] += wz;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const double wq = wv * z[u][q];
                        sc[1 + q] += wq;
                        zr[q] = fma(wz, z[u][q], zr[q]);
#pragma unroll
                        for (int q2 = 0; q2 <= q; ++q2) zz[tri(q, q2)] = fma(wq, z[u][q2], zz[tri(q, q2)]);
                    }
                }
            }
            fold(cur);
            DeviceWave::template sum_n<QQ>(zz);
            DeviceWave::template sum_n<Q>(zr);
            if (kind == 2)
                Me = pick6(zzk, zz[0], zz[1 < L_ ? 1 : L_], zz[2 < L_ ? 2 : L_], zz[3 < L_ ? 3 : L_], zz[4 < L_ ? 4 : L_], zz[L_]);
            if (rj_cont) re = pick3(rj_q, zr[0], zr[q1], zr[q2]);
            S = DeviceWave::sum(sdev);
            if (lane < T) L->ent[lane] = Me;
            if (lane < P) L->ent[T + lane] = re;
            DeviceWave::sync();
#pragma unroll
            for (int i = 0; i < T; ++i) M[i] = L->ent[i];
#pragma unroll
            for (int j = 0; j < P; ++j) r[j] = L->ent[T + j];
            DeviceWave::sync();
        };

        // ---------------------------------------------------------------- the IRLS loop (utils.py:360-421)
        sweep();
        double dev = 1000.0, ratio = 1.0;
        int it = 0;
        bool fallback = false;
        while (ratio > beta_tol) {
            double Hm[T], bh[P];
#pragma unroll
            for (int i = 0; i < T; ++i) Hm[i] = M[i];
#pragma unroll
            for (int j = 0; j < P; ++j) { Hm[tri(j, j)] += 1e-6; bh[j] = r[j]; }
            chol<P>(Hm);
            chol_solve<P>(Hm, bh);
            it += 1;
            bool bad = it >= maxiter;
#pragma unroll
            for (int j = 0; j < P; ++j) bad = bad || (fabs(bh[j]) > max_beta);  // NaN is not "bad" (as in the reference)
            if (bad) { fallback = true; break; }
#pragma unroll
            for (int j = 0; j < P; ++j) beta[j] = bh[j];
            sweep();
            const double old = dev;
            dev = -2.0 * (nlogterm - cst + S);
            ratio = fabs(dev - old) / (fabs(dev) + 0.1);
        }
        if (fallback) {
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < P; ++j) beta_out[(size_t)g * P + j] = beta[j];
                conv[g] = 0;
                if (iters != nullptr) iters[g] = it;
                fb_list[atomicAdd(fb_count, 1)] = g;
            }
            DeviceWave::sync();
            continue;
        }

        // ---------------------------------------------------------------- finish
        // Round 6: mu / hat diagonal / Cook's / Wald run in a kernel of their own (k_mix_epilogue below).  Inside this kernel
        // that pass was 0.7 ms of the 2.1 ms LFC launch at 7500 x 5000 - four sweeps' time for one sweep's arithmetic: the
        // register allocation of this kernel peaks in the sweeps and the p x p algebra between them (256 VGPRs, 170 spilled),
        // and the values spilled for that peak were reloaded from scratch on every trip of the epilogue's sample loop (8
        // scratch loads per 64 samples, two wavefronts per SIMD to hide them behind).  The IRLS kernel now hands over the
        // coefficients and the last sweep's X^T W X (36 doubles per gene).
        if (lane < T && m_out != nullptr) m_out[(size_t)g * T + lane] = L->ent[lane];  // (the last sweep left it there)
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < P; ++j) beta_out[(size_t)g * P + j] = beta[j];
            conv[g] = 1;
            if (iters != nullptr) iters[g] = it;
        }
        DeviceWave::sync();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The per-sample half of what follows a converged fit: mu = sf exp(X beta) (utils.py:435-437), the hat diagonal
// h_n = w_n x_n^T (X^T W X + ridge)^-1 x_n (utils.py:427-433), Cook's distances with their per-gene flags (dds.py:986-1040,
// 1066-1110) and the Wald statistics (utils.py:718-811) - one gene per wavefront, samples in slot order, from the
// coefficients and the last sweep's X^T W X that k_irls_mix left behind.  A kernel of its own (round 6): nothing of the
// IRLS kernel's register peak lives here (no spill in the sample loop), four wavefronts per SIMD hide its loads.
struct MixEpiLds {  // wave-private LDS record
    double cellv[kMixMaxCells];                // x_c . beta of the categorical part, per cell
    double cellq[kMixMaxCells][1 + kMixMaxQ];  // A_c = x_c^T H x_c and b_c = (H x_c) at the covariates' columns
    double ent[kMixMaxP * (kMixMaxP + 1) / 2];  // Wald's X^T W X on its way to all lanes
};

template <int P, int Q>
__global__ __launch_bounds__(256, 3) void k_mix_epilogue(
    const int32_t* __restrict__ y, int ldn, const uint16_t* __restrict__ ys, const uint8_t* __restrict__ ys_big,
    const MixDesign D, unsigned cont_mask, const double* __restrict__ lsfs, const uint8_t* __restrict__ flags_s, int G,
    const double* __restrict__ disp, double min_mu, const double* __restrict__ beta_in, const double* __restrict__ m_in,
    const uint8_t* __restrict__ conv, double* __restrict__ mu_out, double* __restrict__ hat_out, IrlsExtras ex) {
    constexpr int T = Tri<P>::N;
    constexpr int QQ = Q * (Q + 1) / 2;
    constexpr int U = kMixU;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int Ns = D.Ns, ntrips = Ns >> 6, C = D.C, N = D.N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* const xc_s = dyn;                                   // [C][P]
    uint8_t* const tc_s = (uint8_t*)(xc_s + kMixMaxCells * P);  // [ntrips]
    MixEpiLds* const L = (MixEpiLds*)((char*)dyn + (size_t)kMixMaxCells * P * 8 + (size_t)(((Ns >> 6) + 15) & ~15)) + w;
#if defined(DSQ_MIX_FEXP)
    exp_tab_fill();
#endif
    for (int i = threadIdx.x; i < C * P; i += blockDim.x) xc_s[i] = D.Xc[i];
    for (int i = threadIdx.x; i < ntrips; i += blockDim.x) tc_s[i] = D.trip_cell[i];
    __syncthreads();
    const int g = blockIdx.x * 4 + w;
    if (g >= G) return;
    if (ex.part != nullptr && __builtin_amdgcn_readfirstlane((int)ex.part[g]) != ex.part_want) return;  // the other launch's
    if (__builtin_amdgcn_readfirstlane((int)conv[g]) == 0) return;  // diverged: the rescue kernel writes this gene's outputs

    // lane e owns entry e = tri(ei, ej) of X^T W X (as in k_irls_mix)
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
    constexpr int q1 = Q > 1 ? 1 : 0, q2 = Q > 2 ? 2 : 0;
    constexpr int L_ = QQ - 1;

    const double dsp = DeviceWave::uniform(disp[g]);
    const int32_t* const yg = y + (size_t)g * ldn;
    const uint16_t* const ysg = ys + (size_t)g * Ns;
    const bool big_gene = __builtin_amdgcn_readfirstlane((int)ys_big[g]) != 0;
    double beta[P];
#pragma unroll
    for (int j = 0; j < P; ++j) beta[j] = DeviceWave::uniform(beta_in[(size_t)g * P + j]);

    const bool want_cooks = ex.flags != nullptr, want_wald = ex.ridge != nullptr;
    double* const mu_row = mu_out != nullptr ? mu_out + (size_t)g * ldn : nullptr;
    double* const hat_row = hat_out != nullptr ? hat_out + (size_t)g * ldn : nullptr;
    // the Cook's layer: ex.cooks_ld == 0: sample order, pitch ldn; else SLOT order with that pitch (>= Ns) - one coalesced
    // 512-byte store per trip instead of 64 scattered 8-byte stores; the readers (outlier replacement,
    // DeseqPipeline.layer) go through MixDesign::slot_of
    const bool cooks_slots = ex.cooks_ld > 0;
    double* const cooks_row = (want_cooks && ex.cooks != nullptr)
                                  ? ex.cooks + (size_t)g * (cooks_slots ? (size_t)ex.cooks_ld : (size_t)ldn)
                                  : nullptr;
    const bool have_w = hat_row != nullptr || want_cooks;
    double Dq[QQ];  // H at the covariates' columns
#pragma unroll
    for (int i = 0; i < QQ; ++i) Dq[i] = 0.0;
    if (have_w) {
        double M[T], inv[T];
#pragma unroll
        for (int i = 0; i < T; ++i) M[i] = DeviceWave::uniform(m_in[(size_t)g * T + i]);
#pragma unroll
        for (int j = 0; j < P; ++j) M[tri(j, j)] += 1e-6;
        chol<P>(M);
        chol_inverse<P>(M, inv);
#pragma unroll
        for (int qa = 0; qa < Q; ++qa)
#pragma unroll
            for (int qb = 0; qb <= qa; ++qb) {
                double v = 0.0;  // (dynamic column index into a register array: selected entry by entry)
#pragma unroll
                for (int i = 0; i < P; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j)
                        v = ((i == D.zcol[qa] && j == D.zcol[qb]) || (i == D.zcol[qb] && j == D.zcol[qa])) ? inv[tri(i, j)] : v;
                Dq[tri(qa, qb)] = DeviceWave::uniform(v);
            }
        if (lane < kMixMaxCells) {
            const int c = lane < C ? lane : 0;
            double x[P], hx[P];
#pragma unroll
            for (int j = 0; j < P; ++j) x[j] = xc_s[c * P + j];
            sym_matvec<P>(inv, x, hx);
            double A = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) A += x[j] * hx[j];
            L->cellq[lane][0] = A;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < P; ++j) v = (j == D.zcol[q]) ? hx[j] : v;
                L->cellq[lane][1 + q] = v;
            }
        }
    }
    if (lane < kMixMaxCells) {  // eta_c at the final beta
        double e = 0.0;
        if (lane < C) {
#pragma unroll
            for (int j = 0; j < P; ++j) e += xc_s[lane * P + j] * beta[j];
        }
        L->cellv[lane] = e;
    }
    double bz[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        double v = beta[0];
#pragma unroll
        for (int j = 1; j < P; ++j) v = (j == D.zcol[q]) ? beta[j] : v;
        bz[q] = v;
    }
    DeviceWave::sync();
    CooksAcc<DeviceWave> acc(want_cooks ? ex.robust_disp[g] : 0.0, want_cooks ? ex.cutoff : 0.0, P);
    double sc[1 + Q], zz[QQ], Me = 0.0;  // Wald: X^T W X at the UNclamped mu (ds.py:320-324)
#pragma unroll
    for (int i = 0; i < 1 + Q; ++i) sc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < QQ; ++i) zz[i] = 0.0;
    auto fold = [&](int c) {
        DeviceWave::template sum_n<1 + Q>(sc);
        const double va = xc_s[c * P + xa], vb = xc_s[c * P + xb];
        const double s1 = pick3(qz, sc[1], sc[1 + q1], sc[1 + q2]);
        Me += kind == 0 ? (va * vb) * sc[0] : (kind == 1 ? va * s1 : 0.0);
#pragma unroll
        for (int i = 0; i < 1 + Q; ++i) sc[i] = 0.0;
    };
    int cur = -1;
    double etac = 0.0, Ac = 0.0, bc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) bc[q] = 0.0;
    // the loads of the next iteration are issued ahead of this one's arithmetic
    int pn_[U], fn_[U], yn_[U];
    double ln_[U], zn_[U][Q];
    auto issue_e = [&](int t0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = (t0 + u) * 64 + lane;
            pn_[u] = D.perm[s];
            yn_[u] = (int)ysg[s];
            ln_[u] = lsfs[s];
            fn_[u] = (int)flags_s[s];
#pragma unroll
            for (int q = 0; q < Q; ++q) zn_[u][q] = D.Zs[(size_t)q * Ns + s];
        }
    };
    issue_e(0);
    for (int t0 = 0; t0 < ntrips; t0 += U) {
        int pq_[U], fl_[U], yv_[U];
        double lsf_[U], z_[U][Q];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pq_[u] = pn_[u];
            fl_[u] = fn_[u];
            yv_[u] = yn_[u];
            lsf_[u] = ln_[u];
#pragma unroll
            for (int q = 0; q < Q; ++q) z_[u][q] = zn_[u][q];
        }
        issue_e(t0 + U < ntrips ? t0 + U : t0);  // (the last iteration re-reads its own slots: no branch)
        const int cell = __builtin_amdgcn_readfirstlane((int)tc_s[t0]);
        if (cell != cur) {
            if (want_wald && cur >= 0) fold(cur);
            cur = cell;
            etac = DeviceWave::uniform(L->cellv[cur]);
            Ac = DeviceWave::uniform(L->cellq[cur][0]);
#pragma unroll
            for (int q = 0; q < Q; ++q) bc[q] = DeviceWave::uniform(L->cellq[cur][1 + q]);
        }
        double mu_raw_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double t = etac;
#pragma unroll
            for (int q = 0; q < Q; ++q) t = fma(z_[u][q], bz[q], t);
            mu_raw_[u] = DSQ_MIX_EXP(t + lsf_[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = (t0 + u) * 64 + lane;
            const int pq = pq_[u];
            const bool valid = pq >= 0;
            const int n = valid ? pq : 0;
            const int yi = valid ? (big_gene ? yg[n] : yv_[u]) : 0;
            const double mu_raw = mu_raw_[u];
            const double(&z)[Q] = z_[u];
            if (valid && mu_row != nullptr) mu_row[n] = mu_raw;
            double wv = 0.0;
            if (have_w) {
                const double mu = dmax(mu_raw, min_mu);
                wv = mu * frcp_g(1.0 + mu * dsp);
                double qf = Ac;  // x_n^T H x_n = A_c + 2 b_c . z + z^T D z
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    double dz = 0.0;
#pragma unroll
                    for (int qq = 0; qq < Q; ++qq) dz = fma(Dq[tris(q, qq)], z[qq], dz);
                    qf = fma(z[q], 2.0 * bc[q] + dz, qf);
                }
                const double h = wv * qf;
                if (valid && hat_row != nullptr) hat_row[n] = h;
                if (want_cooks) {
                    double ck = 0.0;
                    if (valid) ck = acc.add(n, (double)yi, mu_raw, h, fl_[u]);
                    if (cooks_row != nullptr) {
                        if (cooks_slots) cooks_row[s] = ck;
                        else if (valid) cooks_row[n] = ck;
                    }
                }
            }
            if (want_wald) {
                double wu = wv;  // the same number unless a lane was clamped
                if (!have_w || DeviceWave::any(!(mu_raw >= min_mu))) wu = mu_raw * frcp_g(1.0 + mu_raw * dsp);
                wu = valid ? wu : 0.0;
                sc[0] += wu;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const double wq = wu * z[q];
                    sc[1 + q] += wq;
#pragma unroll
                    for (int qq = 0; qq <= q; ++qq) zz[tri(q, qq)] = fma(wq, z[qq], zz[tri(q, qq)]);
                }
            }
        }
    }
    CooksOut cko{};
    WaldOut wo{};
    if (want_cooks) {
        // "fewer than three samples above the one with the largest Cook's distance" (dds.py:1094-1101): counted over the
        // gene's slot-ordered uint16 row (any order will do; it is in L2 from the pass above)
        if (big_gene) {
            cko = acc.finish(yg, N);
        } else {
            cko = acc.finish_counted(N, [&](int yref) {
                int above = 0;
#pragma unroll 4
                for (int s = lane; s < Ns; s += 64) {
                    const int v = ysg[s];
                    above += (v != kMixPad && v > yref) ? 1 : 0;
                }
                return above;
            });
        }
    }
    if (want_wald) {
        fold(cur);
        DeviceWave::template sum_n<QQ>(zz);
        if (kind == 2)
            Me = pick6(zzk, zz[0], zz[1 < L_ ? 1 : L_], zz[2 < L_ ? 2 : L_], zz[3 < L_ ? 3 : L_], zz[4 < L_ ? 4 : L_], zz[L_]);
        if (lane < T) L->ent[lane] = Me;
        DeviceWave::sync();
        double Mw[T];
#pragma unroll
        for (int i = 0; i < T; ++i) Mw[i] = L->ent[i];
        wo = wald_from_M<P>(Mw, beta, ex.ridge, ex.contrast, ex.lfc_null, ex.alt);
    }
    if (lane == 0) {
        if (want_cooks) {
            ex.any_all[g] = (uint8_t)cko.any_gt_all;
            ex.any_use[g] = (uint8_t)cko.any_gt_use;
            ex.any_use_nr[g] = (uint8_t)cko.any_gt_use_nr;
            ex.few_above[g] = (uint8_t)cko.few_above;
        }
        if (want_wald) { ex.pvals[g] = wo.p; ex.stats[g] = wo.stat; ex.se[g] = wo.se; }
    }
}

DSQ_HD size_t mix_epi_shared_bytes(int Ns, int P) {
    return (size_t)kMixMaxCells * P * 8 + (size_t)(((Ns >> 6) + 15) & ~15) + 4 * sizeof(MixEpiLds);
}

static int mixi_waves_per_block(int Ns, int P) {
    for (int nw : {4, 2, 1}) {
        const size_t smem = mixi_shared_bytes(Ns, P) + mixi_wave_bytes(Ns) * nw + 64;
        if (2 * smem <= 156 * 1024) return nw;
    }
    return mixi_shared_bytes(Ns, P) + mixi_wave_bytes(Ns) + 64 <= 64 * 1024 ? 1 : 0;
}

#define DSQ_MIX_CAT_(a, b) a##b
#define DSQ_MIX_CAT(a, b) DSQ_MIX_CAT_(a, b)

// persistent grid for G genes: (workgroups, wavefronts per workgroup); 0 workgroups: rows too long
void DSQ_MIX_CAT(irls_mix_grid_q, DSQ_MIX_Q)(int Ns, int P, int G, int* blocks, int* nw_out) {
    *blocks = 0; *nw_out = 0;
    const int nw = mixi_waves_per_block(Ns, P);
    const int n_cu = current_device_cus();
    if (nw == 0 || n_cu <= 0 || G <= 0) return;
    const size_t smem = mixi_shared_bytes(Ns, P) + mixi_wave_bytes(Ns) * nw + 64;
    int per_cu = (int)((156 * 1024) / smem);
    if (per_cu * nw > 8) per_cu = 8 / nw;
    if (per_cu < 1) per_cu = 1;
    int b = (G + nw - 1) / nw;
    if (b > per_cu * n_cu) b = per_cu * n_cu;
    *blocks = b; *nw_out = nw;
}

// work (irls_mix_work_bytes): slot-ordered size factors, their logs, Cook's flags
hipError_t DSQ_MIX_CAT(launch_irls_mix_q, DSQ_MIX_Q)(
    hipStream_t st, const int32_t* y, int ldn, const uint16_t* ys, const uint8_t* ys_big, const MixDesign& D,
    const double* sf, int G, int32_t* queue,
    const double* disp, double min_mu, double beta_tol, double max_beta, int maxiter, double* beta, double* mu,
    double* hat, uint8_t* conv, int32_t* iters, int32_t* fb_count, int32_t* fb_list, const IrlsExtras& ex, void* work,
    size_t work_bytes) {
    constexpr int Q = DSQ_MIX_Q;
    if (G <= 0) return hipSuccess;
    if (D.Q != Q || D.P < Q || D.P > kMixMaxP || D.Ginv == nullptr || work == nullptr || ys == nullptr || ys_big == nullptr)
        return hipErrorInvalidValue;
    if (ex.cooks_ld != 0 && ex.cooks_ld < D.Ns) return hipErrorInvalidValue;
    int blocks = 0, nw = 0;
    DSQ_MIX_CAT(irls_mix_grid_q, DSQ_MIX_Q)(D.Ns, D.P, G, &blocks, &nw);
    const bool epilogue = mu != nullptr || hat != nullptr || ex.flags != nullptr || ex.ridge != nullptr;
    const size_t m_off = ((size_t)D.Ns * 17 + 63) & ~(size_t)63;  // (after the three slot-ordered vectors)
    if (blocks == 0 || m_off + (epilogue ? (size_t)G * Tri<kMixMaxP>::N * sizeof(double) : 0) > work_bytes)
        return hipErrorInvalidValue;
    const size_t smem = mixi_shared_bytes(D.Ns, D.P) + mixi_wave_bytes(D.Ns) * nw + 64;
    double* sfs = (double*)work;
    double* lsfs = sfs + D.Ns;
    uint8_t* flags_s = (uint8_t*)(lsfs + D.Ns);
    double* m_buf = epilogue ? (double*)((char*)work + m_off) : nullptr;  // the last sweep's X^T W X, [G][T]
    if (!(ex.part != nullptr && ex.part_shared_ready))  // (the partner launch of a two-launch fit may still be reading them)
        hipLaunchKernelGGL(k_mix_prep<Q>, dim3((D.Ns + 255) / 256), dim3(256), 0, st, sf, ex.flags, D.perm, D.Ns, sfs, lsfs,
                           flags_s);
    {
        const hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
    }
    unsigned cont_mask = 0;
    for (int q = 0; q < Q; ++q) cont_mask |= 1u << D.zcol[q];
#define DSQ_MIXI_LAUNCH(PP)                                                                                             \
    do {                                                                                                                \
        if constexpr (PP >= Q) {                                                                                        \
            if (smem > 48 * 1024) {                                                                                     \
                (void)hipFuncSetAttribute((const void*)k_irls_mix<PP, Q>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                          (int)smem);                                                                   \
                (void)hipGetLastError();                                                                                \
            }                                                                                                           \
            if (getenv("DSQ_DEBUG_ROWS")) {                                                                             \
                int nb = -1;                                                                                            \
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_irls_mix<PP, Q>, 64 * nw, smem); \
                fprintf(stderr, "[k_irls_mix<%d,%d>] smem %zu blocks %d x %d waves, G %d, occupancy %d blocks/CU\n", PP, \
                        Q, smem, blocks, nw, G, nb);                                                                    \
            }                                                                                                           \
            hipLaunchKernelGGL((k_irls_mix<PP, Q>), dim3(blocks), dim3(64 * nw), smem, st, y, ldn, ys, ys_big, D,      \
                               cont_mask, sfs, lsfs, flags_s, G, queue, disp, min_mu, beta_tol, max_beta, maxiter, beta, \
                               m_buf, conv, iters, fb_count, fb_list, ex.part, ex.part_want);                           \
            if (epilogue) {                                                                                             \
                const hipError_t e1 = hipGetLastError();                                                                \
                if (e1 != hipSuccess) return e1;                                                                        \
                hipLaunchKernelGGL((k_mix_epilogue<PP, Q>), dim3((G + 3) / 4), dim3(256),                               \
                                   mix_epi_shared_bytes(D.Ns, PP), st, y, ldn, ys, ys_big, D, cont_mask, lsfs, flags_s, \
                                   G, disp, min_mu, beta, m_buf, conv, mu, hat, ex);                                    \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
    switch (D.P) {
        case 1: DSQ_MIXI_LAUNCH(1); break;
        case 2: DSQ_MIXI_LAUNCH(2); break;
        case 3: DSQ_MIXI_LAUNCH(3); break;
        case 4: DSQ_MIXI_LAUNCH(4); break;
        case 5: DSQ_MIXI_LAUNCH(5); break;
        case 6: DSQ_MIXI_LAUNCH(6); break;
        case 7: DSQ_MIXI_LAUNCH(7); break;
        case 8: DSQ_MIXI_LAUNCH(8); break;
        default: return hipErrorInvalidValue;
    }
#undef DSQ_MIXI_LAUNCH
    return hipGetLastError();
}

bool DSQ_MIX_CAT(irls_mix_fits_q, DSQ_MIX_Q)(int Ns, int P) { return mixi_waves_per_block(Ns, P) > 0; }

}  // namespace dsq
