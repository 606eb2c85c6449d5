// dsq_trend.h — parametric dispersion trend  disp ~ a0 + a1/mean  fitted on the device.
//
// Replaces DeseqDataSet._fit_parametric_dispersion_trend (pydeseq2/dds.py:1199-1275) and
// DefaultInference.dispersion_trend_gamma_glm (pydeseq2/default_inference.py:200-230):
// an outer loop that (re)fits a 2-coefficient gamma GLM by L-BFGS-B (x0 = (1,1), lower bound
// 1e-12 on both coefficients, scipy defaults) and drops genes whose dispersion/prediction ratio
// leaves [1e-4, 15) until the coefficients stop moving.  It is a cross-gene step (needs every
// gene) but touches only two doubles per gene, so ONE wavefront runs the whole thing in one
// launch: lanes stride over the genes for the loss/gradient sums (compensated, fixed order =>
// run-to-run deterministic), the L-BFGS-B state machine (dsq_lbfgsb_dense.h, n = 2) runs
// wave-uniformly on an LDS workspace.  No host round trip per function evaluation.
#pragma once
#include "dsq_lbfgsb_dense.h"
#include "dsq_wave.h"

namespace dsq {

struct TrendWork {
    LbfgsbDenseWork<2> lb;
};

struct TrendOut {
    double a0, a1;
    int ok;       // 1: converged, 0: fit failed -> caller switches to the mean trend (dds.py:1243-1252)
    int n_outer;  // gamma-GLM fits performed
    int n_kept;   // genes in the last fit
};

// The three data passes of the fit, for `n` genes handled by `stride`-spaced workers starting at
// `first`: per-worker partial results; the caller reduces them (wave shuffle, or wave shuffle + LDS
// across the waves of a workgroup) in a fixed order.
struct TrendData {
    const double* disp;   // raw genewise dispersions (clipped to [min_disp, max_disp] on the fly, dds.py:792-794)
    const double* means;  // normalised means
    uint8_t* keep;        // scratch mask (global memory)
    int n;
    double min_disp, max_disp;
    // raw != 0: `means` holds the covariates themselves and `disp` the targets, unclipped — the inputs of
    // Inference.dispersion_trend_gamma_glm (inference.py:284-308), one gamma GLM without the outer loop
    int raw = 0;
};

struct TrendPartial {
    KSum s, g0, g1;
    int cf = 0, c0 = 0, c1 = 0;
};

DSQ_HD int trend_init_keep(const TrendData& D, int first, int stride) {
    int kept = 0;
    for (int i = first; i < D.n; i += stride) {
        const double c = D.raw ? 0.0 : 1.0 / D.means[i];
        const bool bad = (c != c) || (c == INFINITY) || (c == -INFINITY);  // dds.py:1225-1231
        D.keep[i] = bad ? 0 : 1;
        kept += bad ? 0 : 1;
    }
    return kept;
}

// loss = nanmean(t/mu + log mu), grad = -nanmean(((t/mu - 1) A)/mu)   (default_inference.py:209-217)
DSQ_HD void trend_eval_partial(const TrendData& D, int first, int stride, double a0, double a1,
                               TrendPartial& P) {
    for (int i = first; i < D.n; i += stride) {
        if (!D.keep[i]) continue;
        const double cov = D.raw ? D.means[i] : frcp(D.means[i]);
        const double t = D.raw ? D.disp[i] : dmin(dmax(D.disp[i], D.min_disp), D.max_disp);
        const double mu = a0 + a1 * cov;
        const double rmu = frcp(mu);
        const double tm = t * rmu;
        const double v = tm + flog(mu);
        if (v == v) { P.s.add(v); P.cf += 1; }
        const double r = tm - 1.0;
        const double v0 = r * rmu, v1 = (r * cov) * rmu;
        if (v0 == v0) { P.g0.add(v0); P.c0 += 1; }
        if (v1 == v1) { P.g1.add(v1); P.c1 += 1; }
    }
}

DSQ_HD int trend_filter(const TrendData& D, int first, int stride, double a0, double a1) {
    int k2 = 0;
    for (int i = first; i < D.n; i += stride) {
        if (!D.keep[i]) continue;
        const double t = dmin(dmax(D.disp[i], D.min_disp), D.max_disp);
        const double ratio = t / (a0 + a1 * (1.0 / D.means[i]));
        if (ratio < 1e-4 || ratio >= 15.0) D.keep[i] = 0;  // dds.py:1254-1264
        else k2 += 1;
    }
    return k2;
}

// Ops: int init_keep(); void eval(a0, a1, double& f, double* g); int filter(a0, a1)
template <class Ops>
DSQ_HD TrendOut trend_fit_core(Ops& ops, TrendWork& W, bool single = false) {
    TrendOut out;
    out.ok = 0; out.n_outer = 0; out.n_kept = 0;
    int kept = ops.init_keep();
    auto fg = [&](const double* c, double& f, double* g) { ops.eval(c[0], c[1], f, g); };
    double old0 = 0.1, old1 = 0.1, a0 = 1.0, a1 = 1.0;
    // x0 = (1, 1), lower bound 1e-12 on both coefficients (default_inference.py:219-225).  Local constants, not workspace
    // fields: inlined into the optimiser, the bound-type tests of its scalar code fold away (it runs on one wavefront
    // between two data passes - its latency is the kernel's)
    const double lo[2] = {1e-12, 1e-12}, up[2] = {0.0, 0.0};
    const int nbd[2] = {1, 1};
    if (single) {  // DefaultInference.dispersion_trend_gamma_glm (default_inference.py:200-230): one fit
        double x[2] = {1.0, 1.0};
        const LbfgsbResult res = lbfgsb_dense<2>(fg, 2, x, lo, up, nbd, W.lb);
        out.a0 = x[0]; out.a1 = x[1]; out.ok = res.success ? 1 : 0; out.n_outer = 1; out.n_kept = kept;
        return out;
    }
    for (;;) {
        if (!(a0 > 1e-10 && a1 > 1e-10)) break;
        const double l0 = log(fabs(a0 / old0)), l1 = log(fabs(a1 / old1));
        if (!(l0 * l0 + l1 * l1 >= 1e-6)) break;
        old0 = a0; old1 = a1;
        double x[2] = {1.0, 1.0};
        const LbfgsbResult res = lbfgsb_dense<2>(fg, 2, x, lo, up, nbd, W.lb);
        a0 = x[0]; a1 = x[1];
        out.n_outer += 1;
        out.n_kept = kept;
        if (!res.success || a0 <= 1e-10 || a1 <= 1e-10) {
            out.a0 = a0; out.a1 = a1; out.ok = 0;
            return out;
        }
        kept = ops.filter(a0, a1);
    }
    out.a0 = a0; out.a1 = a1; out.ok = 1;
    return out;
}

// single-wave (or host) implementation of the passes
template <class Wv>
struct WaveTrendOps {
    TrendData D;
    DSQ_HD int init_keep() { return Wv::sumi(trend_init_keep(D, Wv::lane(), Wv::W)); }
    DSQ_HD void eval(double a0, double a1, double& f, double* g) {
        TrendPartial P;
        trend_eval_partial(D, Wv::lane(), Wv::W, a0, a1, P);
        const double S = Wv::sum_comp(P.s), G0 = Wv::sum_comp(P.g0), G1 = Wv::sum_comp(P.g1);
        const int cf = Wv::sumi(P.cf), c0 = Wv::sumi(P.c0), c1 = Wv::sumi(P.c1);
        f = S / (double)cf;
        g[0] = -(G0 / (double)c0);
        g[1] = -(G1 / (double)c1);
    }
    DSQ_HD int filter(double a0, double a1) { return Wv::sumi(trend_filter(D, Wv::lane(), Wv::W, a0, a1)); }
};

template <class Wv>
DSQ_HD TrendOut trend_fit(const double* disp, const double* means, int n, double min_disp,
                          double max_disp, uint8_t* keep, TrendWork& W) {
    WaveTrendOps<Wv> ops;
    ops.D = TrendData{disp, means, keep, n, min_disp, max_disp};
    return trend_fit_core(ops, W);
}

}  // namespace dsq
