"""Summary tail on the device: adjusted p-values with independent filtering, through
the C ABI, against the oracle restatement of DeseqStats.summary() and the R fixtures."""
import zlib

import numpy as np
import pytest

from oracle import nbglm_oracle as orc
from tests.helpers import max_rel_err, r_csv

pytestmark = pytest.mark.gpu


def _same(a, b, rtol=1e-12):
    assert (np.isnan(a) == np.isnan(b)).all()
    ok = ~np.isnan(b)
    np.testing.assert_allclose(a[ok], b[ok], rtol=rtol, atol=0)


@pytest.mark.parametrize("case", ["plain", "ties", "ties354", "ties456", "many_zero_means", "few_rejections",
                                  "all_nan_but_few"])
def test_adjusted_pvalues_vs_oracle(case):
    from pydeseq2_amd import summary as sm
    from pydeseq2_amd._lib import Context

    # (hash() of a str is salted per process: this test used it and drew another vector in every run; two of the seeds on
    # which p-values tie exactly on the alpha boundary - where the oracle's BH then rounded differently from scipy's - are
    # now cases of their own)
    rng = np.random.default_rng(int(case[4:]) if case[4:].isdigit() else zlib.crc32(case.encode()) % 1000)
    if case.startswith("ties"):
        case = "ties"
    G = 5000
    bm = 10 ** rng.uniform(-1, 4, G)
    p = rng.uniform(0, 1, G) ** np.where(bm > 50, 6, 1.2)  # expressed genes carry the signal
    p[rng.random(G) < 0.03] = np.nan
    if case == "ties":
        p = np.round(p, 3)
        bm = np.round(bm, 0)
    if case == "many_zero_means":
        z = rng.random(G) < 0.2
        bm[z], p[z] = 0.0, np.nan
    if case == "few_rejections":
        p = rng.uniform(0, 1, G)
    if case == "all_nan_but_few":
        p[20:] = np.nan
    ctx = Context(0)
    for indep in (True, False):
        padj, info = sm.adjusted_pvalues(ctx, bm, p, 0.05, indep)
        if indep:
            ref, rinfo = orc.independent_filtering(bm, p, 0.05)
            np.testing.assert_allclose(info["theta"], rinfo["theta"], rtol=1e-14)
            np.testing.assert_allclose(info["cutoffs"], rinfo["cutoffs"], rtol=1e-14)
            assert (info["num_rej"] == rinfo["num_rej"]).all()
            assert info["j"] == rinfo["j"]
        else:
            ref = orc.p_value_adjustment(p)
        _same(padj, ref)


def test_summary_pipeline_vs_oracle_and_r():
    import pydeseq2_amd
    from pydeseq2_amd import summary as sm
    from tests.test_gpu_parity import _r_case

    counts, X = orc.synth_counts(3000, 60, "2level", 9)
    counts[2, 10] = 300000  # an outlier: exercises the Cook's filter
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, device=0)
    res = pipe.deseq2()
    s = sm.summary(res, [0, 1], ctx=pipe.ctx)
    ref = orc.summary(res, [0, 1])  # same inputs: isolates the summary tail
    for k in ("baseMean", "log2FoldChange", "lfcSE", "stat", "pvalue", "padj"):
        _same(s[k], ref[k])
    assert s["info"]["j"] == ref["info"]["j"]
    # the reference's own known answers (tests/test_pydeseq2.py:94-118, 148-176)
    counts, X, names = _r_case("synthetic", ["condition"])
    res = pydeseq2_amd.deseq2(counts, X, contrast=[0, 1], device=0)
    r_res = r_csv("single_factor", "r_test_res.csv")
    s = sm.summary(res, [0, 1])
    assert (np.isnan(s["padj"]) == r_res["padj"].isna().to_numpy()).all()
    assert max_rel_err(s["padj"], r_res["padj"].to_numpy()) < 0.02
    s2 = sm.summary(res, [0, 1], independent_filter=False)
    assert max_rel_err(s2["padj"], r_csv("single_factor", "r_test_res_no_independent_filtering.csv")["padj"].to_numpy()) < 0.02


@pytest.mark.parametrize("case", ["p2", "p4", "p8"])
def test_lfc_shrink_inference_vs_reference_kats(case):
    """HipInference.lfc_shrink_nbinom_glm against outputs of the unmodified utils.nbinomGLM."""
    import os

    from pydeseq2_amd import HipInference
    from tests.helpers import load_kat

    inf = HipInference(device=0)
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_shrink.npz"))
    kk = load_kat(case)
    G, sidx = int(k[f"{case}_G"]), int(k[f"{case}_sidx"])
    for tag in "ab":
        b, ih, cv = inf.lfc_shrink_nbinom_glm(kk["X"], kk["counts"][:, :G], k[f"{case}_size"], np.log(kk["sf"]), 15,
                                              float(k[f"{case}{tag}_scale"]), "L-BFGS-B", sidx)
        assert (cv == k[f"{case}{tag}_conv"]).all()
        np.testing.assert_allclose(b, k[f"{case}{tag}_beta"], rtol=1e-5, atol=1e-8)
        scale = np.abs(k[f"{case}{tag}_invh"]).max(axis=(1, 2), keepdims=True)
        assert np.max(np.abs(ih - k[f"{case}{tag}_invh"]) / scale) < 1e-6


@pytest.mark.parametrize("case", ["p5", "p6", "p7", "p9", "p10", "p11", "p12"])
def test_lfc_shrink_inference_vs_reference_kats_at_the_widths_between(case):
    """5 ... 12 design columns - the optimiser's inverse matrix in the wavefront's registers (dsq_lbfgsb_wave.h: one lane per
    entry of an 8 x 8 matrix up to 8 columns, four entries per lane of a 16 x 16 one from 9) - against outputs of the
    unmodified utils.nbinomGLM (kat_shrink_mid.npz), convergence flags included."""
    import os

    from pydeseq2_amd import HipInference
    from tests.helpers import load_kat

    inf = HipInference(device=0)
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_shrink_mid.npz"))
    kk = load_kat(case)
    G, sidx = int(k[f"{case}_G"]), int(k[f"{case}_sidx"])
    for tag in "ab":
        b, ih, cv = inf.lfc_shrink_nbinom_glm(kk["X"], kk["counts"][:, :G], k[f"{case}_size"], np.log(kk["sf"]), 15,
                                              float(k[f"{case}{tag}_scale"]), "L-BFGS-B", sidx)
        assert (cv == k[f"{case}{tag}_conv"]).all()
        np.testing.assert_allclose(b, k[f"{case}{tag}_beta"], rtol=1e-5, atol=1e-8)
        scale = np.abs(k[f"{case}{tag}_invh"]).max(axis=(1, 2), keepdims=True)
        assert np.max(np.abs(ih - k[f"{case}{tag}_invh"]) / scale) < 1e-6


@pytest.mark.parametrize("case", ["p2", "p4", "p8", "p12"])
@pytest.mark.parametrize("optimizer,tag", [("BFGS", "bfgs"), ("Newton-CG", "ncg")])
def test_lfc_shrink_other_optimizers_vs_reference_kats(case, optimizer, tag):
    """HipInference.lfc_shrink_nbinom_glm(optimizer="BFGS" | "Newton-CG") against the unmodified utils.nbinomGLM with the
    same optimizer (kat_shrink_opt.npz; tolerances as in tests/test_hostsim.py: Newton-CG tight, BFGS to the resolution of
    its stopping rule), convergence flags equal."""
    import os

    from pydeseq2_amd import HipInference
    from tests.helpers import load_kat

    inf = HipInference(device=0)
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_shrink_opt.npz"))
    kk = load_kat(case)
    G, sidx = int(k[f"{case}_G"]), int(k[f"{case}_sidx"])
    b, ih, cv = inf.lfc_shrink_nbinom_glm(kk["X"], kk["counts"][:, :G], k[f"{case}_size"], np.log(kk["sf"]), 15,
                                          float(k[f"{case}_scale"]), optimizer, sidx)
    same = cv == k[f"{case}_{tag}_conv"]
    if optimizer == "Newton-CG":
        assert same.all()
    else:
        # BFGS at gtol = 1e-8 on this flat scaled objective ends in "precision loss" for 1 of 24 of the reference's own
        # fits: whether the last line search still finds a step is decided by the last bits of the loss (the device's
        # exponential and fused multiply-adds are not numpy's) - one flag of a case may differ
        assert (~same).sum() <= 1
    rtol, atol = (1e-7, 1e-9) if optimizer == "Newton-CG" else (5e-4, 2e-6)
    np.testing.assert_allclose(b[same], k[f"{case}_{tag}_beta"][same], rtol=rtol, atol=atol)
    scale = np.abs(k[f"{case}_{tag}_invh"]).max(axis=(1, 2), keepdims=True)
    assert np.max((np.abs(ih - k[f"{case}_{tag}_invh"]) / scale)[same]) < (1e-6 if optimizer == "Newton-CG" else 5e-6)


def test_lfc_shrink_rejects_unknown_optimizers_and_wide_designs_with_other_optimizers():
    from pydeseq2_amd import HipInference
    from pydeseq2_amd._lib import DsqError
    from tests.helpers import load_kat

    inf = HipInference(device=0)
    kk = load_kat("p16")
    args = (kk["X"], kk["counts"][:, :4], np.full(4, 5.0), np.log(kk["sf"]), 15, 1.0)
    with pytest.raises(ValueError):
        inf.lfc_shrink_nbinom_glm(*args, "Powell", 1)
    with pytest.raises((ValueError, DsqError)):
        inf.lfc_shrink_nbinom_glm(*args, "Newton-CG", 1)  # 16 columns: L-BFGS-B only


@pytest.mark.parametrize("case", ["p16", "p24", "p40", "p48"])
def test_lfc_shrink_wide_designs_vs_reference_kats(case):
    """13 ... 48 design columns (k_shrink_wide: run-time p; 33 ... 48 since round 6, the optimiser's matrices in LDS) against
    outputs of the unmodified utils.nbinomGLM."""
    import os

    from pydeseq2_amd import HipInference
    from tests.helpers import load_kat

    inf = HipInference(device=0)
    k = np.load(os.path.join(os.path.dirname(__file__), "golden",
                             "kat_shrink_wider.npz" if case in ("p40", "p48") else "kat_shrink_wide.npz"))
    kk = load_kat(case)
    G, sidx = int(k[f"{case}_G"]), int(k[f"{case}_sidx"])
    for tag in "ab":
        b, ih, cv = inf.lfc_shrink_nbinom_glm(kk["X"], kk["counts"][:, :G], k[f"{case}_size"], np.log(kk["sf"]), 15,
                                              float(k[f"{case}{tag}_scale"]), "L-BFGS-B", sidx)
        assert (cv == k[f"{case}{tag}_conv"]).all()
        np.testing.assert_allclose(b, k[f"{case}{tag}_beta"], rtol=1e-5, atol=1e-8)
        scale = np.abs(k[f"{case}{tag}_invh"]).max(axis=(1, 2), keepdims=True)
        assert np.max(np.abs(ih - k[f"{case}{tag}_invh"]) / scale) < 1e-6


@pytest.mark.parametrize("kind,G,N", [("mixed14", 160, 120), ("mixed44", 60, 520)])
def test_lfc_shrink_pipeline_wide_design_vs_oracle(kind, G, N):
    """DeseqStats.lfc_shrink's device path on a 14-column design (two factors + nine continuous covariates) and on a
    44-column one (round 6: shrinkage up to 48 columns): the pipeline-level shrinkage (prior scale from the MLE LFCs,
    k_shrink_wide, shrunken LFC and lfcSE) against the oracle."""
    import pydeseq2_amd
    from pydeseq2_amd import summary as sm
    from tests.test_gpu_parity import _wide_case

    counts, X = _wide_case(kind, G, N, 41)
    counts[:, 3] = 0
    c = np.zeros(X.shape[1])
    c[1] = 1.0
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, device=0)
    res = pipe.deseq2(contrast=c)
    lfc, se, conv, scale = sm.lfc_shrink(pipe, res, 1)
    rl, rs, rc, rscale = orc.lfc_shrink(counts, X, res, 1)
    assert abs(scale - rscale) < 1e-12
    assert (np.isnan(conv) == np.isnan(rc)).all() and (conv[~np.isnan(rc)] == rc[~np.isnan(rc)]).all()
    _same(lfc, rl, rtol=1e-5)
    _same(se, rs, rtol=1e-5)


def test_lfc_shrink_pipeline_vs_oracle_and_r():
    import pydeseq2_amd
    from pydeseq2_amd import summary as sm
    from tests.test_gpu_parity import _r_case

    counts, X = orc.synth_counts(400, 60, "2level", 13)
    counts[:, 9] = 0
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, device=0)
    res = pipe.deseq2(contrast=[0, 1])
    lfc, se, conv, scale = sm.lfc_shrink(pipe, res, 1)
    rl, rs, rc, rscale = orc.lfc_shrink(counts, X, res, 1)
    assert abs(scale - rscale) < 1e-12
    assert (np.isnan(conv) == np.isnan(rc)).all() and (conv[~np.isnan(rc)] == rc[~np.isnan(rc)]).all()
    _same(lfc, rl, rtol=1e-5)
    _same(se, rs, rtol=1e-5)
    # R known answer (tests/test_pydeseq2.py:256-296): start from R's size factors / dispersions / LFC / SE
    counts, X, names = _r_case("synthetic", ["condition"])
    pipe = pydeseq2_amd.DeseqPipeline(counts, X, device=0)
    res = pipe.deseq2(contrast=[0, 1])
    r_res = r_csv("single_factor", "r_test_res.csv")
    res.size_factors = r_csv("single_factor", "r_test_size_factors.csv")["x"].to_numpy()
    res.dispersions = r_csv("single_factor", "r_test_dispersions.csv")["x"].to_numpy()
    res.LFC = res.LFC.copy()
    res.LFC[:, 1] = r_res["log2FoldChange"].to_numpy() * np.log(2)
    res.lfcSE = r_res["lfcSE"].to_numpy() * np.log(2)
    lfc, se, conv, scale = sm.lfc_shrink(pipe, res, 1)
    assert max_rel_err(lfc / np.log(2), r_csv("single_factor", "r_test_lfc_shrink_res.csv")["log2FoldChange"].to_numpy()) < 0.02
