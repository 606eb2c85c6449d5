"""The AnnData-free DeseqDataSet / DeseqStats façade end to end, replaying the reference's own
user-level tests (tests/test_pydeseq2.py:94-176, 180-226, 256-341, 432-509) against the R fixtures."""
import numpy as np
import pytest

from tests.helpers import load_dataset, r_csv

pytestmark = pytest.mark.gpu


def _almost(df, r_res, tol, cols=("log2FoldChange", "pvalue", "padj")):
    for c in cols:
        assert (df[c].isna() == r_res[c].isna()).all(), c
        assert ((df[c] - r_res[c]).abs() / r_res[c].abs()).max() < tol, c


def test_single_factor_summary_and_shrink():
    from pydeseq2_amd.api import DeseqDataSet, DeseqStats

    counts, meta = load_dataset("synthetic")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition").deseq2()
    np.testing.assert_array_almost_equal(dds.obs["size_factors"],
                                         r_csv("single_factor", "r_test_size_factors.csv")["x"].to_numpy(), decimal=6)
    assert dds.uns["disp_function_type"] == "parametric" and "trend_coeffs" in dds.uns
    assert list(dds.varm["LFC"].columns) == ["Intercept", "condition[T.B]"]
    ds = DeseqStats(dds, contrast=["condition", "B", "A"])
    df = ds.summary()
    _almost(df, r_csv("single_factor", "r_test_res.csv"), 0.02)
    # reversed contrast: sign flips, p-values stay
    df_rev = DeseqStats(dds, contrast=["condition", "A", "B"]).summary()
    np.testing.assert_allclose(df_rev["log2FoldChange"], -df["log2FoldChange"], rtol=1e-12)
    np.testing.assert_allclose(df_rev["pvalue"], df["pvalue"], rtol=1e-9)
    # no independent filtering
    df2 = DeseqStats(dds, contrast=["condition", "B", "A"], independent_filter=False).summary()
    _almost(df2, r_csv("single_factor", "r_test_res_no_independent_filtering.csv"), 0.02)
    # alternative hypotheses through summary(**kwargs) (tests/test_pydeseq2.py:180-226)
    for alt, null in (("greater", 0.5), ("lessAbs", 0.5)):
        d = DeseqStats(dds, contrast=["condition", "B", "A"]).summary(lfc_null=null, alt_hypothesis=alt)
        r = r_csv("single_factor", f"r_test_res_{alt}.csv")
        ok = r["stat"] != 0
        assert ((d["pvalue"][ok] - r["pvalue"][ok]).abs() / r["pvalue"][ok]).max() < 0.02
    # LFC shrinkage from R's inputs (tests/test_pydeseq2.py:256-296)
    r_res = r_csv("single_factor", "r_test_res.csv")
    dds.obs["size_factors"] = r_csv("single_factor", "r_test_size_factors.csv")["x"].to_numpy()
    dds.var["dispersions"] = r_csv("single_factor", "r_test_dispersions.csv")["x"].to_numpy()
    ds = DeseqStats(dds, contrast=["condition", "B", "A"])
    ds.summary()
    ds.LFC.iloc[:, 1] = r_res["log2FoldChange"].to_numpy() * np.log(2)
    ds.SE = r_res["lfcSE"] * np.log(2)
    shr = ds.lfc_shrink(coeff="condition[T.B]")
    r_shr = r_csv("single_factor", "r_test_lfc_shrink_res.csv")
    assert ((shr["log2FoldChange"] - r_shr["log2FoldChange"]).abs() / r_shr["log2FoldChange"].abs()).max() < 0.02
    with pytest.raises(KeyError):
        ds.lfc_shrink(coeff="nope")


def test_multi_factor_and_continuous():
    from pydeseq2_amd.api import DeseqDataSet, DeseqStats

    counts, meta = load_dataset("synthetic")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~group + condition").deseq2()
    df = DeseqStats(dds, contrast=["condition", "B", "A"]).summary()
    _almost(df, r_csv("multi_factor", "r_test_res.csv"), 0.04, cols=("log2FoldChange", "pvalue"))
    counts, meta = load_dataset("continuous")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~group + condition + measurement").deseq2()
    contrast = np.zeros(dds.obsm["design_matrix"].shape[1])
    contrast[-1] = 1
    df = DeseqStats(dds, contrast=contrast).summary()
    _almost(df, r_csv("continuous", "r_test_res.csv"), 0.04, cols=("log2FoldChange", "pvalue"))
    assert dds.layers["cooks"].shape == counts.shape and dds.layers["normed_counts"].shape == counts.shape


@pytest.mark.parametrize("use_design,fit_type,fn", [(False, None, "r_vst.csv"), (True, None, "r_vst_with_design.csv"),
                                                    (False, "mean", "r_mean_vst.csv")])
def test_vst(use_design, fit_type, fn):
    """tests/test_pydeseq2.py:761-805 + the oracle restatement on the same inputs."""
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition")
    out = dds.vst(use_design=use_design, fit_type=fit_type)
    r_vst = r_csv("single_factor", fn).T.to_numpy()
    assert np.max(np.abs(r_vst - out) / r_vst) < 0.02
    ref, _ = orc.vst(counts.to_numpy(), dds.obsm["design_matrix"].to_numpy(), use_design=use_design,
                     fit_type=fit_type or "parametric")
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-9)
    assert dds.layers["vst_counts"].shape == counts.shape


def test_size_factor_modes():
    """tests/test_pydeseq2.py:56-91: poscounts vs R, control genes, and both against the oracle."""
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    c = counts.to_numpy()
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition").fit_size_factors("poscounts")
    r_sf = r_csv("single_factor", "r_test_size_factors_poscount.csv")["sizeFactor"].to_numpy()
    np.testing.assert_array_almost_equal(dds.obs["size_factors"], r_sf)
    np.testing.assert_allclose(dds.obs["size_factors"], orc.size_factors_poscounts(c), rtol=1e-12)
    expect = c[:, 3] / np.exp(np.log(c[:, 3]).mean())
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition", control_genes=["gene4"]).fit_size_factors()
    np.testing.assert_array_almost_equal(dds.obs["size_factors"], expect)
    dds.fit_size_factors(fit_type="poscounts")
    np.testing.assert_array_almost_equal(dds.obs["size_factors"], expect)
    # vst_fit() refits the size factors through fit_size_factors, which falls back to the data set's control genes
    # (dds.py:404-407, 628-631): both vst routes keep them
    mask = np.zeros(c.shape[1], dtype=bool)
    mask[[1, 3, 6]] = True
    for use_design in (False, True):
        dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition", control_genes=np.nonzero(mask)[0])
        dds.vst_fit(use_design=use_design)
        np.testing.assert_allclose(dds.obs["size_factors"], orc.size_factors_control(c, mask), rtol=1e-12)
    # zeros in the matrix: poscounts uses the positive entries only
    z = c.copy()
    z[::3, ::2] = 0
    import pandas as pd
    dds = DeseqDataSet(counts=pd.DataFrame(z, index=counts.index, columns=counts.columns), metadata=meta,
                       design="~condition", size_factors_fit_type="poscounts").fit_size_factors()
    np.testing.assert_allclose(dds.obs["size_factors"], orc.size_factors_poscounts(z), rtol=1e-12)


def _shrink_from_r(dds, sub, coeff, contrast):
    """The reference's shrinkage tests start from R's size factors, dispersions, LFC column 1 and SE."""
    from pydeseq2_amd.api import DeseqStats

    r_res = r_csv(sub, "r_test_res.csv")
    dds.obs["size_factors"] = r_csv(sub, "r_test_size_factors.csv").squeeze().to_numpy()
    dds.var["dispersions"] = r_csv(sub, "r_test_dispersions.csv").squeeze().to_numpy()
    dds.varm["LFC"].iloc[:, 1] = r_res["log2FoldChange"].to_numpy() * np.log(2)
    ds = DeseqStats(dds, contrast=contrast)
    ds.summary()
    ds.SE = r_res["lfcSE"] * np.log(2)
    shr = ds.lfc_shrink(coeff=coeff)
    r_shr = r_csv(sub, "r_test_lfc_shrink_res.csv")
    return ((r_shr["log2FoldChange"] - shr["log2FoldChange"]).abs() / r_shr["log2FoldChange"].abs()).max()


def test_lfc_shrink_multi_factor_continuous_and_large_counts():
    """tests/test_pydeseq2.py:367-430, 470-509, 566-622."""
    import pandas as pd

    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~group + condition").deseq2()
    assert _shrink_from_r(dds, "multi_factor", "condition[T.B]", ["condition", "B", "A"]) < 0.02
    counts, meta = load_dataset("continuous")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~group + condition + measurement").deseq2()
    cv = np.zeros(dds.obsm["design_matrix"].shape[1])
    cv[-1] = 1
    assert _shrink_from_r(dds, "continuous", "measurement", cv) < 0.02
    counts = pd.DataFrame(
        [[25, 405, 1355, 12558, 489843], [28, 480, 2144, 13844, 514571], [12, 690, 1919, 15632, 564106],
         [31, 420, 1684, 11513, 556380], [34, 278, 3849, 11577, 412551], [19, 249, 3086, 7296, 295565],
         [17, 491, 4089, 13805, 280945], [15, 251, 2785, 10492, 214062]],
        index=["A1", "A2", "A3", "A4", "B1", "B2", "B3", "B4"], columns=["g1", "g2", "g3", "g4", "g5"])
    meta = pd.DataFrame({"condition": list("AAAABBBB")}, index=counts.index)
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition").deseq2()
    assert _shrink_from_r(dds, "large_counts", "condition[T.B]", ["condition", "B", "A"]) < 0.02


def test_iterative_size_factors():
    """tests/test_pydeseq2.py:344-364 + the automatic switch when every gene contains a zero (dds.py:682-690)."""
    import warnings

    import pydeseq2_amd
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition").fit_size_factors("iterative")
    r = r_csv("single_factor", "r_iterative_size_factors.csv").squeeze().to_numpy()
    assert np.max(np.abs(r - dds.obs["size_factors"]) / np.abs(r)) < 0.02
    ref = orc.size_factors_iterative(counts.to_numpy())
    np.testing.assert_allclose(dds.obs["size_factors"], ref, rtol=1e-4)
    # every gene with a zero -> deseq2() switches by itself
    c, X = orc.synth_counts(60, 24, "2level", 21)
    c[np.arange(60) % 24, np.arange(60)] = 0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = pydeseq2_amd.deseq2(c, X, device=0)
        ref = orc.deseq2(c, X, n_jobs=2)
    assert any("iterative" in str(x.message) for x in w)
    np.testing.assert_allclose(res.size_factors, ref.size_factors, rtol=1e-4)
    ok = ~np.isnan(ref.dispersions)
    np.testing.assert_allclose(res.dispersions[ok], ref.dispersions[ok], rtol=5e-3)
    np.testing.assert_allclose(res.LFC[ok], ref.LFC[ok], rtol=5e-3, atol=1e-4)


def test_vst_train_test_split():
    """tests/test_pydeseq2.py:826-929: fit on samples 25..75, transform 0..25 with the training logmeans."""
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    train, test = counts[25:75], counts[0:25]
    dds = DeseqDataSet(counts=train, metadata=meta[25:75], design="~condition")
    with pytest.raises(RuntimeError):
        dds.vst_transform(test.to_numpy())
    dds.vst_fit()
    assert "vst_trend_coeffs" in dds.uns and "size_factors" in dds.obs
    out = dds.vst_transform(test.to_numpy())
    assert isinstance(out, np.ndarray) and out.shape == (25, 10)
    _, info = orc.vst(train.to_numpy(), dds.obsm["design_matrix"].to_numpy())
    ref = orc.vst_transform_new(test.to_numpy(), train.to_numpy(), info)
    np.testing.assert_allclose(out, ref, rtol=1e-6)
    # the dataset's own samples: same as vst()
    np.testing.assert_allclose(dds.vst_transform(), dds.vst(), rtol=1e-12)


def test_vst_new_samples_with_zero_counts():
    """New samples may hold zeros in genes that were usable in training: log(0) - logmean = -inf takes part in the
    sample's median (preprocessing.py:59-102); a sample with more than half zeros gets size factor 0 as in numpy."""
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    train = counts[25:75]
    test = counts[0:25].to_numpy().copy()
    test[0, :3] = 0          # a few zeros: the median moves down among the finite ratios
    test[1, :6] = 0          # more than half of the 10 genes: the median is -inf -> size factor 0
    dds = DeseqDataSet(counts=train, metadata=meta[25:75], design="~condition")
    dds.vst_fit()
    out = dds.vst_transform(test)
    _, info = orc.vst(train.to_numpy(), dds.obsm["design_matrix"].to_numpy())
    with np.errstate(all="ignore"):
        ref = orc.vst_transform_new(test, train.to_numpy(), info)
    ok = np.isfinite(ref)
    assert (np.isfinite(out) == ok).all()
    np.testing.assert_allclose(out[ok], ref[ok], rtol=1e-6)


def test_more_than_65535_genes():
    """Genes sit on grid.x of every launch (grid.y is limited to 65535): a dataset with 70 000 genes, some of them
    all-zero so that the row gather of the non-zero genes runs, against per-gene results of a 2 000-gene slice."""
    import pydeseq2_amd
    from oracle import nbglm_oracle as orc

    counts, X = orc.synth_counts(70000, 24, "2level", 9)
    counts[:, ::1000] = 0
    res = pydeseq2_amd.deseq2(counts, X, device=0)
    assert (~res.non_zero).sum() >= 70 and np.isnan(res.dispersions[::1000]).all()
    nz = res.non_zero
    assert np.isfinite(res.dispersions[nz]).all() and np.isfinite(res.LFC[nz]).all()
    # the genewise fits depend on the gene and the size factors only: compare the last 2000 genes with the oracle
    sl = slice(68000, 70000)
    sub = counts[:, sl]
    keep = ~(sub == 0).all(0) & ~res.replaced[sl]  # refitted genes carry the dispersions of the replaced counts
    mu = orc.lin_reg_mu(sub[:, keep], res.size_factors, X, 0.5)
    a, cv = orc.alpha_mle(sub[:, keep], X, mu, res.mom_dispersions[sl][keep], 1e-8, 24.0, n_jobs=8)
    same = cv & (res.genewise_converged[sl][keep] == 1)
    assert same.mean() > 0.99
    # (atol: a dispersion at the lower bound 1e-8 is compared to 1e-12 absolute - the loss is flat there)
    np.testing.assert_allclose(res.genewise_dispersions[sl][keep][same], np.clip(a, 1e-8, 24.0)[same], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_upload_narrows_chunks_to_uint16_where_the_counts_fit(dtype, monkeypatch):
    """dsq_upload_counts_i32: a chunk (8 Mi counts) whose values are all below 65 536 travels as uint16 and is widened on
    the device, any other as int32 - the device matrix is the host matrix either way; a negative count is reported."""
    import ctypes as C

    from pydeseq2_amd._lib import Context, DeviceArray

    ctx = Context(0)
    rng = np.random.default_rng(11)
    n = (8 << 20) * 2 + 12345  # two full chunks and a ragged one
    host = rng.integers(0, 60000, n).astype(dtype)
    host[(8 << 20) + 777] = 70000          # second chunk: int32
    host[n - 3] = 65535                    # last chunk: still uint16
    d = DeviceArray(ctx, (n,), np.int32)
    bad = C.c_int(0)
    ctx.call("dsq_upload_counts_i32", C.c_void_p(host.ctypes.data), 0 if dtype == np.int32 else 1, C.c_size_t(n),
             C.c_void_p(d.ptr), C.byref(bad))
    assert bad.value == 0
    assert np.array_equal(d.to_host(), host.astype(np.int32))
    host[5] = -1
    ctx.call("dsq_upload_counts_i32", C.c_void_p(host.ctypes.data), 0 if dtype == np.int32 else 1, C.c_size_t(n),
             C.c_void_p(d.ptr), C.byref(bad))
    assert bad.value == 1


def test_stage_wise_lazy_chaining_and_user_edited_fields():
    """dds.py:725, 812, 849, 892, 944, 992: a stage whose input field is missing runs the stage that writes it - fit_LFC()
    on a fresh data set runs everything before it; size factors the user put into obs (dds.py:724-726 fits them only when
    the column is missing) and dispersions edited between fit_MAP_dispersions() and fit_LFC() are honoured."""
    import pydeseq2_amd
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    one_go = DeseqDataSet(counts=counts, metadata=meta, design="~group + condition").deseq2()
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~group + condition")
    dds.fit_LFC()
    for f in ("genewise_dispersions", "fitted_dispersions", "MAP_dispersions", "dispersions"):
        np.testing.assert_array_equal(dds.var[f].to_numpy(), one_go.var[f].to_numpy(), err_msg=f)
    assert "prior_disp_var" in dds.uns and "size_factors" in dds.obs
    np.testing.assert_array_equal(dds.varm["LFC"].to_numpy(), one_go.varm["LFC"].to_numpy())
    # the open pass is finished when somebody needs the whole fit
    df = __import__("pydeseq2_amd.api", fromlist=["DeseqStats"]).DeseqStats(dds, contrast=["condition", "B", "A"]).summary()
    assert df["padj"].notna().all()
    # user-provided size factors
    sf = np.linspace(0.7, 1.4, len(meta))
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition")
    dds.obs["size_factors"] = sf
    dds.fit_genewise_dispersions()
    X = dds.obsm["design_matrix"].to_numpy()
    ref = pydeseq2_amd.DeseqPipeline(counts.to_numpy(), X, device=0).deseq2(size_factors=sf)
    np.testing.assert_array_equal(dds.obs["size_factors"].to_numpy(), sf)
    np.testing.assert_allclose(dds.var["genewise_dispersions"], ref.genewise_dispersions, rtol=1e-12)
    # dispersions edited between the stages
    dds.fit_MAP_dispersions()
    mine = dds.var["dispersions"].to_numpy() * 1.5
    dds.var["dispersions"] = mine
    dds.fit_LFC()
    beta, _, _, _ = orc.irls(counts.to_numpy(), sf, X, mine, n_jobs=1)
    np.testing.assert_allclose(dds.varm["LFC"].to_numpy(), beta, rtol=1e-6, atol=1e-9)


def test_layers_are_rebuilt_on_demand_copy_and_slices():
    """The N x G layers live on the device until read; one that a later pass has recycled is rebuilt by running the pass
    again (layers["_mu_hat"] after deseq2(), then layers["cooks"]): all of them against the oracle's layers.  copy() gives
    an independent data set, dds[:, genes] slices the fitted fields and the layers."""
    from oracle import nbglm_oracle as orc
    from pydeseq2_amd.api import DeseqDataSet

    counts, meta = load_dataset("synthetic")
    c = counts.copy()
    c["gene3"] = 0
    dds = DeseqDataSet(counts=c, metadata=meta, design="~group + condition").deseq2()
    ref = orc.deseq2(c.to_numpy(), dds.obsm["design_matrix"].to_numpy(), n_jobs=1, keep_layers=True)
    mu_hat = dds.layers["_mu_hat"]       # re-opens a pass: the LFC fit's device layers are recycled ...
    cooks = dds.layers["cooks"]          # ... and rebuilt here
    nz = ref.non_zero
    np.testing.assert_allclose(mu_hat[:, nz], ref.mu_hat[:, nz], rtol=1e-6)
    assert np.isnan(mu_hat[:, ~nz]).all() and np.isnan(cooks[:, ~nz]).all()
    np.testing.assert_allclose(cooks[:, nz], ref.cooks[:, nz], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(dds.obsm["_mu_LFC"], ref.mu_LFC, rtol=1e-6)
    np.testing.assert_allclose(dds.obsm["_hat_diagonals"], ref.hat_diagonals, rtol=1e-6)
    np.testing.assert_allclose(dds.layers["normed_counts"], c.to_numpy() / ref.size_factors[:, None], rtol=1e-12)
    sub = dds[:, ["gene1", "gene3", "gene7"]]
    assert sub.n_vars == 3 and list(sub.var_names) == ["gene1", "gene3", "gene7"]
    np.testing.assert_array_equal(sub.var["dispersions"].to_numpy(), dds.var.loc[["gene1", "gene3", "gene7"], "dispersions"])
    assert np.isnan(sub.varm["LFC"].loc["gene3"]).all()
    np.testing.assert_array_equal(sub.layers["cooks"], cooks[:, [0, 2, 6]])
    rows = dds[["sample1", "sample5"]]
    assert rows.n_obs == 2 and rows.obsm["design_matrix"].shape[0] == 2
    cp = dds.copy()
    cp.var["dispersions"] = 1.0
    assert not np.allclose(dds.var["dispersions"].dropna(), 1.0)
    np.testing.assert_array_equal(cp.layers["cooks"], cooks)
    cp.fit_type = "mean"
    cp.deseq2()  # a copy fits on its own
    assert cp.uns["disp_function_type"] == "mean" and dds.uns["disp_function_type"] == "parametric"


def test_statistics_on_a_gene_slice_of_a_fitted_data_set():
    """dds[:, genes] of a fitted data set carries the fit in its fields (AnnData semantics): DeseqStats on the slice runs the
    Wald test with the parent's size factors, dispersions and LFCs - p-values equal the parent's for those genes, no refit."""
    from pydeseq2_amd.api import DeseqDataSet, DeseqStats

    counts, meta = load_dataset("synthetic")
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition").deseq2()
    full = DeseqStats(dds, contrast=["condition", "B", "A"], independent_filter=False)
    full.summary()
    genes = ["gene2", "gene5", "gene9"]
    sub = dds[:, genes]
    assert sub._res is None and sub._pipe_obj is None
    part = DeseqStats(sub, contrast=["condition", "B", "A"], independent_filter=False)
    part.run_wald_test()
    np.testing.assert_allclose(part.p_values.to_numpy(), full.p_values.loc[genes].to_numpy(), rtol=1e-10)
    np.testing.assert_allclose(part.SE.to_numpy(), full.SE.loc[genes].to_numpy(), rtol=1e-10)
    np.testing.assert_array_equal(sub.var["dispersions"].to_numpy(), dds.var.loc[genes, "dispersions"].to_numpy())


def test_mixed_design_through_the_facade_with_the_lfc_fit_in_two_launches():
    """A categorical factor + a continuous covariate at a size where deseq2() fits the LFCs in two launches (mixed designs,
    pipeline._fork_lfc): the one-go fit, the stage-by-stage fit (one launch: the pass is not known to run to its end) and
    a one-go fit with the fork switched off give the same fields bit for bit; layers (Cook's distances in slot order on
    the device) included."""
    import pandas as pd

    from pydeseq2_amd.api import DeseqDataSet, DeseqStats

    rng = np.random.default_rng(5)
    N, G = 384, 2600  # (three cells of 128 samples: whole 64-sample trips, the mixed-design kernels take it)
    cond = np.array(["A", "B", "C"])[np.arange(N) % 3]
    x = rng.normal(0, 1, N)
    meta = pd.DataFrame({"condition": cond, "x": x}, index=[f"s{i}" for i in range(N)])
    base = np.exp(rng.normal(4, 1.5, G))
    mu = base[None, :] * np.exp(0.3 * x)[:, None] * np.where(cond == "B", 1.4, 1.0)[:, None]
    size = 1.0 / (4.0 / base + 0.1)
    counts = pd.DataFrame(rng.negative_binomial(size[None, :], size[None, :] / (size[None, :] + mu)), index=meta.index,
                          columns=[f"g{j}" for j in range(G)])
    counts.iloc[:, 11] = 0
    one = DeseqDataSet(counts=counts, metadata=meta, design="~condition + x", continuous_factors=["x"]).deseq2()
    assert one._pipe._row_mode == 3 and one._pipe.lfc_forks == 1
    off = DeseqDataSet(counts=counts, metadata=meta, design="~condition + x", continuous_factors=["x"])
    off._pipe._lfc_overlap = False
    off.deseq2()
    assert off._pipe.lfc_forks == 0
    step = DeseqDataSet(counts=counts, metadata=meta, design="~condition + x", continuous_factors=["x"])
    step.fit_size_factors(); step.fit_genewise_dispersions(); step.fit_dispersion_trend(); step.fit_dispersion_prior()
    step.fit_MAP_dispersions(); step.fit_LFC(); step.calculate_cooks(); step.refit()
    for other in (off, step):
        for k in ("genewise_dispersions", "MAP_dispersions", "dispersions"):
            np.testing.assert_array_equal(one.var[k].to_numpy(), other.var[k].to_numpy(), err_msg=k)
        np.testing.assert_array_equal(one.varm["LFC"].to_numpy(), other.varm["LFC"].to_numpy())
        np.testing.assert_array_equal(one.layers["cooks"], other.layers["cooks"])
    a = DeseqStats(one, contrast=["condition", "B", "A"]).summary()
    b = DeseqStats(off, contrast=["condition", "B", "A"]).summary()
    pd.testing.assert_frame_equal(a, b)
