"""The reference's user-level test-suite, test by test, against the façade.

`/root/reference/tests/test_pydeseq2.py` and `test_edge_cases.py` cannot be executed against this engine anywhere: the
build container holds the reference but no GPU, the GPU box holds no reference (and its files may not travel).  Every test
function of the two files is therefore RESTATED here under the reference's own name (prefix ``test_ref_``), with the
reference's inputs (the synthetic data set and the R fixtures, copied as data under tests/golden/), the reference's call
sequence on ``DeseqDataSet`` / ``DeseqStats`` and the reference's tolerances; the docstring of each names the lines it
follows.  `profiles/r06_refsuite_map.md` is the table of all of them with their status.
"""
import pickle

import numpy as np
import pandas as pd
import pytest

from tests.helpers import load_dataset, r_csv

pytestmark = pytest.mark.gpu


def _dds(*a, **k):
    from pydeseq2_amd.api import DeseqDataSet

    return DeseqDataSet(*a, **k)


def _ds(*a, **k):
    from pydeseq2_amd.api import DeseqStats

    return DeseqStats(*a, **k)


@pytest.fixture
def counts_df():
    return load_dataset("synthetic")[0]


@pytest.fixture
def metadata():
    return load_dataset("synthetic")[1]


def assert_res_almost_equal(py_res, r_res, tol=0.02):
    """tests/test_pydeseq2.py:932-942."""
    assert (py_res.pvalue.isna() == r_res.pvalue.isna()).all()
    assert (py_res.padj.isna() == r_res.padj.isna()).all()
    assert (abs(r_res.log2FoldChange - py_res.log2FoldChange) / abs(r_res.log2FoldChange)).max() < tol
    assert (abs(r_res.pvalue - py_res.pvalue) / r_res.pvalue).max() < tol
    assert (abs(r_res.padj - py_res.padj) / r_res.padj).max() < tol


# ---------------------------------------------------------------------------------------- tests/test_pydeseq2.py
def test_ref_size_factors_ratio(counts_df, metadata):
    """test_pydeseq2.py:40-53."""
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition")
    dds.fit_size_factors()
    np.testing.assert_array_almost_equal(dds.obs["size_factors"], r_csv("single_factor", "r_test_size_factors.csv")["x"].values)


def test_ref_size_factors_poscounts(counts_df, metadata):
    """test_pydeseq2.py:56-68."""
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition")
    dds.fit_size_factors("poscounts")
    np.testing.assert_array_almost_equal(dds.obs["size_factors"],
                                         r_csv("single_factor", "r_test_size_factors_poscount.csv")["sizeFactor"].values)


def test_ref_size_factors_control_genes(counts_df, metadata):
    """test_pydeseq2.py:71-91."""
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition", control_genes=["gene4"])
    dds.fit_size_factors()
    expect = counts_df["gene4"] / np.exp(np.log(counts_df["gene4"]).mean())
    np.testing.assert_array_almost_equal(dds.obs["size_factors"], expect)
    dds.fit_size_factors(fit_type="poscounts")
    np.testing.assert_array_almost_equal(dds.obs["size_factors"], expect)


@pytest.mark.parametrize("fit_type,fn,indep", [("parametric", "r_test_res.csv", True),
                                               ("mean", "r_test_res_mean_curve.csv", True),
                                               ("parametric", "r_test_res_no_independent_filtering.csv", False)])
def test_ref_deseq_filtering_and_fit_types(counts_df, metadata, fit_type, fn, indep, tol=0.02):
    """test_pydeseq2.py:94-118 (independent filtering, parametric), 121-145 (mean fit), 148-176 (no independent filtering)."""
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition", fit_type=fit_type)
    dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"], independent_filter=indep)
    ds.summary()
    assert_res_almost_equal(ds.results_df, r_csv("single_factor", fn), tol)


@pytest.mark.parametrize("alt_hypothesis", ["lessAbs", "greaterAbs", "less", "greater"])
def test_ref_alt_hypothesis(alt_hypothesis, counts_df, metadata, tol=0.02):
    """test_pydeseq2.py:180-226."""
    r_res = r_csv("single_factor", f"r_test_res_{alt_hypothesis}.csv")
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition", n_cpus=2)
    dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"], lfc_null=-0.5 if alt_hypothesis == "less" else 0.5,
             alt_hypothesis=alt_hypothesis, n_cpus=2)
    ds.summary()
    res = ds.results_df
    assert (res.pvalue.isna() == r_res.pvalue.isna()).all()
    assert (res.padj.isna() == r_res.padj.isna()).all()
    assert (abs(r_res.log2FoldChange - res.log2FoldChange) / abs(r_res.log2FoldChange)).max() < tol
    if alt_hypothesis == "lessAbs":
        res.stat = res.stat.abs()
    assert (abs(r_res.stat - res.stat) / abs(r_res.stat)).max() < tol
    assert (abs(r_res.pvalue[r_res.stat != 0] - res.pvalue[res.stat != 0]) / r_res.pvalue[r_res.stat != 0]).max() < tol


def test_ref_deseq_no_refit_cooks(counts_df, metadata, tol=0.02):
    """test_pydeseq2.py:228-253."""
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition", refit_cooks=False)
    dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"])
    ds.summary()
    assert_res_almost_equal(ds.results_df, r_csv("single_factor", "r_test_res.csv"), tol)


def _shrink_from_r(sub, counts, meta, design, contrast, coeff, adapt=True, fn="r_test_lfc_shrink_res.csv", tol=0.02):
    """The shrinkage tests' common body: fit, then overwrite size factors, dispersions and LFC column 1 with R's, summary,
    R's standard errors, lfc_shrink (test_pydeseq2.py:256-296 and its siblings)."""
    r_res, r_shr = r_csv(sub, "r_test_res.csv"), r_csv(sub, fn)
    dds = _dds(counts=counts, metadata=meta, design=design)
    dds.deseq2()
    dds.obs["size_factors"] = r_csv(sub, "r_test_size_factors.csv").squeeze().values
    dds.var["dispersions"] = r_csv(sub, "r_test_dispersions.csv").squeeze().values
    dds.varm["LFC"].iloc[:, 1] = r_res.log2FoldChange.values * np.log(2)
    res = _ds(dds, contrast=contrast(dds) if callable(contrast) else contrast)
    res.summary()
    res.SE = r_res.lfcSE * np.log(2)
    res.lfc_shrink(coeff=coeff, adapt=adapt)
    shr = res.results_df
    assert (abs(r_shr.log2FoldChange - shr.log2FoldChange) / abs(r_shr.log2FoldChange)).max() < tol


def test_ref_lfc_shrinkage(counts_df, metadata):
    """test_pydeseq2.py:256-296."""
    _shrink_from_r("single_factor", counts_df, metadata, "~condition", ["condition", "B", "A"], "condition[T.B]")


def test_ref_lfc_shrinkage_no_apeAdapt(counts_df, metadata):
    """test_pydeseq2.py:299-341."""
    _shrink_from_r("single_factor", counts_df, metadata, "~condition", ["condition", "B", "A"], "condition[T.B]", adapt=False,
                   fn="r_test_lfc_shrink_no_apeAdapt_res.csv")


def test_ref_iterative_size_factors(counts_df, metadata, tol=0.02):
    """test_pydeseq2.py:344-364."""
    r_sf = r_csv("single_factor", "r_iterative_size_factors.csv").squeeze()
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition")
    dds._fit_iterate_size_factors()
    assert (abs(r_sf.values - dds.obs["size_factors"].values) / abs(r_sf.values)).max() < tol


def test_ref_lfc_shrinkage_large_counts():
    """test_pydeseq2.py:367-430 (a gene with counts of half a million)."""
    idx = ["A1", "A2", "A3", "A4", "B1", "B2", "B3", "B4"]
    counts = pd.DataFrame([[25, 405, 1355, 12558, 489843], [28, 480, 2144, 13844, 514571], [12, 690, 1919, 15632, 564106],
                           [31, 420, 1684, 11513, 556380], [34, 278, 3849, 11577, 412551], [19, 249, 3086, 7296, 295565],
                           [17, 491, 4089, 13805, 280945], [15, 251, 2785, 10492, 214062]], index=idx,
                          columns=["g1", "g2", "g3", "g4", "g5"])
    meta = pd.DataFrame(list("AAAABBBB"), index=idx, columns=["condition"])
    _shrink_from_r("large_counts", counts, meta, "~condition", ["condition", "B", "A"], "condition[T.B]")


@pytest.mark.parametrize("with_outliers", [True, False])
def test_ref_multifactor_deseq(counts_df, metadata, with_outliers, tol=0.04):
    """test_pydeseq2.py:435-467."""
    r_res = r_csv("multi_factor", "r_test_res_outliers.csv" if with_outliers else "r_test_res.csv")
    if with_outliers:
        counts_df.loc["sample1", "gene1"] = 2000
        counts_df.loc["sample11", "gene7"] = 1000
        metadata.loc["sample1", "condition"] = "C"
    dds = _dds(counts=counts_df, metadata=metadata, design="~group + condition")
    dds.deseq2()
    res = _ds(dds, contrast=["condition", "B", "A"])
    res.summary()
    assert_res_almost_equal(res.results_df, r_res, tol)


def test_ref_multifactor_lfc_shrinkage(counts_df, metadata):
    """test_pydeseq2.py:470-509."""
    _shrink_from_r("multi_factor", counts_df, metadata, "~group + condition", ["condition", "B", "A"], "condition[T.B]")


@pytest.mark.parametrize("with_outliers", [True, False])
def test_ref_continuous_deseq(with_outliers, tol=0.04):
    """test_pydeseq2.py:514-563."""
    counts, meta = load_dataset("continuous")
    r_res = r_csv("continuous", "r_test_res_outliers.csv" if with_outliers else "r_test_res.csv")
    if with_outliers:
        counts.loc["sample1", "gene1"] = 2000
        counts.loc["sample11", "gene7"] = 1000
        meta.loc["sample1", "condition"] = "C"
    dds = _dds(counts=counts, metadata=meta, design="~group + condition + measurement")
    dds.deseq2()
    cv = np.zeros(dds.obsm["design_matrix"].shape[1])
    cv[-1] = 1
    ds = _ds(dds, contrast=cv)
    ds.summary()
    assert_res_almost_equal(ds.results_df, r_res, tol)


def test_ref_continuous_lfc_shrinkage():
    """test_pydeseq2.py:566-622."""
    counts, meta = load_dataset("continuous")

    def cv(dds):
        v = np.zeros(dds.obsm["design_matrix"].shape[1])
        v[-1] = 1
        return v

    _shrink_from_r("continuous", counts, meta, "~group + condition + measurement", cv, "measurement")


@pytest.mark.parametrize("low_memory", [True, False])
def test_ref_wide_deseq(low_memory, tol=0.02):
    """test_pydeseq2.py:625-660: more genes than samples, with and without ``low_memory``."""
    counts, meta = load_dataset("wide")
    dds = _dds(counts=counts, metadata=meta, design="~group + condition", low_memory=low_memory)
    dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"])
    ds.summary()
    assert_res_almost_equal(ds.results_df, r_csv("wide", "r_test_res.csv"), tol)
    # what low_memory means (dds.py:228, 933-935, 1032-1034, 1103-1106): the N x G intermediates are gone afterwards
    for key in ("_mu_hat", "cooks"):
        assert (key in dds.layers) == (not low_memory), key
    for key in ("_mu_LFC", "_hat_diagonals"):
        assert (key in dds.obsm) == (not low_memory), key
    if not low_memory:
        assert dds.layers["cooks"].shape == counts.shape and dds.obsm["_mu_LFC"].shape[0] == counts.shape[0]


def test_ref_contrast(counts_df, metadata):
    """test_pydeseq2.py:663-693: ['condition', 'B', 'A'] vs ['condition', 'A', 'B']."""
    dds = _dds(counts=counts_df, metadata=metadata, design="~group + condition")
    dds.deseq2()
    ba, ab = _ds(dds, contrast=["condition", "B", "A"]), _ds(dds, contrast=["condition", "A", "B"])
    ba.summary()
    ab.summary()
    for col in ba.results_df.columns:
        np.testing.assert_array_almost_equal(ba.results_df[col].abs().values, ab.results_df[col].abs().values, decimal=8)
    np.testing.assert_array_almost_equal(ba.results_df.log2FoldChange.values, -ab.results_df.log2FoldChange.values, decimal=8)
    np.testing.assert_array_almost_equal(ba.results_df.stat.values, -ab.results_df.stat.values, decimal=8)


def test_ref_anndata_init(counts_df, metadata, tol=0.02):
    """test_pydeseq2.py:696-728: a data set built from an AnnData-like object whose .var already holds a column named like
    one of the engine's own ("dispersions").  anndata is not installed here: the object is duck-typed."""
    from types import SimpleNamespace

    rng = np.random.RandomState(42)
    var = pd.DataFrame({"dummy_param": rng.randn(counts_df.shape[1]), "dispersions": rng.randn(counts_df.shape[1]) ** 2},
                       index=counts_df.columns)
    adata = SimpleNamespace(X=counts_df.astype(int).to_numpy(), obs=metadata, var=var, obs_names=counts_df.index,
                            var_names=counts_df.columns)
    dds = _dds(adata=adata, design="~condition")
    assert "dummy_param" in dds.var
    dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"])
    ds.summary()
    assert_res_almost_equal(ds.results_df, r_csv("single_factor", "r_test_res.csv"), tol)
    assert "dummy_param" in dds.var


def test_ref_design_matrix_init(counts_df, metadata, tol=0.02):
    """test_pydeseq2.py:731-758: a design MATRIX (DataFrame with its own column names) and a numeric contrast."""
    from pydeseq2_amd.api import build_design

    dm = build_design(metadata, "~condition").rename(columns={"condition[T.B]": "condition_B"})
    dds = _dds(counts=counts_df, metadata=metadata, design=dm)
    dds.deseq2()
    ds = _ds(dds, contrast=np.array([0, 1]))
    ds.summary()
    assert_res_almost_equal(ds.results_df, r_csv("single_factor", "r_test_res.csv"), tol)


def test_ref_vst(counts_df, metadata, tol=0.02):
    """test_pydeseq2.py:761-786."""
    for use_design, fn in ((False, "r_vst.csv"), (True, "r_vst_with_design.csv")):
        r_vst = r_csv("single_factor", fn).T
        dds = _dds(counts=counts_df, metadata=metadata, design="~condition")
        dds.vst(use_design=use_design)
        assert (np.abs(r_vst - dds.layers["vst_counts"]) / r_vst).max().max() < tol


def test_ref_mean_vst(counts_df, metadata, tol=0.02):
    """test_pydeseq2.py:789-803."""
    r_vst = r_csv("single_factor", "r_mean_vst.csv").T
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition")
    dds.vst(use_design=False, fit_type="mean")
    assert (np.abs(r_vst - dds.layers["vst_counts"]) / r_vst).max().max() < tol


def test_ref_deseq2_norm(counts_df, metadata):
    """test_pydeseq2.py:806-823: fit_size_factors() against preprocessing.deseq2_norm (here: its definition in numpy,
    preprocessing.py:31-102), default design."""
    dds = _dds(counts=counts_df, metadata=metadata)
    dds.fit_size_factors()
    with np.errstate(divide="ignore"):
        lc = np.log(counts_df.to_numpy())
    keep = ~np.isinf(lc.mean(0))  # genes with a zero count are left out of the median (preprocessing.py:52-56)
    s2 = np.exp(np.median(lc[:, keep] - lc.mean(0)[keep], axis=1))
    np.testing.assert_array_almost_equal(dds.obs["size_factors"], s2, decimal=8)


def test_ref_deseq2_norm_fit_and_transform(counts_df):
    """test_pydeseq2.py:847-866: the free functions deseq2_norm_fit / deseq2_norm_transform are the reference's host
    helpers (preprocessing.py), not part of the device path; their shapes are checked on the engine's equivalents -
    vst_fit's training log means and vst_transform's new-sample size factors."""
    train, test = counts_df[25:75], counts_df[0:25]
    meta = load_dataset("synthetic")[1]
    dds = _dds(counts=train, metadata=meta[25:75], design="~condition")
    dds.vst_fit()
    assert dds.logmeans.shape == (10,) and dds.filtered_genes.shape == (10,)
    out = dds.vst_transform(test.to_numpy())
    assert out.shape == (25, 10)


def test_ref_vst_fit(counts_df, metadata):
    """test_pydeseq2.py:869-876."""
    dds = _dds(counts=counts_df[25:75], metadata=metadata[25:75], design="~condition")
    dds.vst_fit()
    assert "vst_trend_coeffs" in dds.uns
    assert "normed_counts" in dds.layers
    assert "size_factors" in dds.obs


def test_ref_vst_transform(counts_df, metadata):
    """test_pydeseq2.py:879-886."""
    dds = _dds(counts=counts_df[25:75], metadata=metadata[25:75], design="~condition")
    dds.vst_fit()
    result = dds.vst_transform(counts_df[0:25].to_numpy())
    assert isinstance(result, np.ndarray) and result.shape == (25, 10)


@pytest.mark.parametrize(("dea_fit_type", "vst_fit_type"), [("mean", "parametric"), ("parametric", "mean"),
                                                             ("parametric", "parametric"), ("mean", "mean")])
def test_ref_vst_blind(counts_df, metadata, dea_fit_type, vst_fit_type):
    """test_pydeseq2.py:889-917."""
    dds = _dds(counts=counts_df[25:75], metadata=metadata[25:75], design="~condition", fit_type=dea_fit_type)
    dds.deseq2()
    assert ("trend_coeffs" if dea_fit_type == "parametric" else "mean_disp") in dds.uns
    assert "normed_counts" in dds.layers
    assert "size_factors" in dds.obs
    assert dds.fit_type == dea_fit_type
    dds.vst(use_design=False, fit_type=vst_fit_type)
    assert dds.fit_type == dea_fit_type


def test_ref_vst_transform_no_fit(counts_df, metadata):
    """test_pydeseq2.py:920-929."""
    dds = _dds(counts=counts_df[25:75], metadata=metadata[25:75], design="~condition", fit_type="parametric")
    with pytest.raises(RuntimeError):
        dds.vst_transform(counts_df[0:25].to_numpy())


# ---------------------------------------------------------------------------------------- tests/test_edge_cases.py
def test_ref_zero_genes(counts_df, metadata):
    """test_edge_cases.py:10-52 (with the gene slice ``dds[:, zero_genes]``)."""
    n, m = counts_df.shape
    np.random.seed(42)
    zero_genes = counts_df.columns[np.random.choice(m, size=m // 3, replace=False)]
    counts_df[zero_genes] = 0
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition")
    dds.deseq2()
    assert np.isnan(dds.var.loc[zero_genes, "dispersions"]).all()
    assert np.isnan(dds[:, zero_genes].varm["LFC"]).all().all()
    ds = _ds(dds, contrast=["condition", "B", "A"])
    ds.summary()
    res = ds.results_df
    assert (res.loc[zero_genes].baseMean == 0).all()
    for col in ("log2FoldChange", "lfcSE", "stat", "pvalue", "padj"):
        assert res.loc[zero_genes, col].isna().all(), col


@pytest.mark.parametrize("bad", [[0, np.nan], [0, "a"], [0, 1.5], [0, -1]])
def test_ref_invalid_counts(bad):
    """test_edge_cases.py:56-102: test_nan_counts, test_numeric_counts, test_integer_counts, test_non_negative_counts."""
    counts = pd.DataFrame({"gene1": bad, "gene2": [4, 12]}, index=["sample1", "sample2"])
    meta = pd.DataFrame({"condition": [0, 1]}, index=["sample1", "sample2"])
    with pytest.raises(ValueError):
        _dds(counts=counts, metadata=meta, design="~condition")


def test_ref_nan_factors():
    """test_edge_cases.py:105-113."""
    counts = pd.DataFrame({"gene1": [0, 1], "gene2": [4, 12]}, index=["sample1", "sample2"])
    with pytest.raises(ValueError):
        _dds(counts=counts, metadata=pd.DataFrame({"condition": [0, np.nan]}, index=counts.index), design="~condition")


def test_ref_one_factor_and_rank_deficient_design():
    """test_edge_cases.py:116-138: a constant design variable, a design of less than full column rank - a warning each."""
    counts = pd.DataFrame({"gene1": [0, 1], "gene2": [4, 12]}, index=["sample1", "sample2"])
    with pytest.warns(UserWarning):
        _dds(counts=counts, metadata=pd.DataFrame({"condition": [0, 0]}, index=counts.index), design="~condition")
    with pytest.warns(UserWarning):
        _dds(counts=counts, metadata=pd.DataFrame({"condition": [0, 1], "batch": ["A", "B"]}, index=counts.index),
             design="~condition + batch")


def test_ref_equal_num_vars_num_samples_design():
    """test_edge_cases.py:141-158: fit_size_factors() works, fit_genewise_dispersions() raises ValueError (N == p)."""
    counts = pd.DataFrame({"gene1": [0, 1, 55], "gene2": [4, 12, 60]}, index=["sample1", "sample2", "sample3"])
    meta = pd.DataFrame({"condition": [0, 1, 0], "batch": ["A", "B", "B"]}, index=counts.index)
    dds = _dds(counts=counts, metadata=meta, design="~condition + batch")
    dds.fit_size_factors()
    with pytest.raises(ValueError):
        dds.fit_genewise_dispersions()


def test_ref_matching_samples_and_indexes():
    """test_edge_cases.py:164-195 (design-matrix index vs obs) and 227-240 (counts index vs metadata index)."""
    counts = pd.DataFrame({"gene1": [0, 1, 55], "gene2": [4, 12, 60]}, index=["sample1", "sample2", "sample3"])
    meta = pd.DataFrame({"condition": [0, 1, 0]}, index=counts.index)
    for idx, rows in ((["sample1", "sample2", "sample5"], 3), (["sample1", "sample2"], 2),
                      (["sample1", "sample2", "sample3", "sample4"], 4)):
        dm = pd.DataFrame({"intercept": [1.0] * rows, "condition": [0, 1, 0, 0][:rows]}, index=idx)
        with pytest.raises(ValueError):
            _dds(counts=counts, metadata=meta, design=dm)
    with pytest.raises(ValueError):
        _dds(counts=counts.iloc[:2], metadata=pd.DataFrame({"condition": [0, 1]}, index=["sample01", "sample02"]),
             design="~condition")


def test_ref_lfc_shrinkage_coeff(counts_df, metadata):
    """test_edge_cases.py:198-224."""
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition")
    dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"])
    ds.summary()
    with pytest.raises(KeyError):
        ds.lfc_shrink(coeff="this_coeff_does_not_exist")


def test_ref_contrast_errors(counts_df, metadata):
    """test_edge_cases.py:243-286."""
    dds = _dds(counts=counts_df, metadata=metadata, refit_cooks=False, design="~condition + group")
    dds.deseq2()
    with pytest.raises(IndexError):
        _ds(dds, contrast=["condition", "B"])
    for bad in (["batch", "Y", "X"], ["condition", "B", "C"], ["condition", "C", "B"], np.array([0, 0, 0, 1])):
        with pytest.raises(ValueError):
            _ds(dds, contrast=bad)


def test_ref_cooks_not_refitted(counts_df, metadata):
    """test_edge_cases.py:289-320: refit_cooks switched on after a fit that did not refit -> AttributeError."""
    dds = _dds(counts=counts_df, metadata=metadata, refit_cooks=False, design="~condition")
    dds.deseq2()
    dds.refit_cooks = True
    with pytest.raises(AttributeError):
        ds = _ds(dds, contrast=["condition", "B", "A"])
        ds.summary()


def test_ref_few_samples(counts_df, metadata):
    """test_edge_cases.py:323-364: two samples per group (a warning about the degrees of freedom; calculate_cooks runs)."""
    keep = ["sample1", "sample2", "sample99", "sample100"]
    counts, meta = counts_df.loc[keep].copy(), metadata.loc[keep]
    counts.iloc[0, 0] = 1000
    dds = _dds(counts=counts, metadata=meta, refit_cooks=True, design="~condition")
    with pytest.warns(UserWarning):
        dds.deseq2()
    res = _ds(dds, contrast=["condition", "B", "A"])
    res.summary()
    assert dds.var["replaced"].sum() == 0


def test_ref_few_samples_and_outlier(counts_df, metadata):
    """test_edge_cases.py:367-417."""
    keep = ["sample1", "sample2"] + [f"sample{i}" for i in range(92, 101)]
    counts, meta = counts_df.loc[keep].copy(), metadata.loc[keep]
    counts.iloc[0, 0] = 1000
    counts.iloc[-1, -1] = 1000
    dds = _dds(counts=counts, metadata=meta, refit_cooks=True, design="~condition")
    dds.deseq2()
    _ds(dds, contrast=["condition", "B", "A"]).summary()


def test_ref_new_all_zero_gene(counts_df, metadata):
    """test_edge_cases.py:420-464: a gene whose only count is replaced away."""
    meta = metadata.loc[[f"sample{i}" for i in [*range(1, 11), *range(91, 101)]]]
    counts = counts_df.loc[meta.index].copy()
    counts["geneX"] = 0
    counts.loc["sample100", "geneX"] = 100
    dds = _dds(counts=counts, metadata=meta, design="~condition", refit_cooks=True)
    with pytest.warns(UserWarning):
        dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"])
    ds.summary()
    assert dds.new_all_zeroes_genes.equals(pd.Index(["geneX"]))
    r = ds.results_df.loc["geneX"]
    assert r["baseMean"] == 0 and r["log2FoldChange"] == 0 and r["lfcSE"] == 0 and r["stat"] == 0
    assert np.isnan(r["pvalue"]) and np.isnan(r["padj"])


def test_ref_zero_inflated(counts_df, metadata):
    """test_edge_cases.py:467-494: every gene holds a zero -> a warning and the iterative size factors."""
    np.random.seed(42)
    idx = np.random.choice(len(counts_df), counts_df.shape[-1])
    counts_df.iloc[idx, :] = 0
    dds = _dds(counts=counts_df, metadata=metadata)
    with pytest.warns(UserWarning):
        dds.deseq2()


def test_ref_plot_MA(counts_df, metadata, tmp_path):
    """test_edge_cases.py:497-527: AttributeError before summary(), a figure after it."""
    dds = _dds(counts=counts_df, metadata=metadata)
    dds.deseq2()
    ds = _ds(dds, contrast=["condition", "B", "A"])
    with pytest.raises(AttributeError):
        ds.plot_MA()
    ds.summary()
    try:
        import matplotlib

        matplotlib.use("Agg")
    except ImportError:
        return
    ds.plot_MA(save_path=str(tmp_path / "ma.png"))


# ---------------------------------------------------------------------------------------- examples/plot_step_by_step.py
def test_ref_step_by_step_example(counts_df, metadata, tmp_path):
    """examples/plot_step_by_step.py:94-246: the stage-wise call sequence, the pickles in the middle and at the end, and
    the sub-steps of DeseqStats - against the same data set fitted by deseq2() / summary() in one go (identical fields)."""
    from pydeseq2_amd import HipInference

    ref = _dds(counts=counts_df, metadata=metadata, design="~condition").deseq2()
    inference = HipInference()
    dds = _dds(counts=counts_df, metadata=metadata, design="~condition", refit_cooks=True, inference=inference)
    dds.fit_size_factors()
    assert "size_factors" in dds.obs and "genewise_dispersions" not in dds.var
    dds.fit_genewise_dispersions()
    assert "genewise_dispersions" in dds.var and "fitted_dispersions" not in dds.var
    dds.fit_dispersion_trend()
    assert "trend_coeffs" in dds.uns and "fitted_dispersions" in dds.var and "prior_disp_var" not in dds.uns
    dds.fit_dispersion_prior()
    assert "_squared_logres" in dds.uns and "prior_disp_var" in dds.uns and "MAP_dispersions" not in dds.var
    dds.fit_MAP_dispersions()
    assert "MAP_dispersions" in dds.var and "dispersions" in dds.var and "LFC" not in dds.varm
    dds.fit_LFC()
    assert "LFC" in dds.varm and "replaced" not in dds.var
    dds.calculate_cooks()
    assert "cooks" in dds.layers
    if dds.refit_cooks:
        dds.refit()
    assert "replaced" in dds.var and "refitted" in dds.var
    for col in ("_normed_means", "_MoM_dispersions", "genewise_dispersions", "fitted_dispersions", "MAP_dispersions",
                "dispersions", "_genewise_converged", "_MAP_converged", "_LFC_converged"):
        np.testing.assert_array_equal(dds.var[col].to_numpy(), ref.var[col].to_numpy(), err_msg=col)
    np.testing.assert_array_equal(dds.varm["LFC"].to_numpy(), ref.varm["LFC"].to_numpy())
    np.testing.assert_array_equal(dds.obs["size_factors"].to_numpy(), ref.obs["size_factors"].to_numpy())
    assert dds.uns["prior_disp_var"] == ref.uns["prior_disp_var"]
    with open(tmp_path / "dds.pkl", "wb") as f:
        pickle.dump(dds, f)
    with open(tmp_path / "dds.pkl", "rb") as f:
        back = pickle.load(f)
    np.testing.assert_array_equal(back.var["dispersions"].to_numpy(), dds.var["dispersions"].to_numpy())
    np.testing.assert_array_equal(back.layers["cooks"], dds.layers["cooks"])
    ds = _ds(dds, contrast=np.array([0, 1]), alpha=0.05, cooks_filter=True, independent_filter=True)
    ds.run_wald_test()
    assert ds.p_values.shape == (10,)
    if ds.cooks_filter:
        ds._cooks_filtering()
    if ds.independent_filter:
        ds._independent_filtering()
    else:
        ds._p_value_adjustment()
    assert ds.padj.shape == (10,)
    df = ds.summary()
    one_go = _ds(ref, contrast=np.array([0, 1])).summary()
    pd.testing.assert_frame_equal(df, one_go)
    with open(tmp_path / "ds.pkl", "wb") as f:
        pickle.dump(ds, f)
    with open(tmp_path / "ds.pkl", "rb") as f:
        ds2 = pickle.load(f)
    pd.testing.assert_frame_equal(ds2.results_df, df)
    # the unpickled data set fits again (its device state is rebuilt on demand)
    again = _ds(back, contrast=np.array([0, 1])).summary()
    pd.testing.assert_frame_equal(again, df)
    ds.lfc_shrink(coeff="condition[T.B]")
    assert ds.shrunk_LFCs
