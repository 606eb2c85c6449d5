"""Edge cases the reference tests (tests/test_edge_cases.py): all-zero genes, invalid inputs,
cohorts with < 3 replicates, a gene that becomes all-zero after outlier replacement.  The oracle
is checked on CPU; the HIP engine is checked against the oracle on the GPU (-m gpu)."""
import warnings

import numpy as np
import pytest

from oracle import nbglm_oracle as orc
from tests.helpers import assert_close, load_dataset, treatment_design


def _few_samples(rows, outliers):
    counts, meta = load_dataset("synthetic")
    keep = [f"sample{i}" for i in rows]
    counts, meta = counts.loc[keep].copy(), meta.loc[keep].copy()
    for (i, j) in outliers:
        counts.iloc[i, j] = 1000
    X, _ = treatment_design(meta, ["condition"])
    return counts.to_numpy(), X


def _new_all_zero():
    counts, meta = load_dataset("synthetic")
    keep = [f"sample{i}" for i in [*range(1, 11), *range(91, 101)]]
    counts, meta = counts.loc[keep].copy(), meta.loc[keep].copy()
    counts["geneX"] = 0
    counts.loc["sample100", "geneX"] = 100
    X, _ = treatment_design(meta, ["condition"])
    return counts.to_numpy(), X


CASES = {
    "few_samples": lambda: _few_samples([1, 2, 99, 100], [(0, 0)]),
    "few_samples_outlier": lambda: _few_samples([1, 2, 92, 93, 94, 95, 96, 97, 98, 99, 100], [(0, 0), (-1, -1)]),
    "new_all_zero": _new_all_zero,
}


def test_oracle_few_samples_no_refit():
    counts, X = CASES["few_samples"]()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = orc.deseq2(counts, X)
    assert res.replaced.sum() == 0  # tests/test_edge_cases.py:363


def test_oracle_new_all_zero_gene():
    counts, X = CASES["new_all_zero"]()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = orc.deseq2(counts, X)
    g = counts.shape[1] - 1
    assert res.new_all_zeroes[g] and res.new_all_zeroes.sum() == 1
    assert res.normed_means[g] == 0 and (res.LFC[g] == 0).all() and res.lfcSE[g] == 0 and res.stat[g] == 0
    assert res.cooks_outlier[g]  # -> p-value set to NaN by the Cook's filter (ds.py:544-550)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_edge_case_matches_oracle(case):
    import pydeseq2_amd

    counts, X = CASES[case]()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.deseq2(counts, X)
        res = pydeseq2_amd.deseq2(counts, X, device=0)
    assert res.disp_function_type == ref.disp_function_type
    assert (res.replaced == ref.replaced).all() and (res.refitted == ref.refitted).all()
    assert (res.new_all_zeroes == ref.new_all_zeroes).all()
    assert (res.cooks_outlier == ref.cooks_outlier).all()
    assert_close(res.size_factors, ref.size_factors, 1e-12, 0, "sf")
    assert_close(res.normed_means, ref.normed_means, 1e-12, 0, "means")
    assert_close(res.dispersions, ref.dispersions, 1e-5, 0, "dispersions")
    assert_close(res.LFC, ref.LFC, 1e-5, 1e-8, "LFC")
    assert_close(res.lfcSE, ref.lfcSE, 1e-5, 0, "SE")
    assert_close(res.pvalue, ref.pvalue, 2e-5, 1e-300, "p")


@pytest.mark.gpu
def test_gpu_invalid_inputs_raise():
    import pydeseq2_amd

    counts, X = CASES["few_samples_outlier"]()
    bad = counts.astype(float)
    bad[0, 0] = np.nan
    with pytest.raises(ValueError):
        pydeseq2_amd.DeseqPipeline(bad, X, device=0)
    bad = counts.astype(float)
    bad[0, 0] = 1.5
    with pytest.raises(ValueError):
        pydeseq2_amd.DeseqPipeline(bad, X, device=0)
    neg = counts.copy()
    neg[0, 0] = -1
    with pytest.raises(ValueError):
        pydeseq2_amd.DeseqPipeline(neg, X, device=0)
    Xn = X.copy()
    Xn[0, 1] = np.nan
    with pytest.raises(ValueError):
        pydeseq2_amd.DeseqPipeline(counts, Xn, device=0)
    big = counts.astype(np.int64)
    big[1, 2] = 2**31  # documented limit: counts live as int32 on the device (the reference holds int64, dds.py:245-249)
    with pytest.raises(ValueError, match="below 2\\^31"):
        pydeseq2_amd.DeseqPipeline(big, X, device=0)
    from pydeseq2_amd import HipInference

    with pytest.raises(Exception, match="2\\^31"):
        HipInference(device=0).lin_reg_mu(big, np.ones(len(big)), X, 0.5)
    with pytest.raises(ValueError):  # N == p: no replicates (utils.py:839-844)
        pydeseq2_amd.deseq2(counts[:2], X[:2] + np.array([[0, 0], [0, 1.0]]) * 0 + np.eye(2), device=0)


@pytest.mark.gpu
def test_input_dtypes_and_layouts_give_identical_results():
    """int64 / int32 / integer-valued float64 counts, C- or F-ordered: the same bits come out (the
    reference casts with ``.astype(int)``, dds.py:245-249)."""
    import pydeseq2_amd
    from oracle import nbglm_oracle as orc

    counts, X = orc.synth_counts(300, 40, "2level", 17)
    base = pydeseq2_amd.deseq2(counts, X, device=0)
    for variant in (counts.astype(np.int32), counts.astype(np.float64), np.asfortranarray(counts),
                    np.asfortranarray(counts.astype(np.int32))):
        r = pydeseq2_amd.deseq2(variant, X, device=0)
        for f in ("size_factors", "dispersions", "LFC", "pvalue"):
            assert np.array_equal(getattr(r, f), getattr(base, f), equal_nan=True), (variant.dtype, f)
    with pytest.raises(ValueError):
        pydeseq2_amd.deseq2(counts + 0.5, X, device=0)
