"""Formula -> design matrix of the AnnData-free façade (pydeseq2_amd/api.py), CPU only."""
import numpy as np
import pandas as pd
import pytest

from pydeseq2_amd.api import build_design
from tests.helpers import load_dataset, treatment_design


@pytest.mark.parametrize("which,factors,cont,formula", [
    ("synthetic", ["condition"], (), "~condition"),
    ("synthetic", ["group", "condition"], (), "~group + condition"),
    ("continuous", ["group", "condition"], ("measurement",), "~ group + condition + measurement"),
])
def test_formula_designs(which, factors, cont, formula):
    _, meta = load_dataset(which)
    X, names = treatment_design(meta, factors, cont)
    dm = build_design(meta, formula)
    assert list(dm.columns) == names
    assert np.array_equal(dm.to_numpy(), X)


def test_formula_errors_and_matrix_input():
    meta = pd.DataFrame({"a": ["x", "y", "x"], "b": [1.0, 2.0, np.nan]})
    with pytest.raises(KeyError):
        build_design(meta, "~zzz")
    with pytest.raises(ValueError):
        build_design(meta, "~b")
    with pytest.raises(NotImplementedError):
        build_design(meta, "~log(b)")
    with pytest.raises(ValueError):
        build_design(meta, "a + b")
    assert list(build_design(meta, "~0 + a").columns) == ["a[T.y]"]
    M = np.ones((3, 2))
    assert build_design(meta, M).shape == (3, 2)


def test_ref_level_and_design_factors():
    meta = pd.DataFrame({"condition": ["A", "B", "C", "A", "B", "C"], "x": [1, 2, 3, 4, 5, 6]})
    dm = build_design(meta, "~condition", ref_level=["condition", "B"])
    assert list(dm.columns) == ["Intercept", "condition[T.A]", "condition[T.C]"]
    assert dm["condition[T.A]"].tolist() == [1, 0, 0, 1, 0, 0]
    with pytest.raises(KeyError):
        build_design(meta, "~condition", ref_level=["condition", "Z"])
    assert list(build_design(meta, "~condition + x").columns) == ["Intercept", "condition[T.B]", "condition[T.C]", "x"]


def test_interactions():
    meta = pd.DataFrame({"g": list("XXYYXXYY"), "c": list("ABABABAB"), "x": np.arange(8.0)})
    dm = build_design(meta, "~g*c")
    assert list(dm.columns) == ["Intercept", "g[T.Y]", "c[T.B]", "g[T.Y]:c[T.B]"]
    assert np.array_equal(dm["g[T.Y]:c[T.B]"], dm["g[T.Y]"] * dm["c[T.B]"])
    dm2 = build_design(meta, "~g + c + g:x")
    assert list(dm2.columns) == ["Intercept", "g[T.Y]", "c[T.B]", "g[T.Y]:x"]
    assert np.array_equal(dm2["g[T.Y]:x"], dm2["g[T.Y]"] * meta["x"])
    assert np.linalg.matrix_rank(dm.to_numpy()) == 4


def test_count_validation_on_the_host():
    """The reference's constructor checks (tests/test_edge_cases.py: test_nan_counts, test_numeric_counts,
    test_integer_counts, test_non_negative_counts) run before anything touches the GPU."""
    from pydeseq2_amd.api import DeseqDataSet, check_counts

    meta = pd.DataFrame({"condition": list("ABABAB")}, index=[f"s{i}" for i in range(6)])
    good = pd.DataFrame(np.arange(18).reshape(6, 3), index=meta.index, columns=list("xyz"))
    check_counts(good)
    check_counts(good.astype(float))
    check_counts(good.to_numpy())
    cases = {
        "NaNs are not allowed": good.astype(float).mask(good == 4),
        "only contain numbers": good.astype(str),
        "only contain integers": good + 0.5,
        "non-negative": good - 3,
    }
    for msg, bad in cases.items():
        with pytest.raises(ValueError, match=msg):
            DeseqDataSet(counts=bad, metadata=meta, design="~condition")
    with pytest.raises(ValueError, match="number of samples"):
        DeseqDataSet(counts=good.iloc[:4], metadata=meta, design="~condition")


def test_rank_deficient_design_warns():
    """tests/test_edge_cases.py::test_rank_deficient_design: a warning, not an error, at construction."""
    from pydeseq2_amd.api import DeseqDataSet

    meta = pd.DataFrame({"a": list("XXYYXXYY"), "b": list("PPQQPPQQ")}, index=[f"s{i}" for i in range(8)])
    counts = pd.DataFrame(np.arange(16).reshape(8, 2) + 1, index=meta.index, columns=["g1", "g2"])
    with pytest.warns(UserWarning, match="not full rank"):
        try:
            DeseqDataSet(counts=counts, metadata=meta, design="~a + b")
        except Exception:  # noqa: BLE001 - without a GPU the pipeline behind the facade cannot be created
            pass


def test_sample_indexes_must_match():
    """tests/test_edge_cases.py::test_indexes and ::test_matching_samples."""
    from pydeseq2_amd.api import DeseqDataSet

    counts = pd.DataFrame({"gene1": [0, 1, 55], "gene2": [4, 12, 60]}, index=["sample1", "sample2", "sample3"])
    meta = pd.DataFrame({"condition": [0, 1, 0]}, index=["sample1", "sample2", "sample3"])
    with pytest.raises(ValueError):
        DeseqDataSet(counts=counts, metadata=meta.set_axis(["sample01", "sample02", "sample03"]), design="~condition")
    for idx, rows in ((["sample1", "sample2", "sample5"], 3), (["sample1", "sample2"], 2),
                      (["sample1", "sample2", "sample3", "sample4"], 4)):
        dm = pd.DataFrame({"intercept": [1.0] * rows, "condition": [0, 1, 0, 0][:rows]}, index=idx)
        with pytest.raises(ValueError):
            DeseqDataSet(counts=counts, metadata=meta, design=dm)


def test_contrast_validation():
    """tests/test_edge_cases.py::test_contrast on a stand-in for a fitted dataset (the checks are host logic)."""
    from types import SimpleNamespace

    from pydeseq2_amd.api import DeseqStats, build_design

    meta = pd.DataFrame({"condition": list("ABABAB"), "group": list("XXYYXY")}, index=[f"s{i}" for i in range(6)])
    dm = build_design(meta, "~condition + group")
    genes = pd.Index(["g1", "g2"])
    dds = SimpleNamespace(_res=object(), obs=meta, obsm={"design_matrix": dm}, var_names=genes,
                          varm={"LFC": pd.DataFrame(np.zeros((2, 3)), index=genes, columns=dm.columns)},
                          var=pd.DataFrame({"_normed_means": [1.0, 2.0]}, index=genes))
    ds = DeseqStats(dds, contrast=["condition", "B", "A"])
    assert ds.contrast_vector.tolist() == [0.0, 1.0, 0.0]
    assert DeseqStats(dds, contrast=["condition", "A", "B"]).contrast_vector.tolist() == [0.0, -1.0, 0.0]
    assert DeseqStats(dds, contrast=np.array([0.0, 0.0, 1.0])).contrast_vector.tolist() == [0.0, 0.0, 1.0]
    with pytest.raises(IndexError):
        DeseqStats(dds, contrast=["condition", "B"])
    for bad in (["batch", "Y", "X"], ["condition", "B", "C"], ["condition", "C", "B"], np.array([0, 0, 0, 1])):
        with pytest.raises(ValueError):
            DeseqStats(dds, contrast=bad)
    with pytest.raises(ValueError):
        DeseqStats(dds, contrast=None)
    with pytest.raises(AttributeError):
        DeseqStats(SimpleNamespace(_res=None), contrast=["condition", "B", "A"])


def test_documented_limits_are_refused_with_a_clear_error():
    """DESIGN.md 7 (limits the reference does not have): more than 32 design columns (utils.py:345-371 takes any width) is
    refused when the design is packed - before any GPU work - with a message that names the limit."""
    from pydeseq2_amd._design import MAX_DESIGN_COLUMNS, DesignPack

    rng = np.random.default_rng(0)
    X = np.column_stack([np.ones(200)] + [rng.normal(size=200) for _ in range(MAX_DESIGN_COLUMNS)])
    with pytest.raises(ValueError, match=f"at most {MAX_DESIGN_COLUMNS}"):
        DesignPack(X)
    assert DesignPack(X[:, :MAX_DESIGN_COLUMNS]).P == MAX_DESIGN_COLUMNS


def test_construction_slicing_copy_and_pickling_never_touch_the_gpu():
    """The device pipeline behind the facade is created on first use: building a data set, dds[:, genes] (dds.py:868-873,
    1330), copy() and pickling (examples/plot_step_by_step.py:175-177) are host operations."""
    import pickle

    from pydeseq2_amd.api import DeseqDataSet

    meta = pd.DataFrame({"condition": list("ABABAB"), "x": np.arange(6.0)}, index=[f"s{i}" for i in range(6)])
    counts = pd.DataFrame(np.arange(24).reshape(6, 4), index=meta.index, columns=list("wxyz"))
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~condition + x", low_memory=True)
    assert dds._pipe_obj is None and dds.n_obs == 6 and dds.n_vars == 4 and dds.low_memory
    dds.var["note"] = [1, 2, 3, 4]
    for key, names in ((["x", "z"], ["x", "z"]), (np.array([False, True, False, True]), ["x", "z"]), ([1, 3], ["x", "z"]),
                       (slice(1, 3), ["x", "y"]), ("w", ["w"])):
        sub = dds[:, key]
        assert list(sub.var_names) == names and sub.X.shape == (6, len(names)) and list(sub.var["note"]) == \
            [dds.var.loc[n, "note"] for n in names]
    sub = dds[["s1", "s4"], ["w"]]
    assert sub.X.tolist() == [[4], [16]] and list(sub.obs_names) == ["s1", "s4"] and sub.obsm["design_matrix"].shape == (2, 3)
    with pytest.raises(KeyError):
        dds[:, ["nope"]]
    cp = dds.copy()
    cp.X[0, 0] = 99
    cp.var["note"] = 0
    assert dds.X[0, 0] == 0 and list(dds.var["note"]) == [1, 2, 3, 4]
    back = pickle.loads(pickle.dumps(dds))
    assert back._pipe_obj is None and np.array_equal(back.X, dds.X) and back.obs.equals(dds.obs) and back.var.equals(dds.var)
    assert list(back.obsm["design_matrix"].columns) == ["Intercept", "condition[T.B]", "x"] and back.low_memory
    with pytest.raises(ValueError, match="below 2\\^31"):
        DeseqDataSet(counts=counts * 2 ** 29, metadata=meta, design="~condition")
    with pytest.raises(ValueError):
        DeseqDataSet(metadata=meta, design="~condition")


def test_cond_contrast_and_variables():
    """dds.variables / cond() / contrast() (dds.py:339-347, 564-582; the reference delegates to formulaic_contrasts): the design
    row of a condition with the unnamed variables at their reference level, and pairwise contrasts as differences of two."""
    from pydeseq2_amd.api import DeseqDataSet

    meta = pd.DataFrame({"condition": list("ABABAB"), "group": list("XXYYXY"), "x": np.arange(6.0)},
                        index=[f"s{i}" for i in range(6)])
    counts = pd.DataFrame(np.arange(24).reshape(6, 4), index=meta.index, columns=list("wxyz"))
    dds = DeseqDataSet(counts=counts, metadata=meta, design="~group + condition + x")
    assert dds.variables == ["group", "condition", "x"]
    assert dds.cond(condition="B").tolist() == [1, 0, 1, 0] and dds.cond(condition="A", group="Y").tolist() == [1, 1, 0, 0]
    assert dds.contrast("condition", "A", "B").tolist() == [0, 0, 1, 0] and dds.cond(x=2.5).tolist() == [1, 0, 0, 2.5]
    inter = DeseqDataSet(counts=counts, metadata=meta, design="~group*condition")
    assert inter.cond(group="Y", condition="B").tolist() == [1, 1, 1, 1] and inter.cond(group="Y").tolist() == [1, 1, 0, 0]
    relevel = DeseqDataSet(counts=counts, metadata=meta, design="~condition", ref_level=["condition", "B"])
    assert list(relevel.obsm["design_matrix"].columns) == ["Intercept", "condition[T.A]"]
    assert relevel.cond(condition="A").tolist() == [1, 1] and relevel.cond().tolist() == [1, 0]
    with pytest.raises(ValueError):
        dds.cond(batch="Q")
    with pytest.raises(ValueError):
        dds.cond(condition="Z")
    with pytest.raises(ValueError):
        DeseqDataSet(counts=counts, metadata=meta, design=np.ones((6, 1))).variables
