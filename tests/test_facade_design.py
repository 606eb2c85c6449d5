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
