"""Shared test helpers (fixture loading, treatment-coded designs, comparisons)."""
import os

import numpy as np
import pandas as pd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kat(name):
    return dict(np.load(os.path.join(GOLD, f"kat_{name}.npz"), allow_pickle=False))


def load_dataset(which):
    """Return (counts DataFrame samples x genes, metadata DataFrame)."""
    d = {"synthetic": "synthetic", "continuous": "r_continuous", "wide": "r_wide"}[which]
    counts = pd.read_csv(os.path.join(GOLD, d, "test_counts.csv"), index_col=0).T
    meta = pd.read_csv(os.path.join(GOLD, d, "test_metadata.csv"), index_col=0)
    return counts, meta


def r_csv(sub, name):
    return pd.read_csv(os.path.join(GOLD, f"r_{sub}", name), index_col=0)


def treatment_design(meta, factors, continuous=()):
    """Intercept + treatment-coded (first sorted level = reference) factor columns.

    Same column order formulaic gives for "~f1 + f2 + x": Intercept, f1[T.*], f2[T.*], x.
    Returns (X ndarray, column names).
    """
    cols, names = [np.ones(len(meta))], ["Intercept"]
    for f in factors:
        levels = sorted(meta[f].unique())
        for lv in levels[1:]:
            cols.append((meta[f] == lv).to_numpy().astype(float))
            names.append(f"{f}[T.{lv}]")
    for c in continuous:
        cols.append(meta[c].to_numpy().astype(float))
        names.append(c)
    return np.column_stack(cols), names


def max_rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    ok = ~np.isnan(b)
    assert (np.isnan(a) == np.isnan(b)).all(), "NaN pattern differs"
    return float(np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))) if ok.any() else 0.0


def assert_close(a, b, rtol, atol=0.0, what=""):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert (np.isnan(a) == np.isnan(b)).all(), f"{what}: NaN pattern differs"
    inf = np.isinf(b)
    assert (a[inf] == b[inf]).all() and (np.isinf(a) == inf).all(), f"{what}: inf pattern differs"
    ok = ~np.isnan(b) & ~inf
    err = np.abs(a[ok] - b[ok]) - (atol + rtol * np.abs(b[ok]))
    assert (err <= 0).all(), (
        f"{what}: max excess {err.max():.3e}; worst rel "
        f"{np.max(np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1e-300)):.3e}"
    )
