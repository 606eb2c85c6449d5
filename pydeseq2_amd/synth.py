"""Synthetic negative-binomial count matrices for benchmarks and tests (SURVEY.md §8(d) generator).

Pure numpy, no device code: `bench.py` and the tests build their inputs with it so that the workload
definition lives outside `oracle/` (the oracle is the checker only).
"""
from __future__ import annotations

import numpy as np


def make_design(kind: str, N: int, rng: np.random.Generator) -> np.ndarray:
    """Design matrices for the benchmark configs (intercept + treatment-coded factors)."""
    def bal(levels):
        v = np.arange(N) % levels
        rng.shuffle(v)
        return v

    def dummies(v, levels):
        return np.stack([(v == k).astype(float) for k in range(1, levels)], axis=1)

    cols = [np.ones((N, 1))]
    if kind == "2level":
        cols.append(dummies(np.arange(N) % 2, 2))
    elif kind == "3factor":  # 2/3/5 levels -> p = 8
        for lv in (2, 3, 5):
            cols.append(dummies(bal(lv), lv))
    elif kind == "2factor":  # 2/3 levels -> p = 4, 6 cells (developer measurements: condition + batch)
        for lv in (2, 3):
            cols.append(dummies(bal(lv), lv))
    elif kind == "mixed":  # 2 + 4 levels + 3 continuous -> p = 8
        for lv in (2, 4):
            cols.append(dummies(bal(lv), lv))
        cols.append(rng.normal(size=(N, 3)))
    else:
        raise KeyError(kind)
    return np.concatenate(cols, axis=1)


def synth_counts(G: int, N: int, design: str = "2level", seed: int = 0):
    """Synthetic NB counts as specified in SURVEY.md §8(d).  Returns (counts int64 N x G, X)."""
    rng = np.random.default_rng(seed)
    X = make_design(design, N, rng)
    p = X.shape[1]
    beta = np.zeros((p, G))
    beta[0] = rng.normal(4, 2, G)
    beta[1] = rng.normal(0, 1, G) * (rng.random(G) < 0.3)
    for j in range(2, p):
        cont = design == "mixed" and j >= p - 3
        beta[j] = rng.normal(0, 0.2 if cont else 0.5, G)
    disp = 4 / np.maximum(2.0 ** beta[0], 1e-3) + 0.1
    sf = np.exp(rng.normal(0, 0.2, N))
    mu = sf[:, None] * 2.0 ** (X @ beta)
    size = 1 / disp
    counts = rng.negative_binomial(size[None, :], size[None, :] / (size[None, :] + mu))
    return counts.astype(np.int64), X
