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


# ---------------------------------------------------------------------------------------------------------------
# Tiled variant: any (gene range x sample range) block of ONE well-defined matrix can be generated on its own, so a
# rank of a multi-GPU job builds its gene shard (and its sample block for the two-collective size factors) without
# ever holding the whole matrix - BASELINE configs[4] is 60 000 x 5 000 (2.4 GB as int64).  Same distributions as
# synth_counts; a different random stream (per-tile generators), hence a different matrix for the same seed.
GENE_TILE, SAMPLE_TILE = 500, 125


def synth_design(N: int, design: str, seed: int):
    """(X, size factors) of the tiled matrix: depend on (N, design, seed) only."""
    rng = np.random.default_rng([seed, 0])
    X = make_design(design, N, rng)
    return X, np.exp(rng.normal(0, 0.2, N))


def _gene_params(G: int, design: str, seed: int, p: int, block: int):
    """log2 coefficients [p x n] and dispersions of the genes of one GENE_TILE block."""
    n = min(GENE_TILE, G - block * GENE_TILE)
    rng = np.random.default_rng([seed, 1, block])
    beta = np.zeros((p, n))
    beta[0] = rng.normal(4, 2, n)
    beta[1] = rng.normal(0, 1, n) * (rng.random(n) < 0.3)
    for j in range(2, p):
        cont = design == "mixed" and j >= p - 3
        beta[j] = rng.normal(0, 0.2 if cont else 0.5, n)
    return beta, 4 / np.maximum(2.0 ** beta[0], 1e-3) + 0.1


def synth_counts_block(G: int, N: int, design: str, seed: int, genes=None, samples=None):
    """Counts [n1 - n0, g1 - g0] (int64) of the tiled matrix for genes [g0, g1) and samples [n0, n1), and X (all N
    samples).  Any two calls agree on their overlap."""
    g0, g1 = genes if genes is not None else (0, G)
    n0, n1 = samples if samples is not None else (0, N)
    X, sf = synth_design(N, design, seed)
    p = X.shape[1]
    out = np.empty((n1 - n0, g1 - g0), dtype=np.int64)
    for gb in range(g0 // GENE_TILE, (g1 + GENE_TILE - 1) // GENE_TILE):
        beta, disp = _gene_params(G, design, seed, p, gb)
        size = 1 / disp
        ga = gb * GENE_TILE
        lo, hi = max(g0, ga), min(g1, ga + beta.shape[1])
        for sb in range(n0 // SAMPLE_TILE, (n1 + SAMPLE_TILE - 1) // SAMPLE_TILE):
            sa = sb * SAMPLE_TILE
            se = min(N, sa + SAMPLE_TILE)
            mu = sf[sa:se, None] * 2.0 ** (X[sa:se] @ beta)
            tile = np.random.default_rng([seed, 2, gb, sb]).negative_binomial(size[None, :], size[None, :] / (size[None, :] + mu))
            a, b = max(n0, sa), min(n1, se)
            out[a - n0:b - n0, lo - g0:hi - g0] = tile[a - sa:b - sa, lo - ga:hi - ga]
    return out, X
