"""Mean-based dispersion trend (host side, O(G)).

Replaces DeseqDataSet._fit_mean_dispersion_trend (dds.py:1277-1299): the fallback when the parametric trend
(device kernels k_trend_fit / k_trend_fit_grid, csrc/dsq_trend.h) does not converge, and fit_type="mean".
The parametric trend and the MAD prior run on the device (dsq_dev_trend_prior).
"""
from __future__ import annotations

import numpy as np
from scipy.stats import trim_mean


def mean_trend(genewise_all: np.ndarray, min_disp: float) -> float:
    """Trimmed-mean trend (dds.py:1288-1293); NaN entries (all-zero genes) are ignored."""
    with np.errstate(invalid="ignore"):
        sel = genewise_all[genewise_all > 10 * min_disp]
    return float(trim_mean(sel, proportiontocut=0.001))
