"""Host-side logic of the summary tail (pydeseq2_amd/summary.py): the vectorised lowess and the
cut-off choice, against vectors from the unmodified reference and against the oracle restatement."""
import os

import numpy as np

from oracle import nbglm_oracle as orc
from pydeseq2_amd import summary as sm

K = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_lowess.npz"))


def test_vectorised_lowess_matches_reference_vectors():
    for i in range(6):
        out = sm.lowess(K[f"x{i}"], K[f"y{i}"], frac=float(K[f"f{i}"]))
        np.testing.assert_allclose(out, K[f"out{i}"], rtol=1e-9, atol=1e-9)


def test_choose_cutoff_matches_oracle_rule():
    rng = np.random.default_rng(5)
    for t in range(20):
        theta = np.linspace(rng.uniform(0, 0.3), 0.95, 50)
        num_rej = np.round(np.maximum(rng.uniform(5, 900) * np.exp(-((theta - rng.uniform(0.2, 0.7)) / 0.4) ** 2)
                                      + rng.normal(0, 10, 50), 0)).astype(int)
        if t % 5 == 0:
            num_rej[:] = np.minimum(num_rej, 9)
        j, fit = sm.choose_cutoff(theta, num_rej)
        ref_fit = orc.lowess(theta, num_rej, frac=1 / 5)
        np.testing.assert_allclose(fit, ref_fit, rtol=1e-9, atol=1e-9)
        if num_rej.max() <= 10:
            assert j == 0
        else:
            res = num_rej[num_rej > 0] - ref_fit[num_rej > 0]
            thr = ref_fit.max() - np.sqrt(np.mean(res**2))
            exp = int(np.where(num_rej > thr)[0][0]) if np.any(num_rej > thr) else 0
            assert j == exp
