"""Pass-to-pass reproducibility and two-launch LFC equality on one design: python tools/probes/lfc_fork_probe.py <kind>"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import nbglm_oracle as orc
from pydeseq2_amd import DeseqPipeline

kind = sys.argv[1]
if kind == "continuous":
    from tests.test_gpu_parity import _mixed_case
    counts, X = _mixed_case(5, 4, 200, 3000, 12, ())
elif kind == "mixed":
    from tests.test_gpu_parity import _mixed_case
    counts, X = _mixed_case(8, 3, 1400, 3000, 5, (2, 4))
else:
    counts, X = orc.synth_counts(3200, 120, kind, 11)
counts = counts.copy(); counts[:, 17] = 0; counts[3, 40:44] = 150000
pipe = DeseqPipeline(counts, X, device=0)
F = ("genewise_dispersions", "MAP_dispersions", "dispersions", "LFC", "lfcSE", "pvalue", "MAP_converged", "genewise_converged")
def diff(a, b, tag):
    for f in F:
        va, vb = np.asarray(getattr(a, f), float), np.asarray(getattr(b, f), float)
        bad = ~((va == vb) | (np.isnan(va) & np.isnan(vb)))
        if bad.ndim > 1: bad = bad.any(axis=1)
        if bad.any():
            i = np.nonzero(bad)[0]
            print(tag, f, len(i), "genes differ, first", i[:6], np.ravel(va[i[0]])[:3], np.ravel(vb[i[0]])[:3])
seq = [False, False] + [True, True, False] * int(sys.argv[2] if len(sys.argv) > 2 else 1)
if len(sys.argv) > 3 and sys.argv[3] == "off":
    seq = [False] * len(seq)
for ov in seq:
    pipe._lfc_overlap = ov
    r = pipe.deseq2()
    res = {f: np.array(getattr(r, f), copy=True) for f in F}
    cur = type("R", (), res)
    if "prev" in dir():
        diff(prev, cur, f"overlap {prev_ov}->{ov}:")
    prev, prev_ov = cur, ov
print("forks", pipe.lfc_forks)
