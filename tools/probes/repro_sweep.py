"""Pass-to-pass bit reproducibility of deseq2() across the kernel families: python tools/probes/repro_sweep.py [passes]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import nbglm_oracle as orc
from tests.test_gpu_parity import _mixed_case, _wide_case
from pydeseq2_amd import DeseqPipeline

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F = ("size_factors", "mom_dispersions", "genewise_dispersions", "genewise_converged", "fitted_dispersions", "MAP_dispersions",
     "MAP_converged", "dispersions", "LFC", "LFC_converged", "lfcSE", "stat", "pvalue", "cooks_outlier", "replaced", "refitted")
cases = {
    "2level 6000x200": lambda: orc.synth_counts(6000, 200, "2level", 13),
    "2factor 4000x120": lambda: orc.synth_counts(4000, 120, "2factor", 3),
    "3factor 3200x120": lambda: orc.synth_counts(3200, 120, "3factor", 11),
    "3factor 800x60 (small batch)": lambda: orc.synth_counts(800, 60, "3factor", 5),
    "mixed 3000x1400": lambda: _mixed_case(8, 3, 1400, 3000, 5, (2, 4)),
    "general 3000x200": lambda: _mixed_case(5, 4, 200, 3000, 12, ()),
    "wide factor16": lambda: _wide_case("factor16", 1200, 160, 3),
    "wide mixed14": lambda: _wide_case("mixed14", 800, 200, 4),
}
for name, make in cases.items():
    counts, X = make()
    counts = np.array(counts, copy=True)
    counts[:, 17] = 0
    counts[3, 40:44] = 150000
    counts[7, 100] = 90000
    pipe = DeseqPipeline(counts, X, device=0)
    prev, bad = None, {}
    for it in range(passes):
        r = pipe.deseq2()
        cur = {f: np.array(getattr(r, f), copy=True) for f in F}
        if prev is not None:
            for f in F:
                a, b = np.asarray(prev[f], float), np.asarray(cur[f], float)
                d = ~((a == b) | (np.isnan(a) & np.isnan(b)))
                if d.ndim > 1:
                    d = d.any(axis=1)
                if d.any():
                    bad[f] = max(bad.get(f, 0), int(d.sum()))
        prev = cur
    print(f"{name:32s} row_mode {pipe._row_mode} forks {pipe.lfc_forks}: ", "bit-identical" if not bad else bad)
    pipe.close()
