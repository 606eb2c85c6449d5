"""Run-to-run bit reproducibility of the tail entry points (apeGLM shrinkage, padj) and of the plug-in's Inference methods."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import nbglm_oracle as orc
from tests.test_gpu_parity import _mixed_case, _wide_case
from pydeseq2_amd import DeseqPipeline, HipInference
from pydeseq2_amd.summary import lfc_shrink, summary


def same(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


cases = {
    "2level p2": lambda: orc.synth_counts(3000, 120, "2level", 13),
    "3factor p8": lambda: orc.synth_counts(2000, 120, "3factor", 11),
    "mixed p8": lambda: _mixed_case(8, 3, 600, 1500, 5, (2, 4)),
    "wide mixed14": lambda: _wide_case("mixed14", 600, 160, 4),
    "wide factor16": lambda: _wide_case("factor16", 600, 160, 3),
}
for name, make in cases.items():
    counts, X = make()
    counts = np.array(counts, copy=True)
    counts[:, 17] = 0
    counts[3, 40:44] = 150000
    pipe = DeseqPipeline(counts, X, device=0)
    P = X.shape[1]
    c = np.zeros(P); c[1] = 1.0
    res = pipe.deseq2(contrast=c)
    out = []
    for it in range(4):
        # something else on the device in between: another pass (recycles buffers, leaves other LDS contents behind)
        pipe.deseq2(contrast=c)
        lfc, se, conv, ps = lfc_shrink(pipe, res, 1)
        s = summary(res, c, ctx=pipe.ctx)
        out.append((lfc, se, conv, s["padj"]))
    bad = [k for k, nm in enumerate(("shrunk LFC", "shrunk SE", "shrink converged", "padj"))
           if not all(same(out[0][k], o[k]) for o in out[1:])]
    print(f"{name:16s} tail:", "bit-identical" if not bad else [("shrunk LFC", "shrunk SE", "shrink converged", "padj")[k] for k in bad])
    # the plug-in's methods on host arrays, twice
    inf = HipInference(device=0)
    nz = counts.sum(0) > 0
    cn = counts[:, nz]
    sf = res.size_factors
    r1 = []
    for it in range(3):
        if it:
            pipe.deseq2(contrast=c)
        normed = cn / sf[:, None]
        rough = inf.fit_rough_dispersions(normed, X)
        mom = inf.fit_moments_dispersions(normed, sf)
        a0 = np.clip(np.minimum(rough, mom), 1e-8, 10.0)
        b, mu, H, conv = inf.irls(cn, sf, X, a0, 0.5, 1e-8)
        al, ac = inf.alpha_mle(cn, X, mu, a0, 1e-8, 10.0)
        r1.append((rough, mom, b, mu, H, conv, al, ac))
    names = ("rough", "moments", "irls beta", "irls mu", "irls hat", "irls conv", "alpha", "alpha conv")
    bad = [names[k] for k in range(len(names)) if not all(same(r1[0][k], o[k]) for o in r1[1:])]
    print(f"{name:16s} plug-in:", "bit-identical" if not bad else bad)
    pipe.close()
