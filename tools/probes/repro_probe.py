"""Pass-to-pass reproducibility of the genewise stage's pieces on the general kernels (one gene with a huge outlier count)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _mixed_case
from pydeseq2_amd import DeseqPipeline

counts, X = _mixed_case(5, 4, 200, 3000, 12, ())
counts = counts.copy(); counts[:, 17] = 0; counts[3, 40:44] = 150000
pipe = DeseqPipeline(counts, X, device=0)
pipe._lfc_overlap = False
prev = None
for it in range(8):
    st = pipe.begin_step(upto=None)
    pipe.advance(st, "genewise")
    pipe.publish(st, ("nm", "mom", "gw", "gconv"))
    mu = pipe.mu_hat_host(st)
    beta = pipe._down(st.mh.d_beta_fit, st.Gn * pipe.P).reshape(st.Gn, pipe.P)
    its = pipe._down(st.S["_irls_it"], st.Gn, np.int32)
    cur = dict(sf=np.array(st.r.size_factors), mom=np.array(st.r.mom_dispersions), gw=np.array(st.r.genewise_dispersions),
               mu=mu.copy(), beta=beta.copy(), its=its.copy())
    if prev is not None:
        for k in cur:
            a, b = prev[k], cur[k]
            bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
            if bad.any():
                idx = np.argwhere(bad)
                print(it, k, len(idx), "differ; first", idx[0], a[tuple(idx[0])], b[tuple(idx[0])])
    prev = cur
print("its gene 40", prev["its"][39:44], "beta40", prev["beta"][39])
