"""Whole-pass mode, stopped after the genewise stage: which vector varies from pass to pass?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import _mixed_case
from pydeseq2_amd import DeseqPipeline

counts, X = _mixed_case(5, 4, 200, 3000, 12, ())
counts = counts.copy(); counts[:, 17] = 0; counts[3, 40:44] = 150000
pipe = DeseqPipeline(counts, X, device=0)
pipe._lfc_overlap = False
pipe.overlap = len(sys.argv) > 1 and sys.argv[1] == "overlap"
prev = None
for it in range(10):
    st = pipe.begin_step(upto="finish")
    pipe.advance(st, "genewise")
    pipe.ctx.sync()
    Gn = st.Gn
    cur = dict(sf=pipe._down(st.d_sf, pipe.N), mom=pipe._down(st.S["mom"], Gn), gw=pipe._down(st.S["gw"], Gn),
               beta=pipe._down(st.mh.d_beta_fit, Gn * pipe.P).reshape(Gn, pipe.P), its=pipe._down(st.S["_irls_it"], Gn, np.int32),
               mu=pipe.ctx.d2h_rows(st.mh.d_mu.ptr, Gn, pipe.N, pipe.ldn) if st.mh.d_mu is not None else np.zeros(1),
               nllc=pipe._down(st.mh.nll_const, Gn))
    pipe.advance(st, "finish")
    if prev is not None:
        for k in cur:
            a, b = prev[k], cur[k]
            bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
            if bad.any():
                idx = np.argwhere(bad)
                print(it, k, len(idx), "differ; first", idx[0], a[tuple(idx[0])], b[tuple(idx[0])])
    prev = cur
print("done", pipe.ldn)
