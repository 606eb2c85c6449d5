"""Host-side (Python) cost of one DeseqPipeline.deseq2() step with every C call stubbed out: what the GPU
waits for between dependent stages.  Runs anywhere (no GPU):  python tools/host_overhead.py [G N]"""
import cProfile
import ctypes as C
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pydeseq2_amd import _lib  # noqa: E402
from pydeseq2_amd import pipeline as pl  # noqa: E402

libc = C.CDLL(None)
libc.calloc.restype = C.c_void_p
libc.calloc.argtypes = [C.c_size_t, C.c_size_t]


class _FakeLib:
    def dsq_prior_mad_work_doubles(self, n):
        return 16

    def dsq_size_factors_work_doubles(self, n, g):
        return 16

    def __getattr__(self, name):
        return lambda *a: 0


class FakeContext(_lib.Context):
    def __init__(self):
        self.lib = _FakeLib()
        self.h = None
        self.device = 0
        self._next = 1 << 20
        self.calls = {}

    def close(self):
        pass

    def call(self, name, *args):
        self.calls[name] = self.calls.get(name, 0) + 1
        if name == "dsq_malloc":
            args[1]._obj.value = self._next
            self._next += (args[0].value + 255) & ~255
        elif name == "dsq_host_alloc":
            args[1]._obj.value = libc.calloc(1, args[0].value + 64)
        elif name == "dsq_dev_trend_fit":
            args[-2]._obj.value = 1
            args[-3][0], args[-3][1] = 1.0, 0.05
        elif name == "dsq_dev_trend_prior":
            args[-4][0], args[-4][1] = 1.0, 0.05
            args[-3]._obj.value = 1
            args[-1]._obj.value = 0.5
        elif name == "dsq_dev_prior_mad":
            args[-1]._obj.value = 0.5

    def d2h(self, arr, dptr):
        arr.fill(1)
        return arr


def main():
    G, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60000, 1000)
    from pydeseq2_amd.synth import synth_counts

    counts, X = synth_counts(2000, N, "2level", 0)
    counts = np.tile(counts, (1, G // 2000))
    ctx = FakeContext()
    pipe = pl.DeseqPipeline(counts, X, ctx=ctx)
    pipe.deseq2()
    ctx.calls.clear()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        pipe.deseq2()
    dt = (time.perf_counter() - t0) / reps
    print(f"host-only step: {dt * 1e3:.3f} ms   ({sum(ctx.calls.values()) / reps:.0f} C calls per step)")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        pipe.deseq2()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
    print(sorted(ctx.calls.items(), key=lambda kv: -kv[1])[:30])


if __name__ == "__main__":
    main()
