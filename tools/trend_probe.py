#!/usr/bin/env python3
"""Latency of the parametric-trend kernel (dsq_dev_trend_fit) per data pass.

The fit is a chain of dependent passes (one per L-BFGS-B evaluation + one per outlier filter): its duration is
passes x latency of a pass, nearly independent of the number of genes.  This prints, for a few gene counts, the
kernel's duration (HIP events around the call) so that variants of the pass protocol can be compared:

    DSQ_TREND_GRID=-1 python tools/trend_probe.py      # one workgroup
    DSQ_TREND_GRID=1  python tools/trend_probe.py      # cooperative grid at every size
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from pydeseq2_amd._lib import Context, DeviceArray  # noqa: E402


def main():
    ctx = Context(0)
    rng = np.random.default_rng(0)
    for n in (256, 2048, 7500, 60000):
        means = np.exp(rng.normal(4, 1.5, n))
        disp = (0.05 + 2.0 / means) * np.exp(rng.normal(0, 0.6, n))
        d_disp = DeviceArray.from_host(ctx, disp)
        d_means = DeviceArray.from_host(ctx, means)
        d_keep = DeviceArray(ctx, (n,), np.uint8)
        c2, ok, n_outer = (C.c_double * 2)(), C.c_int(0), C.c_int(0)
        ts = []
        for _ in range(12):
            ctx.timer_start()
            ctx.call("dsq_dev_trend_fit", C.c_void_p(d_disp.ptr), C.c_void_p(d_means.ptr), n, C.c_double(1e-8),
                     C.c_double(10.0), C.c_void_p(d_keep.ptr), c2, C.byref(ok), C.byref(n_outer))
            ts.append(ctx.timer_stop())
        ts = sorted(ts[2:])
        print(f"n={n:6d}  median {1e3 * ts[len(ts) // 2]:8.1f} us  min {1e3 * ts[0]:8.1f} us  outer={n_outer.value} "
              f"ok={ok.value} coeffs=({c2[0]:.6g}, {c2[1]:.6g})", flush=True)


if __name__ == "__main__":
    main()
