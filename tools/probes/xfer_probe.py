"""PCIe transfer rates of the shapes the plug-in path moves (c3: 60 000 x 1000 fp64 = 480 MB): D2H into fresh / touched
pageable memory and into page-locked memory, contiguous and pitched (ldn 1008 -> 1000); H2D from pageable and pinned."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pydeseq2_amd._lib import Context, DeviceArray  # noqa: E402

ctx = Context(0)
_vp = C.c_void_p
G, N, ldn = 60000, 1000, 1008
d = DeviceArray(ctx, (G, N), np.float64, ld=ldn)
ctx.call("dsq_memset", _vp(d.ptr), 0, C.c_size_t(G * ldn * 8))
ctx.sync()
nb = G * N * 8


def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def pinned(nbytes):
    p = _vp()
    ctx.call("dsq_host_alloc", C.c_size_t(nbytes), C.byref(p))
    buf = (C.c_char * nbytes).from_address(p.value)
    return np.frombuffer(buf, dtype=np.float64), p


def fresh_contig():
    o = np.empty(G * N)
    ctx.call("dsq_d2h", _vp(o.ctypes.data), _vp(d.ptr), C.c_size_t(nb))


def fresh_2d():
    o = np.empty((G, N))
    ctx.call("dsq_d2h_2d", _vp(o.ctypes.data), C.c_size_t(N * 8), _vp(d.ptr), C.c_size_t(ldn * 8), C.c_size_t(N * 8),
             C.c_size_t(G))


print("np.empty + first touch (np.zeros-like fill) of 480 MB: %.1f ms" % t(lambda: np.empty(G * N).fill(0.0)))
print("D2H contiguous 480 MB -> fresh np.empty : %.1f ms" % t(fresh_contig))
print("D2H pitched   480 MB -> fresh np.empty : %.1f ms" % t(fresh_2d))
o = np.zeros((G, N))
print("D2H contiguous -> touched pageable      : %.1f ms" % t(lambda: ctx.call("dsq_d2h", _vp(o.ctypes.data), _vp(d.ptr), C.c_size_t(nb))))
print("D2H pitched    -> touched pageable      : %.1f ms" % t(lambda: ctx.call("dsq_d2h_2d", _vp(o.ctypes.data), C.c_size_t(N * 8), _vp(d.ptr), C.c_size_t(ldn * 8), C.c_size_t(N * 8), C.c_size_t(G))))
hp, p = pinned(nb)
print("D2H contiguous -> pinned                : %.1f ms" % t(lambda: ctx.call("dsq_d2h", p, _vp(d.ptr), C.c_size_t(nb))))
print("D2H pitched    -> pinned                : %.1f ms" % t(lambda: ctx.call("dsq_d2h_2d", p, C.c_size_t(N * 8), _vp(d.ptr), C.c_size_t(ldn * 8), C.c_size_t(N * 8), C.c_size_t(G))))
print("H2D contiguous <- touched pageable      : %.1f ms" % t(lambda: ctx.call("dsq_h2d", _vp(d.ptr), _vp(o.ctypes.data), C.c_size_t(nb))))
print("H2D contiguous <- pinned                : %.1f ms" % t(lambda: ctx.call("dsq_h2d", _vp(d.ptr), p, C.c_size_t(nb))))
print("host copy 480 MB pinned -> fresh np.empty: %.1f ms" % t(lambda: np.copyto(np.empty(G * N), hp)))
print("host copy 480 MB pinned -> touched       : %.1f ms" % t(lambda: np.copyto(o.reshape(-1), hp)))
x = np.random.default_rng(0).poisson(50, (N, G)).astype(np.int64)
print("numpy x[:, idx] fancy copy of 480 MB int64: %.1f ms" % t(lambda: x[:, np.arange(G)], 2))
v = x / 1.5
print("numpy (v == 0).all(axis=0): %.1f ms" % t(lambda: (v == 0).all(axis=0), 2))
import os
print("cpus", os.cpu_count())
