"""DMA into a hipHostRegister-ed numpy buffer (first and later copies) vs into hipHostMalloc memory vs the pageable path."""
import ctypes as C, sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from pydeseq2_amd._lib import Context, DeviceArray, _PinnedPool
ctx = Context(0)
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
G, N = 60000, 1000
d = DeviceArray(ctx, (G, 1008), np.float64)
ctx.call("dsq_memset", C.c_void_p(d.ptr), 0, C.c_size_t(d.nbytes)); ctx.sync()
def d2h(host):
    t = time.perf_counter()
    ctx.call("dsq_d2h_2d", C.c_void_p(host.ctypes.data), C.c_size_t(N * 8), C.c_void_p(d.ptr), C.c_size_t(1008 * 8), C.c_size_t(N * 8), C.c_size_t(G))
    ctx.sync()
    return (time.perf_counter() - t) * 1e3
def touch(a, T=16):
    n = a.size; f = a.reshape(-1)
    th = [threading.Thread(target=lambda lo, hi: f.__setitem__(slice(lo, hi, 512), 0.0), args=(i * n // T, (i + 1) * n // T)) for i in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
pool = _PinnedPool(ctx)
for rep in range(3):
    t = time.perf_counter(); a = np.empty((G, N)); touch(a); rc = hip.hipHostRegister(C.c_void_p(a.ctypes.data), a.nbytes, 0); t_prep = (time.perf_counter() - t) * 1e3
    t1, t2, t3 = d2h(a), d2h(a), d2h(a)
    t = time.perf_counter(); slab = pool.take(G * N * 8); t_pin = (time.perf_counter() - t) * 1e3
    h = slab.view(0, G * N, np.float64).reshape(G, N)
    p1, p2 = d2h(h), d2h(h)
    print(f"rep {rep}: registered: prepare {t_prep:.1f} ms (rc {rc}), DMA {t1:.1f} / {t2:.1f} / {t3:.1f} ms | hipHostMalloc {t_pin:.1f} ms, DMA {p1:.1f} / {p2:.1f} ms")
    hip.hipHostUnregister(C.c_void_p(a.ctypes.data)); del a, h, slab
