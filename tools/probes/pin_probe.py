"""First-call cost of an N x G output layer: page-locked buffer (hipHostMalloc + DMA) vs pageable np.empty + hipMemcpy2D."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pydeseq2_amd._lib import Context, DeviceArray, _PinnedPool
ctx = Context(0)
G, N = 60000, 1000
d = DeviceArray(ctx, (G, 1008), np.float64)
ctx.call("dsq_memset", __import__("ctypes").c_void_p(d.ptr), 0, __import__("ctypes").c_size_t(d.nbytes))
ctx.sync()
pool = _PinnedPool(ctx)
for rep in range(2):
    t = time.perf_counter(); slab = pool.take(G * N * 8); t_pin = time.perf_counter() - t
    host = slab.view(0, G * N, np.float64).reshape(G, N)
    import ctypes as C
    t = time.perf_counter()
    ctx.call("dsq_d2h_2d", C.c_void_p(host.ctypes.data), C.c_size_t(N * 8), C.c_void_p(d.ptr), C.c_size_t(1008 * 8), C.c_size_t(N * 8), C.c_size_t(G))
    ctx.sync(); t_dma = time.perf_counter() - t
    t = time.perf_counter(); out = np.empty((G, N)); 
    ctx.call("dsq_d2h_2d", C.c_void_p(out.ctypes.data), C.c_size_t(N * 8), C.c_void_p(d.ptr), C.c_size_t(1008 * 8), C.c_size_t(N * 8), C.c_size_t(G))
    ctx.sync(); t_page = time.perf_counter() - t
    print(f"rep {rep}: pinned alloc {t_pin*1e3:.1f} ms + DMA {t_dma*1e3:.1f} ms | pageable np.empty + copy {t_page*1e3:.1f} ms")
    del host, slab, out
