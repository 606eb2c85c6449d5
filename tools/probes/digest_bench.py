import ctypes as C, numpy as np, time, sys
sys.path.insert(0, ".")
from pydeseq2_amd import _lib
lib = _lib.load()
N, G = 1000, 60000
rng = np.random.default_rng(0)
a = rng.integers(0, 5000, size=(N, G), dtype=np.int64)
f = rng.random((N, G))
out = (C.c_ulonglong * 2)()
for name, arr, et in (("i64", a, 1), ("f64", f, 2)):
    for th in (1, 8, 16, 32, 48, 64):
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            lib.dsq_plugin_digest_host(C.c_void_p(arr.ctypes.data), et, 0, N, G, th, out)
            ts.append(time.perf_counter() - t)
        print(name, th, "threads", round(min(ts) * 1e3, 1), "ms", round(arr.nbytes / min(ts) / 1e9, 1), "GB/s")
