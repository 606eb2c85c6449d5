"""hipHostMalloc of 480 MB vs mmap + parallel first touch + hipHostRegister (is there a cheaper way to a page-locked layer?)"""
import ctypes as C, sys, time, threading
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipSetDevice(0)
nbytes = 60000 * 1000 * 8
for rep in range(3):
    p = C.c_void_p()
    t = time.perf_counter(); rc = hip.hipHostMalloc(C.byref(p), nbytes, 0); t_malloc = time.perf_counter() - t
    hip.hipHostFree(p)
    t = time.perf_counter(); a = np.empty(nbytes // 8); t_empty = time.perf_counter() - t
    # parallel first touch
    def touch(lo, hi):
        a[lo:hi:512] = 0.0
    n = len(a); T = 16
    t = time.perf_counter()
    th = [threading.Thread(target=touch, args=(i * n // T, (i + 1) * n // T)) for i in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    t_touch = time.perf_counter() - t
    t = time.perf_counter(); rc2 = hip.hipHostRegister(C.c_void_p(a.ctypes.data), nbytes, 0); t_reg = time.perf_counter() - t
    b = np.empty(nbytes // 8)
    t = time.perf_counter(); rc3 = hip.hipHostRegister(C.c_void_p(b.ctypes.data), nbytes, 0); t_reg_untouched = time.perf_counter() - t
    hip.hipHostUnregister(C.c_void_p(a.ctypes.data)); hip.hipHostUnregister(C.c_void_p(b.ctypes.data))
    print(f"rep {rep}: hipHostMalloc {t_malloc*1e3:.1f} ms (rc {rc}) | np.empty {t_empty*1e3:.2f} + touch(16 thr) {t_touch*1e3:.1f} + register {t_reg*1e3:.1f} ms (rc {rc2}) | register untouched {t_reg_untouched*1e3:.1f} ms (rc {rc3})")
