"""Developer probe: host-to-device copy rates from pageable and page-locked numpy memory (what the uploads of X, of the kNN
lists and of the operator cost)."""
import ctypes as C, time, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphlearning_amd import _hip
_hip.require_device()
hip = C.CDLL('libamdhip64.so')
for mb in (1, 11, 64, 512):
    n = mb * (1 << 20) // 8
    a = np.random.default_rng(0).random(n)
    p = _hip.pinned_empty((n,), np.float64); p[:] = a
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(n * 8)) == 0
    res = {}
    for name, src in (('pageable', a), ('page-locked', p)):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            assert hip.hipMemcpy(d, C.c_void_p(src.ctypes.data), C.c_size_t(n * 8), 1) == 0
            best = min(best, time.perf_counter() - t0)
        res[name] = best
    t0 = time.perf_counter(); p[:] = a; t_cp = time.perf_counter() - t0
    print('%4d MB: pageable %.2f ms (%.1f GB/s), page-locked %.2f ms (%.1f GB/s), numpy copy into page-locked %.2f ms' % (
        mb, res['pageable'] * 1e3, n * 8 / res['pageable'] / 1e9, res['page-locked'] * 1e3, n * 8 / res['page-locked'] / 1e9, t_cp * 1e3))
    hip.hipFree(d)
