"""bench.py's graph-build sequence with the library's stage timers: a 4096-row warm-up, then the first full-size weightmatrix.knn
(first_call_s), phase by phase.  Usage: GLX_TIMING=1 python scripts/first_call_probe.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip, utils
_hip.require_device()
labels = bench.load_labels(70000)
X = bench.make_features(labels)
gl.weightmatrix.knn(X[:4096], 10)
sys.stderr.write('--- first full-size call\n')
for rep in range(3):
    tt = [time.perf_counter()]
    res = _hip.KnnResult(X, 11, want_order=True); tt.append(time.perf_counter())
    ta = time.perf_counter()
    perm = _hip.pinned_empty((res.n,), np.int32)
    tb = time.perf_counter()
    _hip.load().glx_knn_result_order(res._h, _hip._ptr(perm))
    tc = time.perf_counter()
    _hip.load().glx_knn_result_order(res._h, _hip._ptr(perm))
    td = time.perf_counter()
    print('   order: pinned_empty %.2f ms, copy %.2f ms, copy again %.2f ms' % ((tb - ta) * 1e3, (tc - tb) * 1e3, (td - tc) * 1e3), flush=True)
    order = res.order(); tt.append(time.perf_counter())
    Wy = res.to_csr(11, kernel='gaussian', sym=1); tt.append(time.perf_counter())
    res.close(); tt.append(time.perf_counter())
    fp = utils.symmetric_fingerprint(Wy); tt.append(time.perf_counter())
    print('call %d: search %.2f | order %.2f | to_csr %.2f | close %.2f | fingerprint %.2f ms' % ((rep + 1,) + tuple((b - a) * 1e3 for a, b in zip(tt[:-1], tt[1:]))), flush=True)
    sys.stderr.write('--- next call\n')
