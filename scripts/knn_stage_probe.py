"""Developer probe: wall time of the stages of weightmatrix.knn(X, 10) at config 2, call after call."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
labels = bench.load_labels(70000); X = bench.make_features(labels)
for i in range(10):
    t0 = time.perf_counter()
    J, D = _hip.knn_bruteforce(X, 11)
    t1 = time.perf_counter()
    d = D[:, :11]; sq = d * d; w = np.exp(-4 * sq / sq[:, 10][:, None])
    t2 = time.perf_counter()
    W = _hip.knn_to_csr(J, D, 11, kernel='given', sym=1, weights=w)
    t3 = time.perf_counter()
    W2 = gl.weightmatrix.knn(X, 10)
    t4 = time.perf_counter()
    print('call %d: search %.1f ms (device %.1f)  exp %.1f ms  assemble %.1f ms | knn(X,10) %.1f ms' % (
        i, (t1 - t0) * 1e3, _hip.knn_stats()['total_ms'], (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
