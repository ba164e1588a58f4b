"""Developer probe: weight-matrix assembly when one vertex is every row's neighbour (a hub row of n entries merged by one
workgroup in a global scratch): time and equality with the oracle's scipy assembly."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from oracle import gl_oracle as orc
rng = np.random.default_rng(4)
n, k = 150000, 6
ind = np.empty((n, k), dtype=np.int64)
ind[:, 0] = np.arange(n)
ind[:, 1] = 0
for c in range(2, k):
    ind[:, c] = (np.arange(n) + rng.integers(1, 50, size=n)) % n
ind[0, 1] = 7
dist = np.sort(rng.random((n, k)), axis=1); dist[:, 0] = 0
for kernel in ('gaussian', 'uniform'):
    t0 = time.perf_counter()
    W = gl.weightmatrix.knn(None, k - 1, kernel=kernel, knn_data=(ind, dist.copy()))
    t1 = time.perf_counter()
    Wo = orc.knn_weights(ind, dist.copy(), k - 1, kernel=kernel)
    ok = np.array_equal(W.indptr, Wo.indptr) and np.array_equal(W.indices, Wo.indices) and np.array_equal(W.data, Wo.data)
    print('%-8s: assembly %.1f ms (hub row %d entries), equal to the oracle: %s' % (kernel, (t1 - t0) * 1e3, np.diff(W.indptr).max(), ok))
