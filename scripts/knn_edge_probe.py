"""Edge inputs through the public kNN API around the clustered-search threshold: results must equal the all-pairs search."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip
rng = np.random.default_rng(0)
def check(name, X, k, **kw):
    J0, D0 = _hip.knn_bruteforce(np.asarray(X, dtype=np.float64), k, clustered=0, **kw)
    J0, D0 = np.array(J0), np.array(D0)
    J1, D1 = gl.weightmatrix.knnsearch(X, k, **kw)
    st = _hip.knn_stats()
    ok = np.array_equal(J0, J1) and np.array_equal(D0, D1)
    print('%-44s cells %3d  order left behind %-5s identical %s' % (name, st['cells'], _hip.knn_last_order(len(X)) is not None, ok), flush=True)
    assert ok, name
n = 1 << 17
base = rng.normal(size=(6, 24)) * 5
X = base[rng.integers(0, 6, size=n)] + rng.normal(size=(n, 24))
check('n = 2^17 exactly', X, 11)
check('n = 2^17 - 1', X[:-1], 11)
check('float32 input', X.astype(np.float32), 11)
check('non-contiguous input (every 2nd column of 48)', np.asfortranarray(np.hstack([X, X]))[:, ::2], 11)
check('angular', X + 20.0, 11, similarity='angular')
Xd = X.copy(); Xd[n // 2:] = Xd[:n // 2]
check('every point twice', Xd, 11)
check('k = 60', X[:140000 if n >= 140000 else n], 60)
check('k = 1', X, 1)
Xc = np.zeros((n, 8)); Xc[:, 0] = np.arange(n) % 7
check('seven distinct points, hugely duplicated', Xc, 11)
check('one feature', rng.normal(size=(n, 1)), 5)
W = gl.weightmatrix.knn(X, 10)
W0 = gl.weightmatrix.knn(X, 10, knn_data=_hip.knn_bruteforce(X, 11, clustered=0))
print('weight matrix identical:', np.array_equal(W.indptr, W0.indptr) and np.array_equal(W.indices, W0.indices) and np.array_equal(W.data, W0.data))
