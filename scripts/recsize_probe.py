import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
from scipy import sparse
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
n = 70000
deg = np.asarray(W.sum(axis=1)).ravel()
P = sparse.csr_matrix(sparse.diags(1 / deg) * W)
rng = np.random.default_rng(0)
for C in (10, 12):
    u0 = rng.normal(size=(n, C)); Db = np.zeros((n, C))
    dev = _hip.DeviceGraph(P)
    sw = _hip.Sweep(dev, C, 0, 0, True)
    sw.set_state(u0, Db); sw.iterate(50)
    t0 = time.perf_counter()
    for _ in range(20): sw.iterate(50)
    dt = time.perf_counter() - t0
    ok = np.array_equal(sw.fetch()[:5], sw.fetch()[:5])
    print('NOPAD=%s C=%d layout=%s: %.2f us/sweep' % (os.environ.get('GLX_REC_NOPAD', '0'), C, _hip.record_layout(C, np.float64, False), dt * 1e6 / 1000))
    sw.close(); dev.close()
# correctness of the unpadded layout
A = sparse.random(3000, 3000, density=0.004, random_state=1, format='csr'); x = rng.normal(size=(3000, 10))
G = _hip.DeviceGraph(A); print('bit-exact:', np.array_equal(G.spmm_bias(x), A * x)); G.close()
