"""Config 2 (70 000 x 20): the sweep on the library's breadth-first vertex order vs the chained cell order a clustered search leaves
behind (glx_knn_last_order) -- would the 3.7 ms pass over the graph be dispensable here too?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
from scipy import sparse
labels = bench.load_labels(70000); X = bench.make_features(labels)
n = len(X)
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=5, seed=0)
P, deg, dinv = gl.ssl._poisson_operator_symmetric(W)
src, k = gl.ssl._poisson_source(n, ti, labels[ti])
v0 = np.zeros(n); v0[ti] = 1; v0 /= v0.sum()
T = 50
for name in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['rcm', 'none', 'cells16', 'cells64', 'cells256', 'cells1024']):
    if name == 'rcm':
        order = None
    elif name == 'none':
        order = np.arange(n, dtype=np.int32)
    else:
        _hip.knn_bruteforce(X, 11, clustered=int(name[5:]))
        order = _hip.knn_last_order(n)
    for dtype in (np.float64, np.float32):
        dev = _hip.DeviceGraph(P, dtype=dtype, order=order)
        sw = _hip.Sweep(dev, k, min_iter=T, max_iter=T, use_hipgraph=True)
        sw.set_problem(sparse.spdiags(dinv, 0, n, n).tocsr() * src, v0 / deg, deg, deg / np.sum(deg))
        sw.run()
        tot = 0.0
        for _ in range(200):
            tot += sw.run()[1]
        print('order %-9s %s: %.3f us per sweep' % (name, 'f64' if dtype == np.float64 else 'f32', tot * 1e3 / (200 * T)), flush=True)
        sw.close(); dev.close()
