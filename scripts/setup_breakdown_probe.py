"""Developer probe: where the first fit of ssl.poisson(gradient_descent) spends its time on the host (n given)."""
import os, sys, time
import numpy as np
from scipy import sparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip, graph as graph_mod
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(2)
labels = rng.integers(0, 10, size=n)
X = rng.normal(size=(10, 64))[labels] * 4 + rng.normal(size=(n, 64))
W = gl.weightmatrix.knn(X, 10)
del X
def T(f):
    t0 = time.perf_counter(); r = f(); return r, (time.perf_counter() - t0) * 1e3
Wz, t1 = T(lambda: W - sparse.spdiags(W.diagonal(), 0, n, n))
G, t2 = T(lambda: graph_mod.graph(Wz))
D, t3 = T(lambda: G.degree_matrix(p=-1))
P, t4 = T(lambda: D * Wz.transpose())
deg, t5 = T(lambda: G.degree_vector())
print('scipy: W - diag %.0f ms, graph() %.0f ms, degree_matrix %.0f ms, D * W^T %.0f ms, degree_vector %.0f ms' % (t1, t2, t3, t4, t5))
for reorder in ('1', '0'):
    os.environ['GLX_REORDER'] = reorder
    dev, t6 = T(lambda: _hip.DeviceGraph(P))
    sw, t7 = T(lambda: _hip.Sweep(dev, 10, 50, 50, True))
    print('GLX_REORDER=%s: DeviceGraph (csr copy + validation) %.0f ms, Sweep create (order + plan + upload) %.0f ms' % (reorder, t6, t7))
    sw.close(); dev.close()
