"""Developer probe: device time of the CG solves at configs 2 and 3 (host setup excluded)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip, graph as graph_mod
from scipy import sparse

labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
n = W.shape[0]
G = graph_mod.graph(W)
L = G.laplacian(normalization='normalized'); D = G.degree_matrix(p=-0.5)
src, k = gl.ssl._poisson_source(n, ti, labels[ti])
dev = _hip.DeviceGraph(L)
rhs = np.ascontiguousarray(D * src)
for mode in ('exact', 'tree'):
    if mode == 'tree': os.environ['GLX_CG_REDUCE'] = 'tree'
    dev.cg(rhs, tol=1e-3)
    t0 = time.perf_counter(); x, it, err = dev.cg(rhs, tol=1e-3); dt = time.perf_counter() - t0
    print('poisson CG 70k [%s reductions]: %d iterations in %.1f ms = %.0f it/s  (nnz*C*it/s = %.2e)' % (mode, it, dt * 1e3, it / dt, L.nnz * k * it / dt))
os.environ.pop('GLX_CG_REDUCE', None)

# ssl.laplace at config-3 shape (60 000 vertices, k = 20): the positive definite system, both reference-order sums per iteration
rng = np.random.default_rng(3)
lab3 = rng.integers(0, 10, size=60000)
X3 = (rng.normal(size=(10, 32)) * 2.5)[lab3] + rng.normal(size=(60000, 32))
W3 = gl.weightmatrix.knn(X3, 20)
ti3 = gl.trainsets.generate(lab3, rate=10, seed=0)
model = gl.ssl.laplace(W3)
model.fit(ti3, lab3[ti3])
t0 = time.perf_counter(); u = model.fit(ti3, lab3[ti3]); dt = time.perf_counter() - t0
print('laplace 60k k=20 [exact reductions]: fit %.1f ms' % (dt * 1e3))
