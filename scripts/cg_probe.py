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
