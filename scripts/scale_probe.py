"""Developer probe: config-4-shaped run (isotropic blobs, d=64, k=10, C=10) at n given on the command line."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(2)
labels = rng.integers(0, 10, size=n)
centers = rng.normal(size=(10, 64)) * 4
X = centers[labels] + rng.normal(size=(n, 64))
t0 = time.perf_counter(); ind, dist = gl.weightmatrix.knnsearch(X, 11); t1 = time.perf_counter()
st = _hip.knn_stats()
print('knn n=%d d=64: %.2f s wall, tile %.1f ms (%.1f TFLOP/s), rerank %.1f ms, fallback rows %d' % (
    n, t1 - t0, st['tile_ms'], 2.0 * n * n * st['dpa'] / st['tile_ms'] / 1e9, st['rerank_ms'], st['fallback_rows']))
t0 = time.perf_counter(); W = gl.weightmatrix.knn(None, 10, knn_data=(ind, dist)); t1 = time.perf_counter()
print('weight matrix: nnz=%d max row %d, %.2f s' % (W.nnz, np.diff(W.indptr).max(), t1 - t0))
assert (abs(W - W.T) > 0).nnz == 0
train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
T = 50
m = gl.ssl.poisson(W, solver='gradient_descent', min_iter=T, max_iter=T)
t0 = time.perf_counter(); dev, aux = m._operators(); t1 = time.perf_counter()
print('host operator setup + upload: %.2f s' % (t1 - t0))
src, k = gl.ssl._poisson_source(n, train_ind, labels[train_ind])
v0 = np.zeros(n); v0[train_ind] = 1; v0 /= v0.sum()
sw = _hip.Sweep(dev, k, min_iter=T, max_iter=T, use_hipgraph=True)
sw.set_problem(aux['D'] * src, v0 / aux['deg'], aux['deg'], aux['vinf'])
sw.run()
tot = 0.0
for _ in range(5):
    _, ms = sw.run(); tot += ms
us = tot * 1e3 / (5 * T)
ab = bench.algorithmic_bytes(n, W.nnz, 10, 8, 8)
print('sweep fp64: %.1f us/launch, algorithmic %.1f MB -> %.2f TB/s = %.1f%% of 8 TB/s; %s' % (us, ab / 1e6, ab / us / 1e6, ab / us / 1e6 / 8 * 100, dev.info()))
u = sw.fetch()
pred = np.argmax(u, axis=1)
print('accuracy %.2f%%' % gl.ssl.ssl_accuracy(pred, labels, train_ind))
