"""Developer probe: per-launch time of the sweep kernel on the config-2 graph and variants."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
from scipy import sparse

labels = bench.load_labels(70000)
X = bench.make_features(labels)
ind, dist = gl.weightmatrix.knnsearch(X, 11)
train_ind = gl.trainsets.generate(labels, rate=1, seed=0)


def probe(W, name, dtype=np.float64, steps=30, T=50):
    n = W.shape[0]
    m = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=(dtype == np.float32), min_iter=T, max_iter=T)
    dev, aux = m._operators()
    src, k = gl.ssl._poisson_source(n, train_ind, labels[train_ind])
    v0 = np.zeros(n); v0[train_ind] = 1; v0 /= v0.sum()
    sw = _hip.Sweep(dev, k, min_iter=T, max_iter=T, use_hipgraph=True)
    sw.set_problem(aux['D'] * src, v0 / aux['deg'], aux['deg'], aux['vinf'])
    for _ in range(3):
        sw.run()
    tot = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        _, ms = sw.run(); tot += ms
    wall = time.perf_counter() - t0
    lens = np.diff(W.indptr)
    print('%-28s %s n=%d nnz=%d maxrow=%d  %.2f us/launch (events)  %.2f us/launch (wall)  info=%s' % (
        name, np.dtype(dtype).name, n, W.nnz, lens.max(), tot * 1e3 / (steps * T), wall * 1e6 / (steps * T), dev.info()))
    sw.close()


W = gl.weightmatrix.knn(None, 10, knn_data=(ind, dist))
Wd = gl.weightmatrix.knn(None, 10, knn_data=(ind, dist), symmetrize=False)
probe(W, 'symmetric k=10')
probe(W, 'symmetric k=10', np.float32)
probe(Wd, 'directed 10/row')
# cap hubs: drop entries beyond 32 per row
Wc = W.tolil()
lens = np.diff(W.indptr)
rows = np.where(lens > 32)[0]
Wc = W.copy()
for r in rows:
    s, e = Wc.indptr[r], Wc.indptr[r + 1]
    Wc.data[s + 32:e] = 0
Wc.eliminate_zeros()
probe(Wc, 'symmetric, rows capped at 32')

# regular graphs: exactly L entries per row, random columns
rng = np.random.default_rng(0)
for L in (4, 16, 32, 64):
    n = 70000
    cols = rng.integers(0, n, size=(n, L))
    rows = np.repeat(np.arange(n), L)
    A = sparse.csr_matrix((rng.random(n * L) + 0.1, (rows, cols.ravel())), shape=(n, n))
    A = sparse.csr_matrix(A.T)   # poisson uses W^T; keep row lengths roughly L via transpose of transpose
    probe(sparse.csr_matrix(A.T), 'regular L=%d' % L)
