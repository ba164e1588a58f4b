"""weightmatrix.knn through the public API at large n on one GPU (blobs d = 64, random order): time and a sampled exactness check."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4000000
g = np.random.default_rng(2)
lab = g.integers(0, 10, size=n); cen = g.normal(size=(10, 64)) * 4
t0 = time.perf_counter()
X = cen[lab] + g.normal(size=(n, 64))
print('features %.1f s' % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
J, D = gl.weightmatrix.knnsearch(X, 11)
t1 = time.perf_counter()
st = _hip.knn_stats()
print('knnsearch n=%d: %.2f s (tile %.2f s, %d cells, visited %.1f %%, fallback rows %d)' % (n, t1 - t0, st['tile_ms'] / 1e3, st['cells'], 100 * st['visited_share'], st['fallback_rows']), flush=True)
# sampled exactness: brute force in numpy for a few rows
rows = g.integers(0, n, size=6)
ok = True
for i in rows:
    d2 = np.sum((X - X[i]) ** 2, axis=1)
    idx = np.lexsort((np.arange(n), d2))[:11]
    ok = ok and np.array_equal(np.sort(idx), np.sort(np.asarray(J[i])))
print('sampled rows equal to numpy brute force:', ok, flush=True)
t0 = time.perf_counter()
W = gl.weightmatrix.knn(X, 10, knn_data=(J, D))
print('weights + assembly: %.2f s, nnz %d' % (time.perf_counter() - t0, W.nnz), flush=True)
