"""Wall-time split of weightmatrix.knn at n = 10^6, d = 64 (blobs, random order): search / weights / assembly / stamp."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip, utils, weightmatrix as wm
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
g = np.random.default_rng(2)
lab = g.integers(0, 10, size=n); X = (g.normal(size=(10, 64)) * 4)[lab] + g.normal(size=(n, 64))
gl.weightmatrix.knn(X[:200000], 10)
for rep in range(3):
    t0 = time.perf_counter(); J, D = wm.knnsearch(X, 11); t1 = time.perf_counter()
    st = _hip.knn_stats()
    W = wm._knn(None, 10, 'gaussian', None, True, 'raw', 'euclidean', (J, D), None); t2 = time.perf_counter()
    fp = utils.symmetric_fingerprint(W); t3 = time.perf_counter()
    t4 = time.perf_counter(); W2 = gl.weightmatrix.knn(X, 10); t5 = time.perf_counter()
    print('search %.3f s (tile %.1f ms, rerank %.1f ms, fallback %.1f ms, device total %.1f ms) | weights + assembly %.3f s | fingerprint %.3f s | weightmatrix.knn %.3f s'
          % (t1 - t0, st['tile_ms'], st['rerank_ms'], st['fallback_ms'], st['total_ms'], t2 - t1, t3 - t2, t5 - t4), flush=True)
    del J, D, W, W2
