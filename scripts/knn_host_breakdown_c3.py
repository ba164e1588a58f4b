"""Developer probe: where the wall time of weightmatrix.knn(X, 20) goes at config 3 (n = 60000, d = 32, k = 20)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip, utils

lab3 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'cifar_labels.npz'))['labels'][:60000].astype(np.int64)
rng = np.random.default_rng(1)
X = (rng.normal(size=(10, 32)) * 1.2)[lab3] + rng.normal(size=(60000, 32))
K = 20
for _ in range(3):
    gl.weightmatrix.knn(X, K)


def timed(f, reps=7):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = f()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


t_all, W = timed(lambda: gl.weightmatrix.knn(X, K))
t_search, (J, D) = timed(lambda: gl.weightmatrix.knnsearch(X, K + 1))
st = _hip.knn_stats()


def host_weights():
    d = np.asarray(D)[:, :K + 1]
    DD = d * d
    return np.exp(-4 * DD / DD[:, K][:, None])


t_w, w = timed(host_weights)
t_csr, W2 = timed(lambda: _hip.knn_to_csr(J, D, K + 1, kernel='given', sym=1, weights=w))
t_stamp, _ = timed(lambda: utils.symmetric_fingerprint(W2))
print('weightmatrix.knn(X, %d): %.2f ms' % (K, t_all))
print('  knnsearch            %.2f ms (device: tile %.2f + rerank %.2f + fallback %.2f ms, %d fallback rows)' % (t_search, st['tile_ms'], st['rerank_ms'], st['fallback_ms'], st['fallback_rows']))
print('  numpy exp weights    %.2f ms (one thread here; row blocks on a few threads inside weightmatrix.knn)' % t_w)
print('  knn_to_csr           %.2f ms' % t_csr)
print('  symmetric stamp      %.3f ms' % t_stamp)
print('  stats', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
