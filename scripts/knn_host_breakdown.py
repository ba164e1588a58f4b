"""Developer probe: where the wall time of weightmatrix.knn(X, 10) goes at config 2 (n = 70000, d = 20)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip, utils

X = bench.make_features(bench.load_labels(70000))
for _ in range(3):
    gl.weightmatrix.knn(X, 10)


def timed(f, reps=7):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = f()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


t_all, W = timed(lambda: gl.weightmatrix.knn(X, 10))
t_search, (J, D) = timed(lambda: gl.weightmatrix.knnsearch(X, 11))
st = _hip.knn_stats()


def host_weights():
    d = np.asarray(D)[:, :11]
    DD = d * d
    return np.exp(-4 * DD / DD[:, 10][:, None])


t_w, w = timed(host_weights)
t_csr, W2 = timed(lambda: _hip.knn_to_csr(J, D, 11, kernel='given', sym=1, weights=w))
t_stamp, _ = timed(lambda: utils.symmetric_fingerprint(W2))
t_contig, _ = timed(lambda: np.ascontiguousarray(X))
print('weightmatrix.knn(X, 10): %.2f ms' % t_all)
print('  knnsearch            %.2f ms (device: tile %.2f + rerank %.2f + fallback %.2f ms)' % (t_search, st['tile_ms'], st['rerank_ms'], st['fallback_ms']))
print('  numpy exp weights    %.2f ms' % t_w)
print('  knn_to_csr           %.2f ms (upload, 4 kernels, 2 host scans, download, scipy wrap)' % t_csr)
print('  symmetric stamp      %.3f ms' % t_stamp)
