"""Developer probe: tile-kernel time of the split-bf16 kNN filter on four shapes, with a digest of the neighbour lists
(compile-time variants of knn.hip are compared by running this under different GLX_CXXFLAGS in one job)."""
import os, sys, hashlib, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphlearning_amd import _hip


def blobs(n, d, seed, scale):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 10, size=n)
    return rng.normal(size=(10, d))[lab] * scale + rng.normal(size=(n, d))


cases = [('config2 n=70000 d=20 k=11', bench.make_features(bench.load_labels(70000)), 11),
         ('config3 n=60000 d=32 k=21', blobs(60000, 32, 1, 1.2), 21),
         ('n=300000 d=64 k=11', blobs(300000, 64, 2, 4.0), 11)]
if 'big' in sys.argv:
    cases.append(('n=1000000 d=64 k=11', blobs(1000000, 64, 2, 4.0), 11))
ref = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'g4_large_meta.json')))['config2']['J_sha']
print('GLX_CXXFLAGS = %r' % os.environ.get('GLX_CXXFLAGS', ''))
for name, X, k in cases:
    _hip.knn_bruteforce(X, k)
    best = None
    for _ in range(3):
        J, D = _hip.knn_bruteforce(X, k)
        st = _hip.knn_stats()
        best = st if best is None or st['tile_ms'] < best['tile_ms'] else best
    sha = hashlib.sha256(np.ascontiguousarray(J).tobytes()).hexdigest()[:16]
    n, d = X.shape
    note = ''
    if name.startswith('config2'):
        note = ' == cKDTree' if sha == ref else ' != cKDTree (%s)' % ref
    print('  %-28s tile %8.3f ms (%6.1f TFLOP/s of 2n^2d) fallback rows %4d  J %s%s' % (
        name, best['tile_ms'], 2.0 * n * n * d / best['tile_ms'] / 1e9, best['fallback_rows'], sha, note))
