"""Developer probe: exact kNN at raw-image feature widths (d = 784, MNIST-sized n)."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlearning_amd import _hip
rng = np.random.default_rng(0)
for n, d, k in [(20000, 784, 11), (70000, 784, 11), (70000, 256, 11), (70000, 784, 31)]:
    # clustered data with pixel-like range: 10 centres, unit-scale noise
    c = rng.normal(size=(10, d)) * 2.0
    X = c[rng.integers(0, 10, n)] + rng.normal(size=(n, d))
    for i in range(2):
        t0 = time.perf_counter(); ind, dist = _hip.knn_bruteforce(X, k); wall = time.perf_counter() - t0
    st = _hip.knn_stats()
    fl = 2.0 * n * n * st['dpa']
    # spot check 64 rows against numpy
    rows = rng.integers(0, n, 64)
    D2 = ((X[rows][:, None, :] - X[None, :, :]) ** 2).sum(-1)
    ok = np.array_equal(np.argsort(D2, axis=1, kind='stable')[:, :k], ind[rows])
    print('n=%d d=%d k=%d: tile %.1f ms (%.1f TFLOP/s = %.1f%% of 157.3) rerank %.1f ms fallback rows %d (%.1f ms) wall %.2f s  spot-check %s'
          % (n, d, k, st['tile_ms'], fl / st['tile_ms'] / 1e9, fl / st['tile_ms'] / 1e9 / 157.3 * 100, st['rerank_ms'], st['fallback_rows'],
             st.get('fallback_ms', -1), wall, ok), flush=True)
