"""Developer probe: the exact search on data sorted by class (and by a coordinate inside a class): rows that fall back / searches that repeat."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlearning_amd import _hip
rng = np.random.default_rng(3)
for (n, d, k, scale) in ((60000, 32, 21, 1.2), (70000, 20, 11, 2.0), (200000, 50, 31, 2.0), (100000, 8, 11, 1.0)):
    lab = np.sort(rng.integers(0, 10, size=n))
    X = rng.normal(size=(10, d))[lab] * scale + rng.normal(size=(n, d))
    # within a class, sort along the first principal direction: index order follows geometry as far as one coordinate can
    order = np.lexsort((X[:, 0], lab))
    for name, Y in (('class-sorted', X), ('class + coordinate sorted', np.ascontiguousarray(X[order]))):
        _hip.knn_bruteforce(Y, k)
        t0 = time.perf_counter(); J, D = _hip.knn_bruteforce(Y, k); w = time.perf_counter() - t0
        st = _hip.knn_stats()
        print('n=%d d=%d k=%d %-26s wall %.1f ms tile %.2f ms fallback rows %d escalated %d' % (n, d, k, name, w * 1e3, st['tile_ms'], st['fallback_rows'], st['escalated_rows']), flush=True)
