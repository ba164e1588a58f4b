"""Device time of the stages of the kNN search (tile kernel incl. set-up, re-rank, fallback) through the product path
(KnnResult: rows reordered by chained cells) at config 2 (70 000 x 20, k = 11), config 3 (60 000 x 32, k = 21) and a d = 64
shape (200 000 x 64, k = 11): the minimum of `reps` searches, and the digest of the lists (equal across builds = same result).
Usage: python scripts/knn_stage_times.py [reps]"""
import hashlib
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphlearning_amd import _hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12


def shapes():
    labels = bench.load_labels(70000)
    yield 'config2 70000x20 k=11', bench.make_features(labels), 11
    rng = np.random.default_rng(1)
    lab3 = rng.integers(0, 10, size=60000)
    yield 'config3 60000x32 k=21', rng.normal(size=(10, 32))[lab3] * 1.2 + rng.normal(size=(60000, 32)), 21
    rng = np.random.default_rng(2)
    lab4 = rng.integers(0, 10, size=200000)
    yield 'blobs 200000x64 k=11', rng.normal(size=(10, 64))[lab4] * 4.0 + rng.normal(size=(200000, 64)), 11


only = sys.argv[2] if len(sys.argv) > 2 else ''
for name, X, k in shapes():
    if only and not name.startswith(only):
        continue
    best = None
    for r in range(reps):
        res = _hip.KnnResult(X, k, want_order=True)
        st = _hip.knn_stats()
        if r == 0:
            J, D = res.lists()
            sha = hashlib.sha256(np.ascontiguousarray(J).tobytes()).hexdigest()[:12]
        res.close()
        t = (st['tile_ms'], st['rerank_ms'], st['fallback_ms'])
        best = t if best is None else tuple(min(a, b) for a, b in zip(best, t))
    print('%-24s tile %.3f ms  rerank %.3f ms  fallback %.3f ms (%d rows)  lists %s' % (name, best[0], best[1], best[2], st['fallback_rows'], sha), flush=True)
