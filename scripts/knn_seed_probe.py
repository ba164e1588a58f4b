"""Seeding pre-pass of the kNN tile kernel (knn_seed_kernel) A/B: GLX_KNN_SEED = 0 (off) / 4 / 8 / 16 on the config-2 features
(70 000 x 20), locality-sorted 64-d blobs at 3e5 and 1e6.  Prints tile ms (pre-pass included), fallback rows, and checks the
lists are identical to the unseeded search."""
import numpy as np, sys, os, hashlib, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphlearning_amd import _hip

def blobs(n, d, C=10, seed=2):
    g = np.random.default_rng(seed)
    lab = g.integers(0, C, size=n)
    cen = g.normal(size=(C, d)) * 4
    return cen[lab] + g.normal(size=(n, d))

cases = [('config 2 (70000 x 20)', bench.make_features(bench.load_labels(70000)), 11)]
for n in (300000, 1000000):
    cases.append(('blobs %d x 64' % n, blobs(n, 64), 11))
cases.append(('blobs 300000 x 3', blobs(300000, 3), 11))
cases.append(('blobs 200000 x 32 k=26', blobs(200000, 32), 26))
for name, X, k in cases:
    ref = None
    for sub in (0, 4, 8, 16):
        os.environ['GLX_KNN_SEED'] = str(sub)
        best = 1e9
        for i in range(3):
            ind, dist = _hip.knn_bruteforce(X, k)
            st = _hip.knn_stats()
            best = min(best, st['tile_ms'])
        if ref is None:
            ref = (ind.copy(), dist.copy())
        same = bool(np.array_equal(ind, ref[0]) and np.array_equal(dist, ref[1]))
        print('%-26s seed sample 1/%-2d (ran %d): tile %.3f ms  rerank %.3f ms  fallback rows %d  identical to unseeded: %s' % (
            name, sub, st['seed_sample'], best, st['rerank_ms'], st['fallback_rows'], same), flush=True)
