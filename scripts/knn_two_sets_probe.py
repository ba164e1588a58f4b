"""Tile-kernel time of the all-pairs bf16 filter (d = 20, k = 11) at two sizes and two ref-range counts: python scripts/knn_two_sets_probe.py"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphlearning_amd import _hip
for n in (70000, 120000):
    labels = bench.load_labels(n)
    X = bench.make_features(labels)
    for ns in (None, 8):
        best = 1e9
        with _hip.knn_options(nsplit=ns):
            for r in range(6):
                res = _hip.KnnResult(X, 11, want_order=True, clustered=0)
                st = _hip.knn_stats()
                if r == 0:
                    J, _ = res.lists()
                    sha = hashlib.sha256(np.ascontiguousarray(J).tobytes()).hexdigest()[:10]
                res.close()
                best = min(best, st['tile_ms'])
        print('n=%d nsplit=%s(%d): tile %.3f ms  rerank %.3f  lists %s' % (n, ns, st['nsplit'], best, st['rerank_ms'], sha), flush=True)
