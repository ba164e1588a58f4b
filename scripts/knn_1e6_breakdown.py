"""Where weightmatrix.knn(X, 10) spends its time at one GPU's share of config 4 (n = 10^6, d = 64): wall time of three calls, the
device stages of the search (knn_stats) and, with GLX_TIMING=1, the library's own stage stamps.
Usage: GLX_TIMING=1 python scripts/knn_1e6_breakdown.py [n]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(2)
labels = rng.integers(0, 10, size=n)
X = rng.normal(size=(10, 64))[labels] * 4 + rng.normal(size=(n, 64))
for r in range(3):
    t0 = time.perf_counter()
    W = gl.weightmatrix.knn(X, 10)
    dt = time.perf_counter() - t0
    st = _hip.knn_stats()
    print('call %d: %.1f ms wall; search kernels: tile %.2f ms, rerank %.2f ms, fallback %.2f ms (%d rows), cells %d, visited %.3f; nnz %d'
          % (r, dt * 1e3, st['tile_ms'], st['rerank_ms'], st['fallback_ms'], st['fallback_rows'], st['cells'], st['visited_share'], W.nnz), flush=True)
    del W
