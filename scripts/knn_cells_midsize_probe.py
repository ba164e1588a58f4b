"""Developer probe: the clustered search (library-formed cells) against all pairs at mid sizes, on data with and without clusters."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
rng = np.random.default_rng(5)
cases = {
    'config 2 (10 blobs, d=20, n=70000)': bench.make_features(bench.load_labels(70000)),
    'isotropic normal d=20 n=70000': rng.normal(size=(70000, 20)),
    'uniform cube d=8 n=70000': rng.random(size=(70000, 8)),
    'blobs d=20 n=32768': bench.make_features(bench.load_labels(70000))[:32768],
    'isotropic normal d=20 n=32768': rng.normal(size=(32768, 20)),
    'config 3 shape (10 blobs, d=32, n=60000, k=21)': (rng.normal(size=(10, 32)) * 1.2)[rng.integers(0, 10, size=60000)] + rng.normal(size=(60000, 32)),
}
for name, X in cases.items():
    k = 21 if 'k=21' in name else 11
    out = []
    for m in (0, 32, 64, 128):
        os.environ['GLX_KNN_CLUSTERED'] = str(m)
        for _ in range(2):
            gl.weightmatrix.knnsearch(X, k)
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter(); gl.weightmatrix.knnsearch(X, k); best = min(best, time.perf_counter() - t0)
        st = _hip.knn_stats()
        out.append('%d cells: %.2f ms (tile %.2f, fallback rows %d, visited %.0f %%)' % (m, best * 1e3, st['tile_ms'], st['fallback_rows'], 100 * st['visited_share']))
    print(name + ' | ' + ' | '.join(out), flush=True)
