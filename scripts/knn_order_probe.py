"""Developer probe: the exact search on config-4-shaped data in random order and in the coarse locality order of the sharded
pipeline (dist_build.coarse_locality_order): tile time and rows that take the exact fallback."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlearning_amd import _hip, dist_build
from graphlearning_amd.dist_bench import config4_features
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
X, lab = config4_features(n)
for name, Y in (('random order', X), ('coarse locality order', np.ascontiguousarray(X[dist_build.coarse_locality_order(X, ncells=64, seed=0)]))):
    t0 = time.perf_counter()
    J, D = _hip.knn_bruteforce(Y, 11)
    st = _hip.knn_stats()
    print('%-22s n=%d: wall %.2f s, tile %.1f ms, fallback rows %d (%.1f ms), nsplit %d' % (name, n, time.perf_counter() - t0, st['tile_ms'], st['fallback_rows'], st['fallback_ms'], st['nsplit']), flush=True)
