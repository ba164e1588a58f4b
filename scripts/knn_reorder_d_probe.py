"""Developer probe: rows in cell order in front of the all-pairs search (the default below 2^17 rows, d <= 128) against the caller's
order without (GLX_KNN_ORDER=0) and with the cell order worked out on the side (GLX_KNN_REORDER=0), over the feature counts of the bf16 filter."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
from graphlearning_amd import _hip
rng = np.random.default_rng(7)
for d in (8, 20, 32, 48, 64, 100, 128):
    for kind in ('blobs', 'isotropic'):
        n = 50000
        X = rng.normal(size=(n, d)) if kind == 'isotropic' else (rng.normal(size=(10, d)) * 2.0)[rng.integers(0, 10, size=n)] + rng.normal(size=(n, d))
        out = []
        for order in ('0', 'side', '1'):
            os.environ['GLX_KNN_ORDER'] = '0' if order == '0' else '1'
            os.environ['GLX_KNN_REORDER'] = '0' if order == 'side' else '1'
            for _ in range(2):
                _hip.knn_bruteforce(X, 11, want_order=(order != '0'))
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); _hip.knn_bruteforce(X, 11, want_order=(order != '0')); best = min(best, time.perf_counter() - t0)
            out.append('%s %.2f ms (tile %.2f)' % ({'0': 'caller order, no cell order', 'side': 'caller order + cell order on the side', '1': 'rows in cell order'}[order], best * 1e3, _hip.knn_stats()['tile_ms']))
        print('d=%d %s n=%d | %s' % (d, kind, n, ' | '.join(out)), flush=True)
