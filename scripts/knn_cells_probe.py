"""Cell-pruned exact kNN (glx_knn_cells_range) against the all-pairs search on config-4-shaped data: Gaussian blobs, d = 64, C = 10,
coarse geometric order with 64 cells (dist_build.coarse_locality_order).  Prints tile-kernel ms (pre-pass, bounds and runs included),
the visited share, and checks the lists are identical."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlearning_amd import _hip, dist_build

def blobs(n, d, C=10, seed=2):
    g = np.random.default_rng(seed)
    lab = g.integers(0, C, size=n)
    cen = g.normal(size=(C, d)) * 4
    return cen[lab] + g.normal(size=(n, d))

for n, d, ncells in [(300000, 64, 64), (1000000, 64, 64), (1000000, 64, 256), (1000000, 16, 256)]:
    X = blobs(n, d)
    perm, starts = dist_build.coarse_locality_order(X, ncells=ncells, seed=0, return_cells=True)
    X = np.ascontiguousarray(X[perm])
    t0 = time.perf_counter(); J0, D0 = _hip.knn_bruteforce(X, 11); t_all = time.perf_counter() - t0
    s0 = _hip.knn_stats()
    J0, D0 = np.array(J0), np.array(D0)
    t0 = time.perf_counter(); J1, D1 = _hip.knn_bruteforce(X, 11, cell_starts=starts); t_cell = time.perf_counter() - t0
    s1 = _hip.knn_stats()
    print('n=%d d=%d cells=%d: all pairs tile %.1f ms (call %.2f s, %d fallback rows) | cells tile %.1f ms (call %.2f s, %d fallback rows, '
          'visited %.1f %%, sample stride %d) | identical: %s' % (n, d, ncells, s0['tile_ms'], t_all, s0['fallback_rows'], s1['tile_ms'], t_cell,
          s1['fallback_rows'], 100 * s1['visited_share'], s1['seed_sample'], bool(np.array_equal(J0, J1) and np.array_equal(D0, D1))), flush=True)

# cells formed by the library on data in RANDOM order (glx_knn_clustered; what weightmatrix.knn does from 2^17 rows on)
for n, d in [(300000, 64), (1000000, 64), (1000000, 16)]:
    X = blobs(n, d)
    t0 = time.perf_counter(); J0, D0 = _hip.knn_bruteforce(X, 11, clustered=0); t_all = time.perf_counter() - t0
    s0 = _hip.knn_stats()
    J0, D0 = np.array(J0), np.array(D0)
    for rep in range(2):
        t0 = time.perf_counter(); J1, D1 = _hip.knn_bruteforce(X, 11); t_cl = time.perf_counter() - t0
    s1 = _hip.knn_stats()
    print('n=%d d=%d random order: all pairs call %.3f s (tile %.1f ms) | clustered (%d cells) call %.3f s (tile %.1f ms, visited %.1f %%) | identical: %s' % (
        n, d, t_all, s0['tile_ms'], s1['cells'], t_cl, s1['tile_ms'], 100 * s1['visited_share'], bool(np.array_equal(J0, J1) and np.array_equal(D0, D1))), flush=True)
Xu = np.random.default_rng(0).random((1000000, 16))
t0 = time.perf_counter(); J0, D0 = _hip.knn_bruteforce(Xu, 11, clustered=0); t_all = time.perf_counter() - t0
s0 = _hip.knn_stats(); J0 = np.array(J0)
t0 = time.perf_counter(); J1, D1 = _hip.knn_bruteforce(Xu, 11); t_cl = time.perf_counter() - t0
s1 = _hip.knn_stats()
print('uniform 1e6 x 16: all pairs call %.3f s (tile %.1f ms) | clustered (%d cells) call %.3f s (tile %.1f ms, visited %.1f %%) | identical: %s' % (
    t_all, s0['tile_ms'], s1['cells'], t_cl, s1['tile_ms'], 100 * s1['visited_share'], bool(np.array_equal(J0, J1))), flush=True)
