"""What an LRU cache of the size of an XCD's L2 (4 MB = 32768 lines of 128 bytes) misses when it sees the sweep's neighbour gathers in the
order the plan issues them: the rows of the eight XCD ranges (equal work, graph.hip) in windows of 32768 rows sorted by decreasing
length, a row's neighbours in stored order.  Between the unlimited-cache bound (bench.distinct_line_bound) and the counters: if the
counters' misses are about this model's, the misses are the graph's (an expander inside every cluster), not the kernel's.
Usage: python scripts/l2_model.py N [--cache /tmp/knn.npz]   (no GPU needed once the kNN lists are cached; builds them on the GPU otherwise)"""
import os, sys, ctypes, subprocess, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import graphlearning_amd as gl
ap = argparse.ArgumentParser()
ap.add_argument('n', type=int)
ap.add_argument('--cache', default=None)
ap.add_argument('--sigma', type=int, default=32768)
a = ap.parse_args()
n = a.n
so = '/tmp/lru_sim.so'
subprocess.run(['gcc', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'scripts', 'probes', 'lru_sim.c')], check=True)
lib = ctypes.CDLL(so)
lib.lru_misses.restype = ctypes.c_int64
lib.lru_misses.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
rng = np.random.default_rng(2)
labels = rng.integers(0, 10, size=n)
centers = rng.normal(size=(10, 64)) * 4
X = centers[labels] + rng.normal(size=(n, 64))
W = gl.weightmatrix.knn(X, 10)
perm = np.asarray(W._glx_order, dtype=np.int64)            # the order the fp64 operator takes (ssl.poisson._operators)
inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
lens = np.diff(W.indptr).astype(np.int64)
work = np.cumsum(lens[perm] + 3)
cuts = [0] + [int(np.searchsorted(work, work[-1] * x / 8, side='left')) for x in range(1, 8)] + [n]
tot_miss, tot_acc, tot_distinct = 0, 0, 0
for x in range(8):
    rows = perm[cuts[x]:cuts[x + 1]]
    order = []
    for w0 in range(0, len(rows), a.sigma):
        win = rows[w0:w0 + a.sigma]
        order.append(win[np.argsort(-lens[win], kind='stable')])
    rows = np.concatenate(order)
    starts = W.indptr[rows].astype(np.int64)
    ln = lens[rows]
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(ln)[:-1]]), ln) + np.arange(int(ln.sum()))
    stream = np.ascontiguousarray(inv[W.indices[idx]].astype(np.int32))
    miss = lib.lru_misses(stream.ctypes.data, len(stream), n, 32768)
    tot_miss += miss; tot_acc += len(stream); tot_distinct += len(np.unique(stream))
print('n = %d: %d gathers per sweep; distinct records over the 8 ranges %d (unlimited cache); LRU of 32768 lines per XCD, gathers in plan order: %d misses '
      '(hit rate %.1f %%) = %.1f MB of 128-byte lines' % (n, tot_acc, tot_distinct, tot_miss, 100.0 * (1 - tot_miss / tot_acc), tot_miss * 128 / 1e6))
