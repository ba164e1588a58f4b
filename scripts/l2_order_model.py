"""scripts/l2_model.py for several vertex orders of the same graph (which order would a 4 MB LRU L2 per XCD like best?): the search's
chained cells (what the operator takes), scipy's reverse Cuthill-McKee of the whole graph, RCM inside every chained cell, the caller's
order.  Usage: python scripts/l2_order_model.py N"""
import os, sys, ctypes, subprocess, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graphlearning_amd as gl
from scipy.sparse.csgraph import reverse_cuthill_mckee
n = int(sys.argv[1])
so = '/tmp/lru_sim.so'
subprocess.run(['gcc', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'scripts', 'probes', 'lru_sim.c')], check=True)
lib = ctypes.CDLL(so)
lib.lru_misses.restype = ctypes.c_int64
lib.lru_misses.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
rng = np.random.default_rng(2)
labels = rng.integers(0, 10, size=n)
X = (rng.normal(size=(10, 64)) * 4)[labels] + rng.normal(size=(n, 64))
W = gl.weightmatrix.knn(X, 10)
lens = np.diff(W.indptr).astype(np.int64)


def misses(perm, sigma=32768):
    perm = np.asarray(perm, dtype=np.int64)
    inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
    work = np.cumsum(lens[perm] + 3)
    cuts = [0] + [int(np.searchsorted(work, work[-1] * x / 8, side='left')) for x in range(1, 8)] + [n]
    tot = 0
    for x in range(8):
        rows = perm[cuts[x]:cuts[x + 1]]
        rows = np.concatenate([w[np.argsort(-lens[w], kind='stable')] for w in (rows[a:a + sigma] for a in range(0, len(rows), sigma))])
        ln = lens[rows]
        idx = np.repeat(W.indptr[rows].astype(np.int64) - np.concatenate([[0], np.cumsum(ln)[:-1]]), ln) + np.arange(int(ln.sum()))
        stream = np.ascontiguousarray(inv[W.indices[idx]].astype(np.int32))
        tot += lib.lru_misses(stream.ctypes.data, len(stream), n, 32768)
    return tot


cell = np.asarray(W._glx_order, dtype=np.int64)
print('chained cells (shipped): %d misses of %d gathers' % (misses(cell), W.nnz), flush=True)
print('caller order           : %d' % misses(np.arange(n)), flush=True)
t0 = time.time(); rcm = np.asarray(reverse_cuthill_mckee(W, symmetric_mode=True), dtype=np.int64)
print('RCM of the whole graph : %d   (%.1f s on the host)' % (misses(rcm), time.time() - t0), flush=True)
# RCM inside stretches of the cell order (keeps the coarse geometry, orders every stretch by the graph)
for stretch in (32768, 131072):
    parts = []
    t0 = time.time()
    for a in range(0, n, stretch):
        ids = cell[a:a + stretch]
        sub = W[ids][:, ids]
        parts.append(ids[np.asarray(reverse_cuthill_mckee(sub, symmetric_mode=True), dtype=np.int64)])
    print('RCM inside stretches of %6d rows of the cell order: %d   (%.1f s)' % (stretch, misses(np.concatenate(parts)), time.time() - t0), flush=True)
