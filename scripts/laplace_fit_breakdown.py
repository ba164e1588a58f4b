"""Developer probe: steady-state ssl.laplace fit at config 3 (n = 60000, k = 20), host pieces against the device solve."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import utils, ssl as gssl
lab, X = bench.config3_data()
W = gl.weightmatrix.knn(X, 20)
ti = gl.trainsets.generate(lab, rate=10, seed=0)
tl = lab[ti]
for reduce in ('exact', 'tree'):
    m = gl.ssl.laplace(W, reduce=reduce)
    m.fit(ti, tl)
    best = [1e9] * 6
    for _ in range(7):
        t0 = time.perf_counter()
        L, Mv, dev = m._full_system(); t1 = time.perf_counter()
        k = len(np.unique(tl)); F = utils.labels_to_onehot(tl, k)
        rows, b = gssl._neg_columns_times_rows(L, m._full_csc(L), ti, F)
        keep = ~np.isin(rows, ti); rows, b = rows[keep], b[keep]; vals = Mv[rows, None] * b; t2 = time.perf_counter()
        u, its, _ = dev.cg_groups_rows(rows, vals, k, masks=[ti], out_scale=Mv, tol=m.tol, reduce=reduce); t3 = time.perf_counter()
        u[ti, :] = F; t4 = time.perf_counter()
        m.fit(ti, tl); t5 = time.perf_counter()
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            best[i] = min(best[i], v * 1e3)
    print('reduce=%-5s: operator lookup (fingerprint) %.2f ms | rhs rows %.2f ms (%d rows) | cg_groups_rows (%d iterations) %.2f ms | labelled rows %.2f ms | whole fit %.2f ms'
          % (reduce, best[0], best[1], len(rows), int(its[0]), best[2], best[3], best[4]))
