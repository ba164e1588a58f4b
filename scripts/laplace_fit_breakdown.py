"""Developer probe: steady-state ssl.laplace fit at config 3 (n = 60000, k = 20), host pieces against the device solve."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
rng = np.random.default_rng(1)
lab = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'cifar_labels.npz'))['labels'] if os.path.exists(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'cifar_labels.npz')) else rng.integers(0, 10, size=60000)
lab = np.asarray(lab[:60000], dtype=np.int64)
X = rng.normal(size=(10, 32))[lab] * 1.2 + rng.normal(size=(60000, 32))
W = gl.weightmatrix.knn(X, 20)
ti = gl.trainsets.generate(lab, rate=10, seed=0)
for reduce in ('exact', 'tree'):
    m = gl.ssl.laplace(W, reduce=reduce)
    m.fit(ti, lab[ti])
    L, Mv, dev = m._full_system()
    best = [1e9] * 4
    for _ in range(5):
        t0 = time.perf_counter()
        F, B, k = m._rhs(L, Mv, ti, lab[ti]); t1 = time.perf_counter()
        x, its, _ = dev.cg_groups(B, k, tol=m.tol, masks=[ti], reduce=reduce); t2 = time.perf_counter()
        u = m._assemble(x, Mv, ti, F); t3 = time.perf_counter()
        m.fit(ti, lab[ti]); t4 = time.perf_counter()
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            best[i] = min(best[i], v * 1e3)
    print('reduce=%-5s: rhs %.2f ms | cg_groups (%d iterations) %.2f ms | assemble %.2f ms | whole fit %.2f ms' % (reduce, best[0], int(its[0]), best[1], best[2], best[3]))
