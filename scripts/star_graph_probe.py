"""Developer probe: a vertex adjacent to all others (row of n - 1 entries) through the sweep: set-up time, sweep time, bit-parity."""
import os, sys, time
import numpy as np
from scipy import sparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlearning_amd import _hip
from oracle import gl_oracle as orc
rng = np.random.default_rng(0)
n = 200000
# ring + random edges + one vertex adjacent to everyone (a star): symmetric weights
i = np.arange(n); j = (i + 1) % n
r1 = rng.integers(0, n, size=4 * n); r2 = rng.integers(0, n, size=4 * n)
rows = np.concatenate([i, j, r1, r2, np.zeros(n - 1, dtype=np.int64), np.arange(1, n)])
cols = np.concatenate([j, i, r2, r1, np.arange(1, n), np.zeros(n - 1, dtype=np.int64)])
vals = np.concatenate([np.ones(2 * n), rng.random(4 * n), rng.random(4 * n), np.ones(2 * (n - 1)) * 0.01])
W = sparse.csr_matrix((vals, (rows, cols)), shape=(n, n)); W = (W + W.T) * 0.5; W.setdiag(0); W.eliminate_zeros(); W = sparse.csr_matrix(W)
print('n=%d nnz=%d max row %d' % (n, W.nnz, np.diff(W.indptr).max()), flush=True)
ti = rng.choice(n, size=40, replace=False); lab = rng.integers(0, 4, size=n); lab[ti[:4]] = np.arange(4)
import graphlearning_amd as gl
t0 = time.perf_counter()
m = gl.ssl.poisson(W, solver='gradient_descent', min_iter=20, max_iter=20)
u = m.fit(ti, lab[ti]); t1 = time.perf_counter()
u2 = m.fit(ti, lab[ti]); t2 = time.perf_counter()
u_ref = orc.poisson_gd(W, ti, lab[ti], min_iter=20, max_iter=20)
print('first fit %.1f ms, second %.2f ms (20 sweeps), bit-identical to the oracle: %s' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, np.array_equal(np.asarray(u2), u_ref)))
dev, aux = m._operators()
print(dev.info())
