"""Developer probe: the sweep of config 2 under the library's own reverse Cuthill-McKee order vs scipy's (which the distributed
planner uses): microseconds per launch (captured head of 50 sweeps, HIP events)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip, dist as gdist
from scipy import sparse

labels = bench.load_labels(70000)
W = gl.weightmatrix.knn(bench.make_features(labels), 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
n = W.shape[0]
P, deg, dinv = gl.ssl._poisson_operator_symmetric(W)
src, k = gl.ssl._poisson_source(n, ti, labels[ti])
v0 = np.zeros(n); v0[ti] = 1; v0 /= v0.sum()
Db = sparse.spdiags(dinv, 0, n, n).tocsr() * src
orders = {'library rcm': None, 'scipy rcm': gdist.locality_order(P).astype(np.int32)}
rng = np.random.default_rng(0)
orders['scipy rcm reversed'] = orders['scipy rcm'][::-1].copy()
for rep, balance in ((0, '0'), (1, '1'), (2, '0'), (3, '1')):
    os.environ['GLX_XCD_BALANCE'] = balance
    for name, order in orders.items():
        dev = _hip.DeviceGraph(P, dtype=np.float64, order=order)
        sw = _hip.Sweep(dev, k, min_iter=50, max_iter=50, use_hipgraph=True)
        sw.set_problem(Db, v0 / deg, deg, deg / np.sum(deg))
        for _ in range(3):
            sw.run()
        tot = 0.0
        for _ in range(40):
            tot += sw.run()[1]
        perm = dev.order()
        pos = np.empty(n, dtype=np.int64); pos[perm] = np.arange(n)
        rows = np.repeat(np.arange(n), np.diff(P.indptr))
        lens = np.diff(P.indptr)[perm]                      # row lengths along the order
        if balance == '1':
            work = np.cumsum(lens + 3)
            cuts = [0] + [int(np.searchsorted(work, work[-1] * x // 8, side='left')) + 0 for x in range(1, 8)] + [n]
        else:
            cuts = [n * x // 8 for x in range(9)]
        per = [int(lens[cuts[x]:cuts[x + 1]].sum()) for x in range(8)]
        print('%-20s balance=%s: %.2f us/launch; entries per XCD range max/mean %.3f (rows %d..%d); %s'
              % (name, balance, tot * 1e3 / (40 * 50), max(per) * 8 / sum(per), min(np.diff(cuts)), max(np.diff(cuts)), dev.info()), flush=True)
        sw.close()
        dev.close()
