"""PoissonMBO at config 5: where a steady-state fit's wall time goes -- the inner Poisson fit, the 20 x 40 heat sweeps, the 20
volume-constrained thresholdings -- call by call (wrappers around the product path's own calls)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
m = gl.ssl.poisson_mbo(W, gl.utils.class_priors(labels), solver='gradient_descent', Ns=40, mu=1, T=20)
m.fit(ti, labels[ti])
acc = {}


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc.setdefault(label, []).append((time.perf_counter() - t) * 1e3)
    setattr(obj, name, g)


wrap(_hip.Sweep, 'iterate', 'heat.iterate(40)')
wrap(_hip.Sweep, 'project', 'project')
wrap(_hip.Sweep, 'set_state', 'set_state')
wrap(_hip.Sweep, 'fetch', 'fetch')
wrap(_hip.Sweep, 'run', 'poisson sweeps (run)')
wrap(type(m.poisson_model), '_fit_only', 'inner poisson fit')
for rep in range(3):
    acc.clear()
    t0 = time.perf_counter(); m.fit(ti, labels[ti]); tot = (time.perf_counter() - t0) * 1e3
    print('fit %.2f ms: ' % tot + ' | '.join('%s x%d = %.2f ms (median %.3f)' % (k, len(v), sum(v), sorted(v)[len(v) // 2]) for k, v in acc.items())
          + ' | other %.2f' % (tot - sum(sum(v) for k, v in acc.items() if k != 'poisson sweeps (run)')), flush=True)
