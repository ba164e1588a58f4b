"""Developer probe: the CG solves of configs 2 and 3 (exact and tolerance mode), wall time per fit and per iteration.
Run under `scripts/prof_run.py TAG --match cg -- python scripts/cg_configs_probe.py [c3|c2|all] [exact|tree|both]` for kernel stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl

which = sys.argv[1] if len(sys.argv) > 1 else 'all'
modes = sys.argv[2] if len(sys.argv) > 2 else 'both'
modes = ('exact', 'tree') if modes == 'both' else (modes,)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5


def timed(f, reps=reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = f(); ts.append(time.perf_counter() - t0)
    return min(ts), out


if which in ('c3', 'all'):
    lab3, X3 = bench.config3_data()
    W3 = gl.weightmatrix.knn(X3, 20)
    ti3 = gl.trainsets.generate(lab3, rate=10, seed=0)
    ref = None
    for mode in modes:
        m = gl.ssl.laplace(W3, reduce=mode)
        u = m.fit(ti3, lab3[ti3])
        t, u = timed(lambda: m.fit(ti3, lab3[ti3]))
        if ref is None:
            ref = u
        print('config 3 ssl.laplace(reduce=%s).fit: %.2f ms, %d CG iterations = %.1f us per iteration of wall time; max |u - u_first_mode| = %.2e'
              % (mode, t * 1e3, m.num_iter, t * 1e6 / m.num_iter, float(np.max(np.abs(u - ref)))))
if which in ('c2', 'all'):
    labels = bench.load_labels(70000); X = bench.make_features(labels)
    W = gl.weightmatrix.knn(X, 10)
    ti = gl.trainsets.generate(labels, rate=1, seed=0)
    m = gl.ssl.poisson(W)
    m.fit(ti, labels[ti])
    t, _ = timed(lambda: m.fit(ti, labels[ti]))
    print('config 2 ssl.poisson(conjugate_gradient).fit: %.2f ms, %d iterations = %.1f us per iteration' % (t * 1e3, m.num_iter, t * 1e6 / m.num_iter))
