"""Developer probe: what the vertex renumbering costs and buys at config 2 (GLX_REORDER = 1 RCM (default), 2 plain BFS, 0 none):
host time of the order, sweep time of a fit with T = 50 sweeps (HIP events)."""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import bench
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    labels = bench.load_labels(70000); X = bench.make_features(labels)
    W = gl.weightmatrix.knn(X, 10)
    ti = gl.trainsets.generate(labels, rate=1, seed=0)
    m = gl.ssl.poisson(W, solver='gradient_descent')
    t0 = time.perf_counter(); m.fit_predict(ti, labels[ti]); t1 = time.perf_counter()
    dev, aux = m._operators()
    best = 1e9
    for _ in range(20):
        T, ms = aux['sweep'].run()
        best = min(best, ms)
    print('GLX_REORDER=%s: first fit_predict %.1f ms, %d sweeps in %.3f ms = %.2f us per sweep' % (os.environ.get('GLX_REORDER', '(default)'), (t1 - t0) * 1e3, T, best, best / T * 1e3))
else:
    for v in ('1', '2', '0'):
        env = dict(os.environ, GLX_REORDER=v, GLX_TIMING='1')
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
        print('\n'.join(l for l in (r.stdout + r.stderr).splitlines() if 'locality order' in l or 'GLX_REORDER' in l))
