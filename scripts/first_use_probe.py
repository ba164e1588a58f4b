"""First-use latency in a FRESH process at config 2: the first four weightmatrix.knn calls (GLX_TIMING=1 adds the library's stage
timers on stderr), then a fresh model's first fit_predict and its repeats.  Usage: [GLX_TIMING=1] python scripts/first_use_probe.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t0 = time.perf_counter()
import graphlearning_amd as gl
from graphlearning_amd import _hip
labels = bench.load_labels(70000)
X = bench.make_features(labels)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
t1 = time.perf_counter()
_hip.require_device()
t2 = time.perf_counter()
print('import + data %.1f ms, require_device (library load, runtime start) %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
for i in range(4):
    sys.stderr.write('--- weightmatrix.knn call %d\n' % (i + 1))
    t = time.perf_counter()
    W = gl.weightmatrix.knn(X, 10)
    print('weightmatrix.knn call %d: %.2f ms' % (i + 1, (time.perf_counter() - t) * 1e3), flush=True)
for i in range(3):
    sys.stderr.write('--- fresh model %d\n' % (i + 1))
    W = gl.weightmatrix.knn(X, 10)
    t = time.perf_counter()
    m = gl.ssl.poisson(W, solver='gradient_descent')
    ta = time.perf_counter()
    pred = m.fit_predict(ti, labels[ti])
    tb = time.perf_counter()
    pred = m.fit_predict(ti, labels[ti])
    tc = time.perf_counter()
    print('fresh graph %d: model %.2f ms, first fit_predict %.2f ms, second %.2f ms' % (i + 1, (ta - t) * 1e3, (tb - ta) * 1e3, (tc - tb) * 1e3), flush=True)

# ---- the bench's sequence: a 4096-row warm-up, then the first full-size call, phase by phase
if len(sys.argv) > 1 and sys.argv[1] == 'phases':
    from graphlearning_amd import utils
    rng = np.random.default_rng(3)
    for name, Y in (('X2 (new data, same size: arrays recycled?)', X + 0.0), ('80000 rows (new size class)', np.concatenate([X, X[:10000] + 0.5]))):
        tt = [time.perf_counter()]
        res = _hip.KnnResult(Y, 11, want_order=True); tt.append(time.perf_counter())
        order = res.order(); tt.append(time.perf_counter())
        Wy = res.to_csr(11, kernel='gaussian', sym=1); tt.append(time.perf_counter())
        res.close(); tt.append(time.perf_counter())
        fp = utils.symmetric_fingerprint(Wy); tt.append(time.perf_counter())
        print('%s: search %.2f | order %.2f | to_csr %.2f | close %.2f | fingerprint %.2f ms' % ((name,) + tuple((b - a) * 1e3 for a, b in zip(tt[:-1], tt[1:]))), flush=True)
    # a fresh model's first fit_predict, call by call (wrappers around the product path's own calls)
    from graphlearning_amd import ssl as ssl_mod
    acc = {}

    def wrap(obj, name, label):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t) * 1e3
        setattr(obj, name, g)
    wrap(ssl_mod, '_poisson_operator_symmetric', 'P = D^-1 W^T (host)')
    wrap(_hip.DeviceGraph, '__init__', 'DeviceGraph')
    wrap(_hip.Sweep, '__init__', 'Sweep create (plan, buffers)')
    wrap(_hip.Sweep, 'set_vectors', 'set_vectors')
    wrap(_hip.Sweep, 'set_problem_rows', 'set_problem_rows')
    wrap(_hip.Sweep, 'run', 'run (capture + sweeps)')
    wrap(_hip.Sweep, 'project', 'project')
    wrap(utils, 'symmetric_fingerprint', 'fingerprint')
    wrap(utils, 'known_symmetric', 'known_symmetric')
    for i in range(3):
        Wf = gl.weightmatrix.knn(X, 10)
        acc.clear()
        t = time.perf_counter()
        m = gl.ssl.poisson(Wf, solver='gradient_descent')
        pred = m.fit_predict(ti, labels[ti])
        tot = (time.perf_counter() - t) * 1e3
        print('fresh fit_predict %.2f ms: ' % tot + ' | '.join('%s %.2f' % kv for kv in acc.items()) + ' | other %.2f' % (tot - sum(v for k, v in acc.items() if k != 'known_symmetric' or True)), flush=True)
    # where the rest of a fresh fit_predict goes (cProfile, cumulative)
    import cProfile, pstats, io
    Wf = gl.weightmatrix.knn(X, 10)
    pr = cProfile.Profile()
    pr.enable()
    m = gl.ssl.poisson(Wf, solver='gradient_descent')
    pred = m.fit_predict(ti, labels[ti])
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(45)
    print(out.getvalue()[:9000])
