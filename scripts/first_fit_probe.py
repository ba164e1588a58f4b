"""The first fit_predict of a fresh ssl.poisson(gradient_descent) model in a process (bench.py: graph_build.fresh_fit_predict_all_ms[0])
against the following ones: cProfile of the first and of the third call.  Usage: GLX_TIMING=1 python scripts/first_fit_probe.py"""
import cProfile, pstats, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
_hip.require_device()
labels = bench.load_labels(70000)
X = bench.make_features(labels)
gl.weightmatrix.knn(X[:4096], 10)
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
for rep in range(3):
    W = gl.weightmatrix.knn(X, 10)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    m = gl.ssl.poisson(W, solver='gradient_descent')
    pred = m.fit_predict(ti, labels[ti])
    pr.disable()
    dt = (time.perf_counter() - t0) * 1e3
    print('--- fresh model %d: %.2f ms' % (rep, dt), flush=True)
    if rep in (0, 2):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(12)
        print('\n'.join(l[:160] for l in s.getvalue().split('\n')[4:26]), flush=True)
