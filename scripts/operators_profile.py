"""Developer probe: cProfile of ssl.poisson._operators() on a fresh config-2 graph (the one-off part of a first fit)."""
import os, sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
labels = bench.load_labels(70000); X = bench.make_features(labels)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
W = gl.weightmatrix.knn(X, 10)
m0 = gl.ssl.poisson(W, solver='gradient_descent'); m0.fit_predict(ti, labels[ti])
for rep in range(3):
    W2 = gl.weightmatrix.knn(X, 10)
    m = gl.ssl.poisson(W2, solver='gradient_descent')
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable(); dev, aux = m._operators(); pr.disable(); t1 = time.perf_counter()
    print('operators %.2f ms' % ((t1 - t0) * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:4500])
