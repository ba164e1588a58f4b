"""PoissonMBO fit at config 5 (70 000 vertices): wall time of a steady-state fit, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=5, seed=0)
m = gl.ssl.poisson_mbo(W, gl.utils.class_priors(labels), solver='gradient_descent', Ns=40, T=20)
m.fit(ti, labels[ti])
for _ in range(3):
    t0 = time.perf_counter(); u = m.fit(ti, labels[ti]); dt = time.perf_counter() - t0
    print('poisson_mbo fit %.1f ms' % (dt * 1e3), flush=True)
