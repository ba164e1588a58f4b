"""Developer probe: the one-off costs of a first ssl.poisson(gradient_descent) fit on a fresh graph at config 2, call by call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip, ssl as ssl_mod, utils
labels = bench.load_labels(70000); X = bench.make_features(labels)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
W0 = gl.weightmatrix.knn(X, 10)
m0 = gl.ssl.poisson(W0, solver='gradient_descent'); m0.fit_predict(ti, labels[ti])       # process warm-up
for rep in range(3):
    W = gl.weightmatrix.knn(X, 10)
    t = [time.perf_counter()]
    P, deg, dinv = ssl_mod._poisson_operator_symmetric(W); t.append(time.perf_counter())
    dev = _hip.DeviceGraph(P); t.append(time.perf_counter())
    sw = _hip.Sweep(dev, 10, 50, 1000, True); t.append(time.perf_counter())
    sw.set_vectors(deg, deg / np.sum(deg)); t.append(time.perf_counter())
    onehot = utils.labels_to_onehot(labels[ti], 10)
    sw.set_problem_rows(ti, dinv[ti, None] * (onehot - np.mean(onehot, axis=0)), (1.0 / len(ti)) / deg[ti], 0.0); t.append(time.perf_counter())
    sw.run(); t.append(time.perf_counter())
    sw.run(); t.append(time.perf_counter())
    lab, w, err, steps = sw.project(); t.append(time.perf_counter())
    sw.close(); dev.close()
    names = ['P = D^-1 W^T (numpy)', 'DeviceGraph', 'Sweep (order, plan, buffers)', 'set_vectors', 'set_problem_rows', 'first run (capture)', 'second run', 'project']
    print(' | '.join('%s %.2f' % (n, (b - a) * 1e3) for n, a, b in zip(names, t[:-1], t[1:])), '| total %.1f ms' % ((t[-1] - t[0]) * 1e3))
