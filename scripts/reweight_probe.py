import numpy as np, sys, os, time
sys.path.insert(0, '/root/repo')
import bench
import graphlearning_amd as gl
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=10, seed=0)
G = gl.graph(W)
for rep in range(2):
    t0 = time.perf_counter(); Wr = G.reweight(ti, method='poisson'); t = time.perf_counter() - t0
print('graph.reweight(poisson) at 70k: %.3f s' % t)
m = gl.ssl.laplace(W, reweighting='poisson')
t0 = time.perf_counter(); m.fit(ti, labels[ti]); print('laplace(reweighting=poisson).fit %.3f s, iters %s' % (time.perf_counter() - t0, m.num_iter))
