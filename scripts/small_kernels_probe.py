"""A mix of the secondary paths (graph build, resident fit_predict, Laplace exact / tolerance, randomwalk, p-Laplace, reweighting) for
rocprofv3 --kernel-trace --stats: which small kernels take longer than they should?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
labels = bench.load_labels(70000); X = bench.make_features(labels)
for _ in range(3):
    W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=5, seed=0)
m = gl.ssl.poisson(W, solver='gradient_descent')
for _ in range(10):
    m.fit_predict(ti, labels[ti])
lap = gl.ssl.laplace(W)
for _ in range(3):
    lap.fit_predict(ti, labels[ti])
lapt = gl.ssl.laplace(W, reduce='tree')
for _ in range(3):
    lapt.fit_predict(ti, labels[ti])
rw = gl.ssl.randomwalk(W)
for _ in range(2):
    rw.fit_predict(ti, labels[ti])
G = gl.graph(W)
r = G.page_rank()
print('done')
