"""Developer probe: which part of bench.py's single-GPU path crashes under rocprofv3 --kernel-trace (round 3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
mode = sys.argv[1]
if 'req' in mode:
    _hip.require_device()
max_iter = int(sys.argv[2])
labels = bench.load_labels(70000)
X = bench.make_features(labels)
if 'warm4096' in mode:
    gl.weightmatrix.knn(X[:4096], 10)
W = gl.weightmatrix.knn(X, 10)
if 'build4' in mode:
    for _ in range(3):
        W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
n = W.shape[0]
model = gl.ssl.poisson(W, solver='gradient_descent')
dev, aux = model._operators()
src, k = gl.ssl._poisson_source(n, ti, labels[ti])
v0 = np.zeros(n); v0[ti] = 1; v0 /= v0.sum()
print('creating sweep, max_iter', max_iter, flush=True)
sw = _hip.Sweep(dev, k, min_iter=50, max_iter=max_iter, use_hipgraph='nograph' not in mode)
sw.set_problem(aux['D'] * src, v0 / aux['deg'], aux['deg'], aux['vinf'])
print('first run', flush=True)
print(sw.run(), flush=True)
nrun = 3000 if 'many' in mode else 5
for it in range(nrun):
    if it % 5 == 4:
        print('run', it, flush=True)
    if 'launches' in mode:
        sw.launches()
    if 'devsync' in mode:
        _hip.check(_hip.load().glx_device_synchronize(), 'sync')
    sw.run()
print('fetch', flush=True)
u = sw.fetch()
print('ok', mode, max_iter, float(np.abs(u).sum()), flush=True)
