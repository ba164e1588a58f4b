"""Developer probe: what the checked transfers cost on the paths that move host arrays (round 6).  weightmatrix.knn and the first
fit_predict on a fresh graph at config 2 under glx_upload_set_mode 0 (staged + checked: the default), 1 (staged), 2 (direct copies from the
caller's memory: rounds 1-5), and the library's own stage timers of the search."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip

labels = bench.load_labels(70000)
X = bench.make_features(labels)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
tl = labels[ti]
for _ in range(3):
    W = gl.weightmatrix.knn(X, 10)
    gl.ssl.poisson(W, solver='gradient_descent').fit_predict(ti, tl)


def med(f, reps=15):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    return t[len(t) // 2], t[0]


for mode in (0, 1, 2, 0):
    _hip.upload_set_mode(mode)
    k = med(lambda: gl.weightmatrix.knn(X, 10))
    ft = []
    for _ in range(15):                       # one fresh graph at a time, as a user builds them (the page-locked result arrays recycle)
        Wf = gl.weightmatrix.knn(X, 10)
        t0 = time.perf_counter()
        gl.ssl.poisson(Wf, solver='gradient_descent').fit_predict(ti, tl)
        ft.append((time.perf_counter() - t0) * 1e3)
        del Wf
    ft.sort()
    f = (ft[len(ft) // 2], ft[0])
    m = gl.ssl.poisson(gl.weightmatrix.knn(X, 10), solver='gradient_descent'); m.fit(ti, tl)
    r = med(lambda: m.fit(ti, tl))
    print('transfer mode %d: weightmatrix.knn %.2f ms (min %.2f) | fresh fit_predict %.2f ms (min %.2f) | fit on the resident operator %.2f ms'
          % (mode, k[0], k[1], f[0], f[1], r[0]), flush=True)
_hip.upload_set_mode(0)
print('uploads checked / wrong / repaired / given up:', _hip.upload_stats())
