"""Developer probe: host-side setup costs (operator creation / plan build) at config 2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
for reorder in ('1', '0'):
    os.environ['GLX_REORDER'] = reorder
    t0 = time.perf_counter(); G = _hip.DeviceGraph(W); t1 = time.perf_counter()
    u = np.zeros((70000, 10)); G.spmm_bias(u); t2 = time.perf_counter()
    G.spmm_bias(u); t3 = time.perf_counter()
    print('GLX_REORDER=%s: create %.1f ms, first spmm (plan build) %.1f ms, second spmm %.1f ms' % (reorder, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
    G.close()
os.environ['GLX_REORDER'] = '1'
for name, mk in [('poisson GD', lambda: gl.ssl.poisson(W, solver='gradient_descent')), ('poisson CG', lambda: gl.ssl.poisson(W)), ('laplace', lambda: gl.ssl.laplace(W)),
                 ('poisson_mbo', lambda: gl.ssl.poisson_mbo(W, gl.utils.class_priors(labels), solver='gradient_descent'))]:
    m = mk()
    t0 = time.perf_counter(); m.fit(ti, labels[ti]); t1 = time.perf_counter(); m.fit(ti, labels[ti]); t2 = time.perf_counter()
    print('%-12s first fit %.1f ms, second fit %.1f ms' % (name, (t1-t0)*1e3, (t2-t1)*1e3))
