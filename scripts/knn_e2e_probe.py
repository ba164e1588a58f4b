"""Developer probe: where the wall time of weightmatrix.knn(X, 10) goes at config 2."""
import numpy as np, sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
from graphlearning_amd import _hip
labels = bench.load_labels(70000); X = bench.make_features(labels)
for i in range(6):
    t0 = time.perf_counter(); W = gl.weightmatrix.knn(X, 10); print('call %d: weightmatrix.knn(X, 10) %.1f ms' % (i, (time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); ind, dist = gl.weightmatrix.knnsearch(X, 11); t1 = time.perf_counter()
print('knnsearch wall %.1f ms; device stats %s' % ((t1 - t0) * 1e3, _hip.knn_stats()))
t0 = time.perf_counter(); W = gl.weightmatrix.knn(None, 10, knn_data=(ind, dist)); t1 = time.perf_counter()
print('knn(knn_data) wall %.1f ms' % ((t1 - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); W = gl.weightmatrix.knn(X, 10); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(12); print(s.getvalue()[:2500])
