"""Where the wall time of a tolerance-mode ssl.laplace fit at config 3 goes outside its kernels: cProfile of 20 steady-state fits
(cumulative time per call of the host functions), beside the wall time per fit."""
import cProfile, pstats, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl

lab3, X3 = bench.config3_data()
W3 = gl.weightmatrix.knn(X3, 20)
ti3 = gl.trainsets.generate(lab3, rate=10, seed=0)
m = gl.ssl.laplace(W3, reduce='tree')
for _ in range(3):
    m.fit(ti3, lab3[ti3])
ts = []
for _ in range(20):
    t0 = time.perf_counter(); m.fit(ti3, lab3[ti3]); ts.append((time.perf_counter() - t0) * 1e3)
print('fit: median %.3f ms, min %.3f ms, %d iterations' % (float(np.median(ts)), min(ts), m.num_iter))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    m.fit(ti3, lab3[ti3])
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print('\n'.join(l[:170] for l in s.getvalue().split('\n')))
