"""Developer probe: batched Poisson CG trials on the config-2 graph (for rocprofv3 --kernel-trace --stats)."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
labels = bench.load_labels(70000); X = bench.make_features(labels)
W = gl.weightmatrix.knn(X, 10)
trials = [gl.trainsets.generate(labels, rate=1, seed=s) for s in range(nb)]      # rate 1: ~140 iterations each
model = gl.ssl.poisson(W)
for rep in range(2):
    t0 = time.perf_counter()
    out = model._fit_batch([(t, labels[t]) for t in trials]) if nb > 1 else [model.fit(trials[0], labels[trials[0]])]
    wall = time.perf_counter() - t0
its = model.num_iter if nb > 1 else [model.num_iter]
print('batch of %d: %.3f s, %.1f ms per trial, iterations %s, %.2f ms per iteration' % (nb, wall, wall / nb * 1e3, its, wall / max(its) * 1e3))
