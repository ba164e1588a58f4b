"""Developer probe: twenty times weightmatrix.knn + the first fit_predict on the fresh graph (config 2), for an API / kernel / copy trace
(rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats): where the host-array paths spend their time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl

labels = bench.load_labels(70000)
X = bench.make_features(labels)
ti = gl.trainsets.generate(labels, rate=1, seed=0)
for _ in range(int(os.environ.get('REPS', '23'))):
    W = gl.weightmatrix.knn(X, 10)
    gl.ssl.poisson(W, solver='gradient_descent').fit_predict(ti, labels[ti])
