"""Mirrors reference examples/ssl_trials.py on an MNIST-shaped graph: several learners run over the
same list of training sets with ssl_trials (results/<tag>..._accuracy.csv, reference file format);
trials_statistics gives the mean / std per label rate.  Poisson CG trials are stacked into one
device solve (bit-identical per trial); the other learners go trial after trial on the resident operator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphlearning_amd as gl

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
labels = np.load(os.path.join(root, 'tests', 'golden', 'MNIST_labels.npz'))['labels'].astype(np.int64)
rng = np.random.default_rng(0)
X = (rng.normal(size=(10, 20)) * 2.0)[labels] + rng.normal(size=(70000, 20))
W = gl.weightmatrix.knn(X, 10)
trainsets = gl.trainsets.generate(labels, rate=np.array([[1], [2], [3], [4], [5]]), num_trials=10, seed=0)   # 50 training sets

model_list = [gl.ssl.laplace(W),
              gl.ssl.laplace(W, reweighting='wnll'),
              gl.ssl.laplace(W, reweighting='poisson'),
              gl.ssl.poisson(W),
              gl.ssl.poisson(W, solver='gradient_descent')]
tag = 'mnistshaped_k10_'
for model in model_list:
    t0 = time.perf_counter()
    model.ssl_trials(trainsets, labels, tag=tag, overwrite=True)
    print('%s: %d trials in %.2f s' % (model.name, len(trainsets), time.perf_counter() - t0))
    num_train, acc_mean, acc_std, num_trials = model.trials_statistics(tag=tag)
    for m, a, s in zip(num_train, acc_mean[:, 0], acc_std[:, 0]):
        print('   %3d labels: %.2f +- %.2f  (%d trials)' % (m, a, s, num_trials))
