"""MNIST-shaped example (mirrors reference examples/ssl_mnist.py / poisson_mbo.py): the real MNIST
label vector with synthetic 20-d features (the MNIST-VAE kNN blob of the reference is not
redistributable here), k = 10 kNN graph on the GPU, one label per class."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphlearning_amd as gl

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
labels = np.load(os.path.join(root, 'tests', 'golden', 'MNIST_labels.npz'))['labels'].astype(np.int64)
rng = np.random.default_rng(0)
centers = rng.normal(size=(10, 20)) * 2.0
X = centers[labels] + rng.normal(size=(70000, 20))
t0 = time.perf_counter()
W = gl.weightmatrix.knn(X, 10)
print('kNN graph: nnz=%d in %.2f s' % (W.nnz, time.perf_counter() - t0))
train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
train_labels = labels[train_ind]
models = [gl.ssl.laplace(W), gl.ssl.poisson(W), gl.ssl.poisson(W, solver='gradient_descent'),
          gl.ssl.poisson(W, solver='gradient_descent', use_cuda=True),
          gl.ssl.poisson_mbo(W, gl.utils.class_priors(labels), solver='gradient_descent')]
for model in models:
    t0 = time.perf_counter()
    pred_labels = model.fit_predict(train_ind, train_labels)
    dt = time.perf_counter() - t0
    print('%-28s %.2f%%  (%.3f s)' % (model.name, gl.ssl.ssl_accuracy(pred_labels, labels, train_ind), dt))
