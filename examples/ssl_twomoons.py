"""Two-moons plumbing example (mirrors reference examples/ssl_twomoons.py, seeded, no plotting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sklearn.datasets as datasets
import graphlearning_amd as gl

X, labels = datasets.make_moons(n_samples=500, noise=0.1, random_state=0)
W = gl.weightmatrix.knn(X, 10)
train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
train_labels = labels[train_ind]
for model in [gl.ssl.laplace(W), gl.ssl.poisson(W), gl.ssl.poisson(W, solver='gradient_descent'),
              gl.ssl.poisson_mbo(W, gl.utils.class_priors(labels)), gl.ssl.randomwalk(W)]:
    pred_labels = model.fit_predict(train_ind, train_labels)
    accuracy = gl.ssl.ssl_accuracy(pred_labels, labels, train_ind)
    print('%s: %.2f%%' % (model.name, accuracy))
