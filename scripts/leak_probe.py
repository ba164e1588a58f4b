"""Developer probe: device and host memory over repeated graph builds and fits (new graph, new models every round)."""
import os, sys, gc, resource
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
rng = np.random.default_rng(0)
lab = rng.integers(0, 10, size=40000)
C = rng.normal(size=(10, 16)) * 2.5


def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024


for r in range(41):
    X = C[lab] + rng.normal(size=(40000, 16))
    W = gl.weightmatrix.knn(X, 10)
    ti = gl.trainsets.generate(lab, rate=2, seed=r)
    for model in (gl.ssl.poisson(W, solver='gradient_descent'), gl.ssl.poisson(W), gl.ssl.laplace(W), gl.ssl.poisson_mbo(W, gl.utils.class_priors(lab), solver='gradient_descent', T=3)):
        model.fit_predict(ti, lab[ti])
    del model, W
    gc.collect()
    if r % 10 == 0:
        print('round %2d: device memory in use %.0f MiB, host peak RSS %.0f MiB' % ((r,) + used()), flush=True)
