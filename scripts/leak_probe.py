"""Developer probe: device-memory growth over repeated fits / graph builds (should be flat after warm-up)."""
import numpy as np, sys, os, gc
import torch                     # first: one HIP runtime for torch's mem_get_info and libglx
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphlearning_amd as gl
rng = np.random.default_rng(0)
labels = rng.integers(0, 6, 8000)
X = (rng.normal(size=(6, 12)) * 2.0)[labels] + rng.normal(size=(8000, 12))
def used():
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 2**20
base = None
for rep in range(6):
    for _ in range(10):
        W = gl.weightmatrix.knn(X, 10)
        ti = gl.trainsets.generate(labels, rate=2, seed=rep)
        for m in (gl.ssl.poisson(W), gl.ssl.poisson(W, solver='gradient_descent'), gl.ssl.laplace(W), gl.ssl.laplace(W, reweighting='poisson'),
                  gl.ssl.randomwalk(W), gl.ssl.poisson_mbo(W, gl.utils.class_priors(labels), solver='gradient_descent')):
            m.fit_predict(ti, labels[ti])
        G = gl.graph(W); G.page_rank(); G.plaplace(ti, labels[ti].astype(float), 4, fast=False, max_num_it=200)
        del W, G, m
    gc.collect()
    u = used()
    base = u if base is None else base
    print('round %d: %.1f MiB in use (+%.1f since round 0)' % (rep, u, u - base), flush=True)
