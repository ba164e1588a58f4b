"""Mirrors reference examples/plaplace.py: p-Laplace interpolation of boundary values on a random
geometric graph.  (The reference example builds an epsilon-ball graph, which is outside this package;
a kNN graph of the same points is used instead.)  fast=False selects the Jacobi iteration of the
reference's C extension, which runs on the GPU; the reference's default fast=True is a sequential
Gauss-Seidel sweep and is not provided."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphlearning_amd as gl

X = np.random.default_rng(0).random((int(1e4), 2))
x, y = X[:, 0], X[:, 1]
W = gl.weightmatrix.knn(X, 12)
G = gl.graph(W)
eps = 0.02
bdy_set = (x < eps) | (x > 1 - eps) | (y < eps) | (y > 1 - eps)
bdy_val = (x - 0.5) ** 2 + (y - 0.5) ** 2
t0 = time.perf_counter()
u = G.plaplace(bdy_set, bdy_val[bdy_set], p=10, fast=False)
print('p-Laplace (p=10) on %d vertices: %d Jacobi iterations in %.2f s; interior mean %.4f, range [%.4f, %.4f]'
      % (len(x), G.plaplace_iters, time.perf_counter() - t0, u[~bdy_set].mean(), u.min(), u.max()))
print('PageRank: largest entries at', np.argsort(-G.page_rank())[:5], 'after', G.page_rank_iters, 'sweeps')
