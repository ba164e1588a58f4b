"""Developer probe: one exact kNN search at d=64, n=300000 (the command the kNN tile-kernel counter passes wrap)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlearning_amd import _hip
rng = np.random.default_rng(2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
lab = rng.integers(0, 10, size=n)
X = rng.normal(size=(10, 64))[lab] * 4.0 + rng.normal(size=(n, 64))
for _ in range(2):
    J, D = _hip.knn_bruteforce(X, 11)
print(_hip.knn_stats())
