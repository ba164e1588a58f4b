"""Developer probe: the GLX_TIMING stamps of one config-2 search (n = 70000, d = 20, k = 11) after warm-up."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import graphlearning_amd as gl

X = bench.make_features(bench.load_labels(70000))
for _ in range(3):
    gl.weightmatrix.knnsearch(X, 11)
os.environ['GLX_TIMING'] = '1'
for _ in range(2):
    gl.weightmatrix.knnsearch(X, 11)
    print('----', flush=True)
