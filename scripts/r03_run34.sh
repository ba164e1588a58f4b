#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ai
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GLX_TIMING=1 timeout 300 python scripts/knn_probe.py 2>&1 | grep "glx\] knn" | tail -9
GLX_TIMING=1 timeout 300 python - 2>&1 <<'PY' | grep "glx\] knn\|call" | tail -24
import numpy as np, sys, time
sys.path.insert(0, '/root/repo')
from graphlearning_amd import _hip
g = np.random.default_rng(2)
n, d = 1000000, 64
lab = g.integers(0, 10, size=n); cen = g.normal(size=(10, d)) * 4
X = cen[lab] + g.normal(size=(n, d))
for i in range(2):
    t0 = time.perf_counter(); J, D = _hip.knn_bruteforce(X, 11); print('call %.3f s' % (time.perf_counter() - t0), file=sys.stderr)
PY
timeout 300 python scripts/knn_host_breakdown.py 2>&1 | tail -8
