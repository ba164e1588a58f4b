#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 0 1 0 1; do
echo "GLX_KNN_ORDER=$c"
GLX_KNN_ORDER=$c timeout 600 python - <<'PY' 2>&1 | grep -v "^RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -2
import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np
import graphlearning_amd as gl
rng = np.random.default_rng(2)
n = 1000000
labels = rng.integers(0, 10, size=n)
X = (rng.normal(size=(10, 64)) * 4)[labels] + rng.normal(size=(n, 64))
W0 = gl.weightmatrix.knn(X[:200000], 10)      # warm the library up
t0 = time.perf_counter(); W = gl.weightmatrix.knn(X, 10); t1 = time.perf_counter()
ti = gl.trainsets.generate(labels, rate=5, seed=0)
m = gl.ssl.poisson(W, solver='gradient_descent')
t2 = time.perf_counter(); pred = m.fit_predict(ti, labels[ti]); t3 = time.perf_counter()
t4 = time.perf_counter(); pred = m.fit_predict(ti, labels[ti]); t5 = time.perf_counter()
print('weightmatrix.knn %.3f s | first fit_predict %.3f s (%d sweeps) | second %.3f s | accuracy %.2f' % (t1 - t0, t3 - t2, m.num_iter, t5 - t4, 100 * np.mean(pred == labels)))
PY
done
