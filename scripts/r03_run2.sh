#!/bin/bash
# round 3, GPU call 2: dist forms, scale model v2, persistent-kernel A/B, vertex-order experiment, knn host breakdown
cd /root/repo
O=gpurun_out/r03b
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q > $O/pytest_dist.log 2>&1; echo "pytest exit $?" >> $O/pytest_dist.log
tail -4 $O/pytest_dist.log
timeout 300 python scripts/dist_probe.py > $O/dist_probe.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" $O/dist_probe.log | tail -40
timeout 900 python scripts/scale_model.py --n4 2e6 --out $O/scale_model.json > $O/scale_model.log 2>&1; grep scale_model $O/scale_model.log | tail -30
for k in 1 2 3 4 6; do GLX_PERSIST=$k timeout 300 python scripts/persist_probe.py --big 1000000 --cache /tmp/knn_1e6.npz --reps 40 2>&1 | grep GLX_PERSIST; done | tee $O/persist_probe.log
GLX_PERSIST=1 timeout 300 python scripts/persist_probe.py --reps 40 2>&1 | grep GLX_PERSIST | tee -a $O/persist_probe.log
timeout 900 python scripts/order_probe.py 1000000 --cache /tmp/knn_1e6.npz 2>&1 | grep "^order" | tee $O/order_probe.log
GLX_TIMING=1 timeout 120 python scripts/knn_host_breakdown.py > $O/knn_host.log 2>&1; tail -40 $O/knn_host.log
GLX_TIMING=1 timeout 120 python - > $O/knn_first_call.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '/root/repo')
import bench, graphlearning_amd as gl
X = bench.make_features(bench.load_labels(70000))
gl.weightmatrix.knn(X[:4096], 10)
print('---- first full-size call', file=sys.stderr)
t0 = time.perf_counter(); W = gl.weightmatrix.knn(X, 10); t1 = time.perf_counter()
print('first call at 70k after a 4096-row warm-up: %.2f ms' % ((t1 - t0) * 1e3), file=sys.stderr)
t0 = time.perf_counter(); W = gl.weightmatrix.knn(X, 10); t1 = time.perf_counter()
print('second call: %.2f ms' % ((t1 - t0) * 1e3), file=sys.stderr)
PY
grep -A60 "first full-size" $O/knn_first_call.log | head -80
