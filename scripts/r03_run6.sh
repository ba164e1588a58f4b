#!/bin/bash
cd /root/repo
O=gpurun_out/r03f
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cap in "" 12 16 20 24 32; do
  if [ -z "$cap" ]; then unset GLX_SELL_CAP; else export GLX_SELL_CAP=$cap; fi
  timeout 300 python scripts/persist_probe.py --big 1000000 --cache /tmp/knn_1e6.npz --reps 40 2>&1 | grep GLX_PERSIST
done | tee $O/cap_probe.log
unset GLX_SELL_CAP
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py -x -q > $O/pytest_default.log 2>&1; echo "pytest exit $?" >> $O/pytest_default.log; tail -3 $O/pytest_default.log
GLX_SELL_CAP=16 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py -x -q > $O/pytest_cap16.log 2>&1; echo "pytest exit $?" >> $O/pytest_cap16.log; tail -3 $O/pytest_cap16.log
