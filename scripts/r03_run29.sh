#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ad
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest_knn.log | tail -8
for c in 2 1 0; do echo "GLX_KNN_CAT=$c"; GLX_KNN_CAT=$c timeout 300 python scripts/knn_probe.py 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q > $O/pytest_full.log 2>&1; grep -n "passed\|failed" $O/pytest_full.log | tail -3
