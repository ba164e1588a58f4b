#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03an
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_knn.py tests/test_gpu_scale.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest_knn.log | tail -5
timeout 900 python scripts/knn_api_big.py 4e6 2>&1 | tail -5
timeout 1500 python scripts/knn_api_big.py 1e7 2>&1 | tail -5
