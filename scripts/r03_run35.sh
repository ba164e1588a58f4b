#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03aj
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_knn.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest_knn.log | tail -8
timeout 300 python scripts/knn_host_breakdown.py 2>&1 | tail -6
timeout 300 python scripts/knn_filter_probe.py 2>&1 | grep "bf16" | head -6
