#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ag
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest_knn.log | tail -8
timeout 900 python scripts/knn_cells_probe.py > $O/cells_probe.txt 2>&1; grep "^n=\|^uniform\|Error\|error" $O/cells_probe.txt
