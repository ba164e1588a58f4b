#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03aa
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest_knn.log | tail -8
timeout 900 python scripts/knn_cells_probe.py > $O/cells_probe.txt 2>&1; tail -12 $O/cells_probe.txt
timeout 300 python scripts/knn_probe.py 2>&1 | tail -2
