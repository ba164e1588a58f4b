#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03bg
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
timeout 600 python scripts/knn_big_breakdown.py 2>&1 | grep "^search" | tail -2
timeout 600 python scripts/knn_cells_probe.py 2>&1 | grep "^n=" | head -3
