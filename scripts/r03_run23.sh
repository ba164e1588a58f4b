#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03x
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/knn_seed_probe.py > $O/seed_probe.txt 2>&1; cat $O/seed_probe.txt | tail -30
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed" $O/pytest_knn.log | tail -3
