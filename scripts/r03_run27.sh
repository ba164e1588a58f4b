#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ab
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest_knn.log | tail -8
timeout 900 python scripts/knn_cells_probe.py > $O/cells_probe.txt 2>&1; grep "^n=" $O/cells_probe.txt
GLX_KNN_CELL_STATS=1 timeout 1500 python bench.py --config 4 --n 1e7 --steps 2 --warmup 1 > $O/config4_1e7.json 2> $O/config4_1e7.err; grep "config 4\|glx\]" $O/config4_1e7.err | tail -20; head -c 3000 $O/config4_1e7.json; echo
