#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ap
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests/test_gpu_fuzz.py -x -q -k "clustered" > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error\|assert" $O/pytest.log | tail -8
GLX_FUZZ_SCALE=4 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -x -q -k "clustered" > $O/pytest4.log 2>&1; grep -n "passed\|failed\|Error\|error\|assert" $O/pytest4.log | tail -8
