#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ak
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 0 1; do echo "GLX_CG_BLOCKED=$b"; GLX_CG_BLOCKED=$b GLX_TIMING=1 timeout 600 python scripts/cg_probe.py 2>&1 | grep "iterations in\|blocked reference" | tail -12; done
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_trials.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
