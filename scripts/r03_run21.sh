#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03v
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GLX_TIMING=1 timeout 120 python scripts/first_fit_breakdown.py > $O/first_fit.log 2>&1; grep "\[glx\] plan\|\[glx\] locality\|total" $O/first_fit.log | tail -14
timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
