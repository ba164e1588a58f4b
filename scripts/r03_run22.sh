#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03w
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GLX_TIMING=1 timeout 120 python scripts/first_fit_breakdown.py > $O/first_fit.log 2>&1; grep "\[glx\] plan\|\[glx\] locality\|total\|Sweep" $O/first_fit.log | tail -14
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log
