#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03as
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
GLX_FUZZ_SCALE=3 timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_scale3.log 2>&1; echo "exit $?"; grep -n "passed\|failed" $O/pytest_scale3.log | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q -p no:randomly > $O/pytest_again.log 2>&1; echo "exit $?"; grep -n "passed\|failed" $O/pytest_again.log | tail -3
