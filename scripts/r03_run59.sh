#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03bc
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
timeout 300 python scripts/cg_probe.py 2>&1 | grep "iterations in\|laplace" | tail -4
timeout 300 python scripts/configs_report.py 2>&1 | grep "config 3" | tail -3
