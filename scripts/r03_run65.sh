#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03bi
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
timeout 900 python bench.py --config 4 --n 1e6 --steps 2 --warmup 1 2> $O/c4.err | head -c 300; echo; grep "config 4" $O/c4.err | tail -4
