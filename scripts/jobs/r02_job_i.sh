#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest_all.log; cat $O/pytest_all.log | cut -c1-300
timeout 300 python scripts/knn_e2e_probe.py 2>&1 | head -9 | cut -c1-200
timeout 600 python scripts/configs_report.py 2>&1 | tail -9
timeout 300 python scripts/trials_probe.py 2>&1 | grep "^=="
timeout 300 python scripts/leak_probe.py 2>&1 | tail -4
