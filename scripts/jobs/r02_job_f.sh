#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest_all.log; cat $O/pytest_all.log | cut -c1-400
timeout 600 python scripts/dist_probe.py 2>&1 | grep "us/sweep" > $O/dist_probe.log; cat $O/dist_probe.log
