#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_dist.py tests/test_gpu_round2.py -m gpu -x -q -s 2>&1 | tail -45 > $O/pytest.log; cat $O/pytest.log | cut -c1-400
timeout 600 python scripts/dist_probe.py > $O/dist_probe.log 2>&1; tail -12 $O/dist_probe.log | cut -c1-300
