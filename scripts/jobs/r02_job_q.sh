#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | cut -c1-300
python scripts/knn_filter_probe.py big > $O/knn_filter.log 2>&1; grep -v "^W\|amdgpu" $O/knn_filter.log | cut -c1-200
timeout 300 python scripts/knn_stage_probe.py 2>&1 | grep call | tail -4
