#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/knn_filter_probe.py big > $O/knn_filter.log 2>&1; cat $O/knn_filter.log | cut -c1-260
timeout 1200 python -m pytest tests/test_gpu_knn.py tests/test_gpu_fullsize.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-300
