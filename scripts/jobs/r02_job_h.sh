#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python bench.py --config 4 --n 1e7 --steps 2 --warmup 1 > $O/config4_1e7.log 2>&1; tail -3 $O/config4_1e7.log | cut -c1-2500
timeout 300 python scripts/knn_e2e_probe.py > $O/knn_e2e.log 2>&1; head -30 $O/knn_e2e.log | cut -c1-200
