#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python bench.py --config 4 --n 1e7 --steps 2 --warmup 1 > $O/config4_1e7.log 2>&1; grep '^{"metric"' $O/config4_1e7.log | cut -c1-2200
