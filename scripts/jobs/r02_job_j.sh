#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest_all.log; cat $O/pytest_all.log | cut -c1-300
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-3000
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-scale > $GRAFT_REPO_ROOT/$O/bench_prof.log 2>&1; f=$(find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_stats.csv" | head -1); head -8 $f; cp $f $GRAFT_REPO_ROOT/$O/kernel_stats.csv; tail -1 $GRAFT_REPO_ROOT/$O/bench_prof.log | cut -c1-600
