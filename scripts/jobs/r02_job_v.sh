#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | cut -c1-200
timeout 600 python bench.py > $O/bench.log 2>&1; grep '^{"metric"' $O/bench.log | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
