#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | cut -c1-300
timeout 120 python scripts/knn_filter_probe.py 2>&1 | grep -E "bf16:|identical" | cut -c1-170
