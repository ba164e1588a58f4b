#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PROBE_ONLY_D64=1
echo "== shipped build"; python scripts/knn_filter_probe.py big 2>&1 | grep "bf16:" | cut -c1-150
echo "== again"; python scripts/knn_filter_probe.py big 2>&1 | grep "bf16:" | cut -c1-150
python -m graphlearning_amd._build --force > /dev/null 2>&1
echo "== rebuilt on the box"; python scripts/knn_filter_probe.py big 2>&1 | grep "bf16:" | cut -c1-150
rocm-smi --showclocks --showpower 2>/dev/null | head -20
