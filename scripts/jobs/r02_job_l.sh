#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PROBE_ONLY_D64=1
for g in 0 1; do for b in 0 1; do
  GLX_CXXFLAGS="-DKNN_GROUP_GUARD=$g -DKNN_BRANCHFREE_STAGE=$b" python -m graphlearning_amd._build > /dev/null 2>&1
  echo "== GROUP_GUARD=$g BRANCHFREE_STAGE=$b"; GLX_CXXFLAGS="-DKNN_GROUP_GUARD=$g -DKNN_BRANCHFREE_STAGE=$b" python scripts/knn_filter_probe.py big 2>&1 | grep "bf16:" | cut -c1-150
done; done
