#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PROBE_ONLY_D64=1
for fl in "-DKNN_BF16_NSUB=2" "-DKNN_BF16_NSUB=1" "-DKNN_BF16_NSUB=4" "-DKNN_GROUP_GUARD=0 -DKNN_BRANCHFREE_STAGE=0" "-DKNN_GROUP_GUARD=1 -DKNN_BRANCHFREE_STAGE=0" "-DKNN_GROUP_GUARD=0 -DKNN_BRANCHFREE_STAGE=1"; do
  export GLX_CXXFLAGS="$fl"
  python -m graphlearning_amd._build > /dev/null 2>&1
  echo "== $fl  ($(cat graphlearning_amd/libglx.hash))"; python scripts/knn_filter_probe.py big 2>&1 | grep "bf16:" | cut -c1-150
done
