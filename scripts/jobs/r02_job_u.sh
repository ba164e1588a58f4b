#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PROBE_ONLY_D64=1
for fl in "-DKNN_NACC=1" "-DKNN_NACC=2"; do
  export GLX_CXXFLAGS="$fl"
  python -m graphlearning_amd._build > /dev/null 2>&1
  echo "== $fl  ($(cat graphlearning_amd/libglx.hash))"; timeout 200 python scripts/knn_filter_probe.py big 2>&1 | grep "bf16:" | cut -c1-150
done
