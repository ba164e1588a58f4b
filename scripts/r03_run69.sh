#!/bin/bash
# packed fma in the selection, with the norms already in registers (config 3 and n = 1e6 are the shapes that use the fma form)
for v in "-DKNN_PACKED_SELECT=1" "-DKNN_PACKED_SELECT=0"; do
  echo "== $v"
  GLX_CXXFLAGS="$v" python -m graphlearning_amd._build > /dev/null 2>&1
  export GLX_CXXFLAGS="$v"
  python scripts/knn_host_breakdown_c3.py 2>&1 | grep knnsearch
  python scripts/knn_big_breakdown.py 2>&1 | tail -1
  unset GLX_CXXFLAGS
done
