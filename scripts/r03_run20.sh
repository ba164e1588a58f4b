#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03u
mkdir -p $O
for rep in 1 2; do for fl in "-DKNN_PACKED_SELECT=0" "-DKNN_PACKED_SELECT=1"; do
  export GLX_CXXFLAGS="$fl"
  python -m graphlearning_amd._build > /dev/null 2>&1
  timeout 300 python scripts/knn_probe.py 2>&1 | tail -1 | sed "s/^/[$fl] config 2: /"
  timeout 300 python scripts/knn_filter_probe.py 2>&1 | grep "bf16:" | head -8 | sed "s/^/[$fl] /"
done; done | tee $O/knn_packed.log
unset GLX_CXXFLAGS
python -m graphlearning_amd._build > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_knn.py -x -q 2>&1 | tail -2
