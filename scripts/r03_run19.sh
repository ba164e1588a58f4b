#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03t
mkdir -p $O
for rep in 1 2; do for fl in "-DGLX_FULL_CHUNKS=0" "-DGLX_FULL_CHUNKS=1" "-DGLX_FULL_CHUNKS=2" "-DGLX_FULL_CHUNKS=2 -DGLX_OFF32=1" "-DGLX_FULL_CHUNKS=0 -DGLX_OFF32=1"; do
  export GLX_CXXFLAGS="$fl"
  python -m graphlearning_amd._build > /dev/null 2>&1
  timeout 300 python scripts/persist_probe.py --big 1000000 --cache /tmp/knn_1e6.npz --reps 40 2>&1 | grep "float64\|float32\|sha" | sed "s/^/[$fl] /"
done; done | tee $O/full_chunks.log
unset GLX_CXXFLAGS
python -m graphlearning_amd._build > /dev/null 2>&1
