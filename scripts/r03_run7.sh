#!/bin/bash
cd /root/repo
O=gpurun_out/r03g
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cap in "" 20 24 28 32 40; do
  if [ -z "$cap" ]; then unset GLX_SELL_CAP; else export GLX_SELL_CAP=$cap; fi
  timeout 300 python scripts/persist_probe.py --big 1000000 --cache /tmp/knn_1e6.npz --reps 40 2>&1 | grep GLX_PERSIST
done | tee $O/cap_probe.log
unset GLX_SELL_CAP
