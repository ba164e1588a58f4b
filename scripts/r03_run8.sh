#!/bin/bash
cd /root/repo
O=gpurun_out/r03h
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for l in "" "32 128" "40 160" "48 192" "64 256" "96 384"; do
  if [ -z "$l" ]; then unset GLX_SELL_L1 GLX_SELL_L4; else set -- $l; export GLX_SELL_L1=$1 GLX_SELL_L4=$2; fi
  timeout 300 python scripts/persist_probe.py --big 1000000 --cache /tmp/knn_1e6.npz --reps 40 2>&1 | grep GLX_PERSIST
done | tee $O/l1_probe.log
unset GLX_SELL_L1 GLX_SELL_L4
for l in "" "48 192" "96 384"; do
  if [ -z "$l" ]; then unset GLX_SELL_L1 GLX_SELL_L4; else set -- $l; export GLX_SELL_L1=$1 GLX_SELL_L4=$2; fi
  timeout 900 python scripts/order_probe.py 10000000 --cache /tmp/knn_1e7.npz --orders rcm --reps 2 --T 10 2>&1 | grep "^order" | sed "s/^/L1\/L4=$l /"
done | tee $O/l1_probe_1e7.log
