#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03at
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python scripts/order_probe.py 300000 --orders rcm,pairs --dtype f32 --cache /tmp/knn3e5.npz 2>&1 | grep "order\|pairs\|fp32" | tee $O/pairs_3e5.txt
timeout 2400 python scripts/order_probe.py 300000 --orders rcm,pairs --dtype f64 --cache /tmp/knn3e5.npz 2>&1 | grep "order\|pairs\|fp32" | tee -a $O/pairs_3e5.txt
