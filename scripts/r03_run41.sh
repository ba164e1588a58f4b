#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/knn_api_big.py 4e6 2>&1 | tail -6
timeout 1500 python scripts/knn_api_big.py 1e7 2>&1 | tail -6
