#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 0 1; do echo "GLX_CG_BLOCKED=$b"; GLX_CG_BLOCKED=$b GLX_TIMING=1 timeout 600 python scripts/cg_probe.py 2>&1 | grep "iterations in\|blocked reference\|laplace" | tail -6; done
