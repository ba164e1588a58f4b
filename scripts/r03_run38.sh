#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
GLX_CG_BLOCKED=1 GLX_TIMING=1 timeout 600 python scripts/cg_probe.py 2>&1 | grep "iterations in\|blocked reference" | head -3
