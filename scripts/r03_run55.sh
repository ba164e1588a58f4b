#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
GLX_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 2> /tmp/d1.err | head -c 1500; echo; tail -3 /tmp/d1.err
timeout 600 python bench.py --gpus 1 --config 4 --n 3e5 --steps 2 --warmup 1 2> /tmp/d2.err | head -c 700; echo; grep "config 4" /tmp/d2.err | tail -3
