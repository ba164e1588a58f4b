#!/bin/bash
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
for e in ssl_twomoons ssl_mnist_shaped ssl_trials plaplace; do echo "== $e"; timeout 300 python examples/$e.py 2>&1 | grep -v "^RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4; done
echo "== ssl_multi_gpu (1 rank)"; timeout 300 python examples/ssl_multi_gpu.py 2>&1 | grep -v "^RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
GLX_KNN_CELL_STATS=1 timeout 1500 python bench.py --config 4 --n 1e7 --steps 2 --warmup 1 > gpurun_out/r03ao_config4.json 2> gpurun_out/r03ao_config4.err; grep "config 4" gpurun_out/r03ao_config4.err | tail -9
