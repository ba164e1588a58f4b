#!/bin/bash
# round 3, GPU call 1: parity suite, dist probe, scale model, bench (single vs one-rank distributed path)
cd /root/repo
mkdir -p gpurun_out/r03a
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
timeout 300 python scripts/dist_probe.py > gpurun_out/r03a/dist_probe.log 2>&1; tail -40 gpurun_out/r03a/dist_probe.log
timeout 1500 python scripts/scale_model.py --n4 2e6 --out gpurun_out/r03a/scale_model.json > gpurun_out/r03a/scale_model.log 2>&1; tail -30 gpurun_out/r03a/scale_model.log
timeout 300 python bench.py --no-scale > gpurun_out/r03a/bench_single.json 2> gpurun_out/r03a/bench_single.err; cat gpurun_out/r03a/bench_single.json | head -c 1500
GLX_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 > gpurun_out/r03a/bench_dist1.json 2> gpurun_out/r03a/bench_dist1.err; cat gpurun_out/r03a/bench_dist1.json | head -c 1500
