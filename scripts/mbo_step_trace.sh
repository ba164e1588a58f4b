#!/bin/bash
# One outer step of PoissonMBO at config 5 on the device's clock: rocprofv3 --kernel-trace --memory-copy-trace over scripts/mbo_breakdown.py,
# then scripts/mbo_step_trace.py prints the kernels and copies between two thresholdings (the 40 heat sweeps folded into one line).
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/${1:-mbo_step}/t -o run -- python $R/scripts/mbo_breakdown.py 2>&1 | tail -1
python $R/scripts/mbo_step_trace.py $R/gpurun_out/${1:-mbo_step}/t
