#!/bin/bash
cd /root/repo
O=gpurun_out/r03k
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for form in 1 2 3 1 2 3; do
  export GLX_CXXFLAGS="-DGLX_LOOP_FORM=$form"
  python -m graphlearning_amd._build > /dev/null 2>&1
  timeout 300 python scripts/persist_probe.py --big 1000000 --cache /tmp/knn_1e6.npz --reps 40 2>&1 | grep "float64\|float32\|sha" | sed "s/^/form=$form /"
done | tee $O/loop_form.log
unset GLX_CXXFLAGS
python -m graphlearning_amd._build > /dev/null 2>&1
