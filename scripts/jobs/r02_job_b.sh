#!/bin/bash
# round-2 job B: new GPU tests + PMC evidence for the config-4-shaped sweeps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py -m gpu -x -q -s 2>&1 | tail -60 > $O/pytest_new.log; cat $O/pytest_new.log
PMC1="FETCH_SIZE"; PMC2="WRITE_SIZE"; PMC3="TCC_HIT_sum TCC_MISS_sum"; PMC4="TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; PMC5="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
python scripts/scale_probe.py 1000000 --cache /tmp/knn6.npz --reps 1 > /dev/null 2>&1
timeout 900 python scripts/prof_run.py r02b/scale_1e6 --match spmm_sell --pmc "$PMC1" --pmc "$PMC2" --pmc "$PMC3" --pmc "$PMC4" --pmc "$PMC5" -- python scripts/scale_probe.py 1000000 --cache /tmp/knn6.npz --reps 2
python scripts/scale_probe.py 10000000 --cache /tmp/knn7.npz --reps 1 --T 10 --dtype both > $O/scale_1e7_both.log 2>&1; cat $O/scale_1e7_both.log
timeout 1500 python scripts/prof_run.py r02b/scale_1e7 --match spmm_sell --pmc "$PMC1" --pmc "$PMC2" --pmc "$PMC3" --pmc "$PMC4" --pmc "$PMC5" -- python scripts/scale_probe.py 10000000 --cache /tmp/knn7.npz --reps 1 --T 20
