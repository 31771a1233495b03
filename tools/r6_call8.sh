#!/bin/bash
mkdir -p gpurun_out/r6c8
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c8
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "skip_epilogue" > $O/kernels.log 2>&1; echo "kernels exit $?"
timeout 1500 python tools/ab_vars.py --vars "unfused:WN_EPI_FUSED=0 s0:WN_EPI_STAGGER=0 s1800:WN_EPI_STAGGER=1800 s2700:WN_EPI_STAGGER=2700 s4500:WN_EPI_STAGGER=4500" --reps 3 --tag stag > $O/ab.txt 2>&1
cp gpurun_out/ab_vars_stag.json $O/
cat $O/ab.txt | tail -8
