#!/bin/bash
mkdir -p gpurun_out/r6c6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c6
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "skip_epilogue" > $O/kernels.log 2>&1; echo "kernels exit $?"
tail -n 3 $O/kernels.log
timeout 1500 python tools/ab_vars.py --vars "unfused:WN_EPI_FUSED=0" --reps 3 --tag epi2 > $O/ab.txt 2>&1
cp gpurun_out/ab_vars_epi2.json $O/
cat $O/ab.txt | tail -4
