#!/bin/bash
mkdir -p gpurun_out/r6c9
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c9
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -p no:cacheprovider -k "skip_epilogue" > $O/kernels.log 2>&1; echo "kernels exit $?"
tail -n 14 $O/kernels.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "golden or G1 or forward or fused" > $O/parity.log 2>&1; echo "parity exit $?"
tail -n 3 $O/parity.log
timeout 1500 python tools/ab_vars.py --vars "nofb:WN_EPI_FUSED_BWD=0 nof:WN_EPI_FUSED_BWD=0;WN_EPI_FUSED=0" --reps 3 --tag epib > $O/ab.txt 2>&1
cp gpurun_out/ab_vars_epib.json $O/
cat $O/ab.txt | tail -5
