#!/bin/bash
mkdir -p gpurun_out/r6c13
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c13
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "wgrad" > $O/kernels.log 2>&1; echo "kernels exit $?"; tail -2 $O/kernels.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "golden or G1 or forward or fused" > $O/parity.log 2>&1; echo "parity exit $?"; tail -2 $O/parity.log
timeout 1500 python tools/ab_vars.py --vars "shipped" --reps 2 --tag wg > $O/ab.txt 2>&1; tail -3 $O/ab.txt
