#!/bin/bash
mkdir -p gpurun_out/r6c11
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c11
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "skip_epilogue or b_stationary" > $O/kernels.log 2>&1; echo "kernels exit $?"; tail -2 $O/kernels.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "c2" > $O/full.log 2>&1; echo "fullsize exit $?"; tail -2 $O/full.log
timeout 300 python tools/gemm_bench.py --rounds 3 2>/dev/null | grep fused
WN_EPI_BWD_SPLIT=0 timeout 300 python tools/gemm_bench.py --rounds 3 2>/dev/null | grep fused
timeout 1500 python tools/ab_vars.py --vars "nosplit:WN_EPI_BWD_SPLIT=0" --reps 3 --tag split > $O/ab.txt 2>&1; tail -3 $O/ab.txt
