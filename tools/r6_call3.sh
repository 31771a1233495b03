#!/bin/bash
mkdir -p gpurun_out/r6c3
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c3
timeout 1500 python tools/ab_vars.py --vars "nobst:WN_GEMM_BST=0 grid:WN_GEMM_BST=2 bst5:WN_GEMM_BST=1 bst5:WN_GEMM_BST=2" --reps 3 --tag bst2 > $O/ab.txt 2>&1
cp gpurun_out/ab_vars_bst2.json $O/
cat $O/ab.txt | tail -8
for g in 1 2; do WN_GEMM_BST=$g WAVENET_HIP_LIB=tools/_var_bst5.so timeout 300 python tools/gemm_bench.py --rounds 3 2>/dev/null | grep skipT; WN_GEMM_BST=$g timeout 300 python tools/gemm_bench.py --rounds 3 2>/dev/null | grep skipT; done
