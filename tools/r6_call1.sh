#!/bin/bash
# round 6, first GPU call: the B-stationary dZ product (tests, alone, inside the step), wide2's tile-boundary timing builds, host capacity
mkdir -p gpurun_out/r6c1
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -p no:cacheprovider -k "chan_gemm" > $O/kernels.log 2>&1; echo "kernels exit $?" | tee -a $O/summary.txt
timeout 300 python tools/gemm_bench.py --ab bst --rounds 3 > $O/gemm_ab_bst.json 2> $O/gemm_ab_bst.err; echo "gemm ab exit $?" | tee -a $O/summary.txt
WN_GEMM_BST=0 WAVENET_HIP_LIB=tools/_var_gwt7.so timeout 300 python tools/gemm_bench.py --rounds 3 > $O/gemm_gwt7.json 2> $O/gemm_gwt7.err; echo "gwt7 exit $?" | tee -a $O/summary.txt
WN_GEMM_BST=0 WAVENET_HIP_LIB=tools/_var_gwt8.so timeout 300 python tools/gemm_bench.py --rounds 3 > $O/gemm_gwt8.json 2> $O/gemm_gwt8.err; echo "gwt8 exit $?" | tee -a $O/summary.txt
timeout 900 python tools/ab_vars.py --vars "nobst:WN_GEMM_BST=0" --reps 3 --tag bst > $O/ab_bst.txt 2>&1; echo "ab bst exit $?" | tee -a $O/summary.txt
cp gpurun_out/ab_vars_bst.json $O/ 2>/dev/null
timeout 900 python tools/ab_vars.py --env "WN_GEMM_BST=0" --vars "gwt7 gwt8" --reps 3 --tag gwt > $O/ab_gwt.txt 2>&1; echo "ab gwt exit $?" | tee -a $O/summary.txt
cp gpurun_out/ab_vars_gwt.json $O/ 2>/dev/null
timeout 900 python tools/host_capacity.py > $O/host_capacity.json 2> $O/host_capacity.err; echo "host exit $?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -p no:cacheprovider -k "c2" > $O/fullsize_c2.log 2>&1; echo "fullsize exit $?" | tee -a $O/summary.txt
tail -n 5 $O/kernels.log; cat $O/gemm_ab_bst.json; cat $O/gemm_gwt7.json $O/gemm_gwt8.json; cat $O/ab_bst.txt | tail -5; cat $O/ab_gwt.txt | tail -6; tail -3 $O/host_capacity.err; tail -n 5 $O/fullsize_c2.log
