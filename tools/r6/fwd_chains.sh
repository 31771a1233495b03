#!/bin/bash
# the forward stack as two chains of launches (clips 0-3 / 4-7) on two streams against one chain
mkdir -p gpurun_out/r6
WN_FWD_CHAINS=2 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "c2_bench_geometry or c2_full_length" 2>&1 | tail -2
timeout 1200 python tools/ab_vars.py --vars "shipped:WN_FWD_CHAINS=2" --reps 4 --tag fwd_chains > gpurun_out/r6/fwd_chains_ab.log 2>&1; tail -4 gpurun_out/r6/fwd_chains_ab.log
