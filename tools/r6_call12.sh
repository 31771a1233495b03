#!/bin/bash
mkdir -p gpurun_out/r6c12
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c12
timeout 1500 python tools/ab_vars.py --vars "c2048:WN_EPI_WGRAD_CHUNKS=2048,2048,2048 c2048_3264:WN_EPI_WGRAD_CHUNKS=2048,2048,3264 c3264:WN_EPI_WGRAD_CHUNKS=3264,3264,3264 c4096:WN_EPI_WGRAD_CHUNKS=4096,4096,4096 c2048_6528:WN_EPI_WGRAD_CHUNKS=2048,2048,6528" --reps 3 --tag chunks2 > $O/ab2.txt 2>&1; tail -7 $O/ab2.txt
cp gpurun_out/ab_vars_chunks2.json $O/
