#!/bin/bash
mkdir -p gpurun_out/r6c10
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c10
timeout 1500 python tools/ab_vars.py --vars "o1:WN_EPI_BWD_ORDER=1 o2:WN_EPI_BWD_ORDER=2 nofb:WN_EPI_FUSED_BWD=0" --reps 3 --tag order > $O/ab.txt 2>&1
cp gpurun_out/ab_vars_order.json $O/
cat $O/ab.txt | tail -5
