#!/bin/bash
mkdir -p gpurun_out/r6c15
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c15
timeout 1500 python tools/ab_vars.py --vars "plain:WN_EPI_BWD_NT=0" --reps 4 --tag nt > $O/ab.txt 2>&1; tail -3 $O/ab.txt
timeout 1500 python tools/ab_vars.py --bench ae --vars "plain:WN_EPI_BWD_NT=0 nofb:WN_EPI_FUSED_BWD=0" --reps 3 --tag aent > $O/ab2.txt 2>&1; tail -4 $O/ab2.txt
