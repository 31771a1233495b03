#!/bin/bash
mkdir -p gpurun_out/r6c7
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c7
timeout 1500 python tools/ab_vars.py --vars "unfused:WN_EPI_FUSED=0 epi1 epi2 epi3 epi4 epi6 epi7 epi8" --reps 2 --tag epit2 > $O/ab.txt 2>&1
cp gpurun_out/ab_vars_epit2.json $O/
cat $O/ab.txt | tail -10
