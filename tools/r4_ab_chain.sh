#!/bin/bash
# GPU box: same-box A/B of the chain form (final code), five alternations; then bench + rocprofv3 stats + PMC passes (tools/gpu_check.sh prof).
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_ab_chain.log; : > $L
for rep in 1 2 3 4 5; do
  for c in 0 1; do
    echo "== WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 >> $L
  done
done
cat $L
