#!/bin/bash
# GPU box: chain form of the backward block - parity first, then same-box A/B of the backward stack (WN_PQ_CHAIN=0 is round 3's form).
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
L=gpurun_out/r4_chain.log; : > $L
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -x -q -s -p no:cacheprovider -k "chain_form" >> $L 2>&1; echo "chain test exit $?" >> $L
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sweep.py -m gpu -x -q -p no:cacheprovider >> $L 2>&1; echo "fullsize+sweep exit $?" >> $L
for rep in 1 2; do
  for c in 0 1; do
    echo "== WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c timeout 300 python tools/kbench.py bwd --reps 20 2>/dev/null | tail -1 >> $L
  done
done
tail -40 $L
