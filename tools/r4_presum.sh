#!/bin/bash
# GPU box: slab pre-sum + chain form: parity, then same-box A/B of the backward (chain x presum).
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
L=gpurun_out/r4_presum.log; : > $L
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -x -q -s -p no:cacheprovider -k "chain_form" >> $L 2>&1; echo "chain test exit $?" >> $L
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sweep.py tests/test_gpu_parity.py -m gpu -x -q -s -p no:cacheprovider -k "not decode and not generat" > gpurun_out/r4_presum_tests.log 2>&1; echo "fullsize+sweep+parity exit $?" >> $L
grep -E "non-vacuity|c2 at the bench|passed|failed|Error|error" gpurun_out/r4_presum_tests.log | tail -60 >> $L
for rep in 1 2; do
  for cfgv in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfgv
    echo "== WN_PQ_CHAIN=$1 WN_PQ_PRESUM=$2" >> $L
    WN_PQ_CHAIN=$1 WN_PQ_PRESUM=$2 timeout 300 python tools/kbench.py bwd --reps 20 2>/dev/null | tail -1 >> $L
  done
done
tail -70 $L
