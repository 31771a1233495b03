#!/bin/bash
# GPU box: parity tests of the backward block on the shipped library, then same-box A/B of the stack against a prebuilt library ($1), $2 alternations
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_ab3.log; : > $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_kernels.py tests/test_gpu_switches.py -k "pq or chain or block" 2>&1 | tail -3 >> $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py -k "c2" 2>&1 | tail -3 >> $L
for rep in $(seq 1 ${2:-4}); do
  for c in 1 0; do
    echo "== new WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | grep -o '"stack_bwd": [0-9.]*' >> $L
    echo "== base WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c WAVENET_HIP_LIB=$1 timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | grep -o '"stack_bwd": [0-9.]*' >> $L
  done
done
cat $L
