#!/bin/bash
# GPU box: parity tests of the backward block, then the whole bench step with the chain form on / off, alternating
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_form.log; : > $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_kernels.py tests/test_gpu_switches.py -k "pq or chain or block" 2>&1 | tail -2 >> $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py -k "c2" 2>&1 | tail -2 >> $L
for rep in 1 2 3; do
  for c in 1 0; do
    echo "chain=$c: $(WN_PQ_CHAIN=$c timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_stats'], d['phase_ms_per_step']['stack_bwd'])")" >> $L
  done
done
cat $L
