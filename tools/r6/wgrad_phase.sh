#!/bin/bash
# wgrad_big_k with the two waves of a SIMD splitting the next k-step's rows in front of different row tiles (tools/_var_wgAB.so: waves 0-3 at A, 4-7 at B)
mkdir -p gpurun_out/r6
for v in "" wg44 wg26 wgp26 wgq26 wgp04; do
  if [ -z "$v" ]; then python tools/wgrad_bench.py; else WAVENET_HIP_LIB=tools/_var_$v.so python tools/wgrad_bench.py; fi
done 2>&1 | grep "^{" | tee gpurun_out/r6/wgrad_phase.jsonl
WAVENET_HIP_LIB=tools/_var_wg26.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wgrad" 2>&1 | tail -3
