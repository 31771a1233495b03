#!/bin/bash
# the epilogue's weight-gradient launches alone: shipped, and timing builds without their global loads / split + LDS fill / MFMAs (wrong results on purpose)
mkdir -p gpurun_out/r6
for v in "" wg1 wg2 wg3; do
  if [ -z "$v" ]; then python tools/wgrad_bench.py; else WAVENET_HIP_LIB=tools/_var_$v.so python tools/wgrad_bench.py; fi
done 2>&1 | grep "^{" | tee gpurun_out/r6/wgrad_price.jsonl
