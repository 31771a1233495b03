#!/bin/bash
# bit-reproducibility of the fused step over many runs on the same inputs (config 2 and config 4)
mkdir -p gpurun_out/r6
timeout 1500 python tools/soak_determinism.py --reps 2000 2> gpurun_out/r6/soak.err | tee gpurun_out/r6/soak.json; echo "rc ${PIPESTATUS[0]}"; tail -3 gpurun_out/r6/soak.err
