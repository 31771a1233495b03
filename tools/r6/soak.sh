#!/bin/bash
# bit-reproducibility of the fused step over many runs on the same inputs (config 2, config 4, and the other forms of the step on a smaller geometry)
mkdir -p gpurun_out/r6
timeout 2400 python tools/soak_determinism.py --reps 2000 --what c2,c4,alt 2> gpurun_out/r6/soak.err | tee gpurun_out/r6/soak.json; echo "rc ${PIPESTATUS[0]}"; tail -3 gpurun_out/r6/soak.err
