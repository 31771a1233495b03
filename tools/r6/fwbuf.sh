#!/bin/bash
# forward block with an interior copy of its body (buffer addressing, tile-by-tile gate / store) - tools/_var_fwbuf.so - against the shipped library
mkdir -p gpurun_out/r6
WAVENET_HIP_LIB=tools/_var_fwbuf.so timeout 1500 python -m pytest tests -m gpu -q -k "parity or fullsize or kernels or switches" > gpurun_out/r6/fwbuf_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r6/fwbuf_tests.log
timeout 1200 python tools/ab_vars.py --vars "fwbuf" --reps 4 --tag fwbuf > gpurun_out/r6/fwbuf_ab.log 2>&1; tail -4 gpurun_out/r6/fwbuf_ab.log
timeout 1200 python tools/ab_vars.py --bench ae --vars "fwbuf" --reps 3 --tag fwbuf_ae > gpurun_out/r6/fwbuf_ab_ae.log 2>&1; tail -4 gpurun_out/r6/fwbuf_ab_ae.log
