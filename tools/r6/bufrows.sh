#!/bin/bash
# config-4 decoder backward block with its row requests through buffer descriptors (tools/_var_bufrows.so) against the shipped library
mkdir -p gpurun_out/r6
WAVENET_HIP_LIB=tools/_var_bufrows.so timeout 900 python -m pytest tests -m gpu -x -q -k "autoencoder or ae or c4 or model1" > gpurun_out/r6/bufrows_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r6/bufrows_tests.log
timeout 1200 python tools/ab_vars.py --bench ae --vars "bufrows" --reps 3 --tag bufrows > gpurun_out/r6/bufrows_ab.log 2>&1; tail -30 gpurun_out/r6/bufrows_ab.log
