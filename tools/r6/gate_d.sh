#!/bin/bash
# the backward kernels' gate derivatives from wn_gate_d (no cancellation, no overflow): tests, alternation against the library before it (tools/_var_old.so), fuzzers, soak
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k saturated 2>&1 | grep -E "passed|failed|saturated gates|  err " | tail -12
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6/gate_d_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r6/gate_d_tests.log
cp music_amd/libwavenet_hip.so tools/_var_gated.so
WAVENET_HIP_LIB_SAVE=1 timeout 1200 python tools/ab_vars.py --vars "old" --reps 4 --tag gate_d > gpurun_out/r6/gate_d_ab.log 2>&1; tail -4 gpurun_out/r6/gate_d_ab.log
timeout 1200 python tools/ab_vars.py --bench ae --vars "old" --reps 3 --tag gate_d_ae > gpurun_out/r6/gate_d_ab_ae.log 2>&1; tail -4 gpurun_out/r6/gate_d_ab_ae.log
bash tools/r6/soak.sh
