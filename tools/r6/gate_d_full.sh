#!/bin/bash
# after wn_gate_d: every GPU test, then every fuzzer on fresh seeds
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6/gate_d_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r6/gate_d_tests.log
bash tools/fuzz_all.sh 600 1 > gpurun_out/r6/gate_d_fuzz.log 2>&1; grep -E "^==|cases failed|FAIL|ties judged" gpurun_out/r6/gate_d_fuzz.log | tail -30
