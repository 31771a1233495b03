#!/bin/bash
# GPU box: general-plan autoencoder tests, decode tests with the split form as default, decode speeds (kbench), short bench.
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_aegen.log; : > $L
timeout 1200 python -m pytest tests/test_gpu_generic.py -m gpu -q -s -p no:cacheprovider -k "autoencoder" > gpurun_out/r4_aegen_tests.log 2>&1; echo "ae generic exit $?" >> $L
grep -E "^ae_|passed|failed|Error|error|^E  " gpurun_out/r4_aegen_tests.log | head -40 >> $L
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_switches.py tests/test_gpu_sweep.py -m gpu -q -p no:cacheprovider -k "decode or generat" > gpurun_out/r4_aegen_dec.log 2>&1; echo "decode tests exit $?" >> $L
tail -3 gpurun_out/r4_aegen_dec.log >> $L
timeout 600 python tools/kbench.py decode 2>/dev/null | tail -1 >> $L
WN_DEC_KS=1 timeout 600 python tools/kbench.py decode 2>/dev/null | tail -1 >> $L
cat $L
