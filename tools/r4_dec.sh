#!/bin/bash
# GPU box: decode forms (split skip parts, tap-0 table in global memory): parity tests, then speed A/B; then the whole GPU suite.
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_dec.log; : > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider -k "decode or generat" > gpurun_out/r4_dec_tests.log 2>&1; echo "decode tests exit $?" >> $L
tail -4 gpurun_out/r4_dec_tests.log >> $L
for ks in 1 8; do for t0 in 0 1; do
  echo "== shipped 40 x 32/32/512: WN_DEC_KS=$ks WN_DEC_T0=$t0" >> $L
  WN_DEC_VERBOSE=1 WN_DEC_KS=$ks WN_DEC_T0=$t0 timeout 300 python tools/dec_speed_shipped.py 2>&1 | grep -E "samples|wn_decode" | sort | uniq -c | sort -rn | head -6 >> $L
done; done
for ks in 1 4; do
  echo "== config 5 (30 x 64/64/256): WN_DEC_KS=$ks" >> $L
  WN_DEC_KS=$ks timeout 300 python tools/dec_speed.py 2>&1 | tail -4 >> $L
done
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4_dec_suite.log 2>&1; echo "gpu suite exit $?" >> $L
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r4_dec_suite.log | tail -20 >> $L
cat $L
