#!/bin/bash
mkdir -p gpurun_out/r6tol
export PYTHONUNBUFFERED=1
O=gpurun_out/r6tol
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_generic.py tests/test_gpu_surface.py tests/test_gpu_sweep.py -m gpu -q -s -p no:cacheprovider > $O/tol.log 2>&1; echo "exit $?"
grep -E "worst|FAILED|Error|passed|failed" $O/tol.log | awk '{ if (length($0) > 220) print substr($0,1,220); else print }' | tail -60
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
for t in fuzz_parity fuzz_ae fuzz_generic; do timeout 900 python tools/$t.py --cases 40 > $O/$t.log 2>&1; echo "$t exit $?"; tail -2 $O/$t.log; done
