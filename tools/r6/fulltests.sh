#!/bin/bash
mkdir -p gpurun_out/r6tests
export PYTHONUNBUFFERED=1
O=gpurun_out/r6tests
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/all.log 2>&1; echo "all gpu tests exit $?"
tail -n 25 $O/all.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $O/smoke.log
