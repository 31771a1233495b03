#!/bin/bash
# Run on the GPU box (via gpurun): kernel unit tests, parity tests, smoke, bench (+ optional profile).
# Everything is logged under gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
STAGE=${1:-all}
if [ "$STAGE" = "all" ] || [ "$STAGE" = "tests" ]; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -p no:cacheprovider > gpurun_out/kernels.log 2>&1
  echo "kernels exit $?" | tee -a gpurun_out/summary.txt
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider > gpurun_out/parity.log 2>&1
  echo "parity exit $?" | tee -a gpurun_out/summary.txt
  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
  echo "smoke exit $?" | tee -a gpurun_out/summary.txt
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "bench" ]; then
  timeout 900 python bench.py --steps 10 --warmup 3 --phases > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench exit $?" | tee -a gpurun_out/summary.txt
fi
tail -n 60 gpurun_out/kernels.log gpurun_out/parity.log gpurun_out/smoke.log gpurun_out/bench.log gpurun_out/bench.err 2>/dev/null | tail -n 150
