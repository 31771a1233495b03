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
if [ "$STAGE" = "all" ] || [ "$STAGE" = "prof" ]; then
  REPO=$(pwd)
  export TMPDIR=/tmp
  rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/prof.log 2>&1)
  echo "prof exit $?" | tee -a gpurun_out/summary.txt
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --settle 0 --no-cpu-baseline --no-extras > $REPO/gpurun_out/pmc_fetch.log 2>&1)
  echo "pmc fetch exit $?" | tee -a gpurun_out/summary.txt
  (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --settle 0 --no-cpu-baseline --no-extras > $REPO/gpurun_out/pmc_write.log 2>&1)
  echo "pmc write exit $?" | tee -a gpurun_out/summary.txt
  # config 4's stack kernels (conditioned decoder blocks, encoder blocks): the same two passes over its step
  rm -rf gpurun_out/pmc_fetch_ae gpurun_out/pmc_write_ae
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_fetch_ae -- python3 $REPO/tools/ae_phases.py > $REPO/gpurun_out/pmc_fetch_ae.log 2>&1)
  echo "pmc fetch (config 4) exit $?" | tee -a gpurun_out/summary.txt
  (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_write_ae -- python3 $REPO/tools/ae_phases.py > $REPO/gpurun_out/pmc_write_ae.log 2>&1)
  echo "pmc write (config 4) exit $?" | tee -a gpurun_out/summary.txt
  python3 tools/prof_summary.py gpurun_out > gpurun_out/prof_summary.md 2>&1
  # keep only the small files (the raw traces can be large)
  find gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch_ae gpurun_out/pmc_write_ae -type f -size +3M -delete 2>/dev/null
  cat gpurun_out/prof_summary.md | head -60
fi
