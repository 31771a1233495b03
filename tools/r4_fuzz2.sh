#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== deep decode tests"; timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_switches.py -k "tap0_table" 2>&1 | tail -n 5
echo "== fuzz_decode --shapes seed 43 (60 cases)"; timeout 1200 python tools/fuzz_decode.py --shapes --verbose --cases 60 --seed 43 2>&1 | grep -v "^     case" | tail -n 70
echo "== fuzz_decode --shapes seed 48 (60 cases)"; timeout 1200 python tools/fuzz_decode.py --shapes --verbose --cases 60 --seed 48 2>&1 | grep -v "^     case" | tail -n 70
} > gpurun_out/r4_fuzz2.log 2>&1
grep -c FAIL gpurun_out/r4_fuzz2.log; grep "cases failed\|passed\|failed" gpurun_out/r4_fuzz2.log
{
echo "== fuzz_ae --general seed 49 (40)"; timeout 1200 python tools/fuzz_ae.py --general --cases 40 --seed 49 2>&1 | tail -n 45
} >> gpurun_out/r4_fuzz2.log 2>&1
grep -c FAIL gpurun_out/r4_fuzz2.log; grep "cases failed" gpurun_out/r4_fuzz2.log
