#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== fuzz_parity seed 51 (120 cases)"; timeout 900 python tools/fuzz_parity.py --cases 120 --seed 51 2>&1 | tail -n 4
echo "== fuzz_parity seed 52, WN_PQ_CHAIN=0 (60 cases)"; WN_PQ_CHAIN=0 timeout 600 python tools/fuzz_parity.py --cases 60 --seed 52 2>&1 | tail -n 4
echo "== fuzz_ae seed 53 (40)"; timeout 900 python tools/fuzz_ae.py --cases 40 --seed 53 2>&1 | grep -v "^ok" | tail -n 8
echo "== fuzz_decode --shapes seed 54 (40)"; timeout 900 python tools/fuzz_decode.py --shapes --cases 40 --seed 54 2>&1 | grep -v "^ok" | tail -n 6
echo "== fuzz_generic seed 55 (30)"; timeout 900 python tools/fuzz_generic.py --cases 30 --seed 55 2>&1 | grep -v "^ok" | tail -n 6
} > gpurun_out/r4_fuzz3.log 2>&1
grep -c "^FAIL" gpurun_out/r4_fuzz3.log; grep "cases failed\|==" gpurun_out/r4_fuzz3.log
