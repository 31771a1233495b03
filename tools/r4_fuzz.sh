#!/bin/bash
# round 4: long randomised parity runs on the GPU (chain form of the backward block, split decode forms, general plans)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== fuzz_parity seed 41 (120 cases)"; timeout 900 python tools/fuzz_parity.py --cases 120 --seed 41 2>&1 | tail -n 40
echo "== fuzz_parity seed 42, WN_PQ_CHAIN=0 (40 cases)"; WN_PQ_CHAIN=0 timeout 600 python tools/fuzz_parity.py --cases 40 --seed 42 2>&1 | tail -n 8
echo "== fuzz_decode --shapes seed 43 (60 cases)"; timeout 900 python tools/fuzz_decode.py --shapes --cases 60 --seed 43 2>&1 | tail -n 70
echo "== fuzz_decode --shapes seed 44 WN_DEC_KS=8 (30)"; WN_DEC_KS=8 timeout 600 python tools/fuzz_decode.py --shapes --cases 30 --seed 44 2>&1 | tail -n 35
echo "== fuzz_decode --shapes seed 45 WN_DEC_KS=1 (30)"; WN_DEC_KS=1 timeout 600 python tools/fuzz_decode.py --shapes --cases 30 --seed 45 2>&1 | tail -n 35
echo "== fuzz_ae seed 46 (40)"; timeout 900 python tools/fuzz_ae.py --cases 40 --seed 46 2>&1 | tail -n 45
echo "== fuzz_generic seed 47 (40)"; timeout 900 python tools/fuzz_generic.py --cases 40 --seed 47 2>&1 | tail -n 45
} > gpurun_out/r4_fuzz.log 2>&1
grep -c FAIL gpurun_out/r4_fuzz.log; grep "cases failed" gpurun_out/r4_fuzz.log
