#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/r4_clocks.sh > /dev/null 2>&1
{ echo "== fuzz_ae --general seed 49 (40)"; timeout 1200 python tools/fuzz_ae.py --general --cases 40 --seed 49 2>&1 | grep -v "^ok" | tail -n 30; } > gpurun_out/r4_fuzz3.log 2>&1
cat gpurun_out/r4_clocks.log; cat gpurun_out/r4_fuzz3.log
