#!/bin/bash
# GPU box: every randomised parity fuzzer on fresh seeds; log under gpurun_out/ (copied into profiles/rNN_fuzz.md by hand).
#   bash tools/fuzz_all.sh SEED0 [SCALE]      SCALE multiplies the case counts (default 1)
S=${1:-100}; K=${2:-1}; L=gpurun_out/fuzz_all_$S.log; mkdir -p gpurun_out; : > $L
run() { echo "== $*" >> $L; timeout 3000 "$@" 2>&1 | grep -v "^ok   case.*e-[0-9][0-9] *$\|^ok   case.*rows equal$\|amdgpu.ids\|torch intra-op" | tail -12 >> $L; }
run python tools/fuzz_parity.py --cases $((120 * K)) --seed $S
WN_PQ_CHAIN=0 run python tools/fuzz_parity.py --cases $((60 * K)) --seed $((S + 1))
run python tools/fuzz_parity.py --epi --cases $((120 * K)) --seed $((S + 7))          # 256 skip channels: the fused epilogue launches on random shapes
WN_EPI_FUSED=0 WN_EPI_FUSED_BWD=0 run python tools/fuzz_parity.py --epi --cases $((40 * K)) --seed $((S + 8))      # ... and the three-launch forms (B-stationary dZ product)
run python tools/fuzz_ae.py --cases $((80 * K)) --seed $((S + 2))
run python tools/fuzz_ae.py --general --cases $((40 * K)) --seed $((S + 3))
run python tools/fuzz_decode.py --shapes --cases $((40 * K)) --seed $((S + 4))
run python tools/fuzz_decode.py --cases $((40 * K)) --seed $((S + 5))
run python tools/fuzz_generic.py --cases $((40 * K)) --seed $((S + 6))
run python tools/fuzz_epilogue.py --cases $((150 * K)) --seed $((S + 9))       # the fused epilogue launches at kernel level (NaN outside the window, canaries)
echo "ties judged by the float64 oracle: $(grep -c 'float64 oracle' $L)" >> $L
cat $L
