#!/bin/bash
# GPU box: XW (x operands of the recompute from the W waves): parity tests, then same-box A/B against a -DPQ_T_NOXW build
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_xw.log; : > $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_kernels.py tests/test_gpu_switches.py -k "pq or chain or block" 2>&1 | tail -5 >> $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py -k "c2" 2>&1 | tail -5 >> $L
D=/tmp/pqb/NOXW; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j16 EXTRA="-DPQ_T_NOXW" > $D/make.log 2>&1) || { echo "build failed" >> $L; tail -5 $D/make.log >> $L; }
for rep in 1 2 3 4; do
  for c in 1 0; do
    echo "== XW WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 >> $L
    echo "== NOXW WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 >> $L
  done
done
bash tools/r4_clocks.sh > /dev/null 2>&1
cat gpurun_out/r4_clocks.log >> $L
cat $L
