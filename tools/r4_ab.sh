#!/bin/bash
# GPU box: parity tests of the backward block, then same-box A/B of the stack against a build with extra flags ($1), $2 alternations
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_ab.log; : > $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_kernels.py tests/test_gpu_switches.py -k "pq or chain or block" 2>&1 | tail -3 >> $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py -k "c2" 2>&1 | tail -3 >> $L
D=/tmp/pqb/AB; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j16 EXTRA="$1" > $D/make.log 2>&1) || { echo "build failed" >> $L; tail -5 $D/make.log >> $L; }
for rep in $(seq 1 ${2:-4}); do
  for c in 1 0; do
    echo "== new WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | grep -o '"stack_bwd": [0-9.]*' >> $L
    echo "== $1 WN_PQ_CHAIN=$c" >> $L
    WN_PQ_CHAIN=$c WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | grep -o '"stack_bwd": [0-9.]*' >> $L
  done
done
cat $L
