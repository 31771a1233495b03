#!/bin/bash
# GPU box: phase clocks of the backward block (developer build -DPQ_DBG), pair and chain forms; extra flags in $1
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_clocks.log; : > $L
D=/tmp/pqb/DBG; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j16 EXTRA="-DPQ_DBG $1" > $D/make.log 2>&1) || { echo "build failed" >> $L; tail -5 $D/make.log >> $L; }
for c in 0 1; do
  echo "== PQ_DBG $1 WN_PQ_CHAIN=$c" >> $L
  WN_PQ_CHAIN=$c WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/pq_clocks.py 2>/dev/null | tail -3 >> $L
done
cat $L
