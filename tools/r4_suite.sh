#!/bin/bash
# GPU box: the whole GPU suite (every failure listed), then phase clocks / spans of the backward block (developer builds).
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_suite.log; : > $L
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4_suite_tests.log 2>&1; echo "gpu suite exit $?" >> $L
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r4_suite_tests.log | tail -30 >> $L
for v in DBG SPAN; do
  D=/tmp/pqb/$v; rm -rf $D; mkdir -p $D/music_amd $D/include
  cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
  (cd $D/music_amd/csrc && make -j16 EXTRA="-DPQ_$v" > $D/make.log 2>&1) || { echo "build $v failed" >> $L; tail -5 $D/make.log >> $L; continue; }
  for c in 0 1; do
    echo "== PQ_$v WN_PQ_CHAIN=$c" >> $L
    if [ $v = DBG ]; then WN_PQ_CHAIN=$c WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/pq_clocks.py 2>/dev/null | tail -2 >> $L
    else WN_PQ_CHAIN=$c WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/pq_spans.py 2>/dev/null | tail -60 >> $L; fi
  done
done
cat $L
