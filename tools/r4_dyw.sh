#!/bin/bash
# GPU box: dy-fragment fill on the W waves (chain form) A/B, then the whole GPU suite and the CPU-side suite on the box.
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_dyw.log; : > $L
D=/tmp/pqb/dyw0; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j16 EXTRA="-DPQ_DYW=0" > $D/make.log 2>&1) || { echo "build failed" >> $L; tail -5 $D/make.log >> $L; }
for rep in 1 2 3; do
  echo "== DYW (default)" >> $L; timeout 300 python tools/kbench.py bwd --reps 20 2>/dev/null | tail -1 >> $L
  echo "== DYW=0 (R waves fill)" >> $L; WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/kbench.py bwd --reps 20 2>/dev/null | tail -1 >> $L
done
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4_dyw_suite.log 2>&1; echo "gpu suite exit $?" >> $L
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r4_dyw_suite.log | tail -20 >> $L
cat $L
