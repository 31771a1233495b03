#!/bin/bash
# GPU box: same-box A/B of the whole step's phases, shipped library against a build with extra flags ($1), $2 alternations (chain form on)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_ab2.log; : > $L
D=/tmp/pqb/AB; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j16 EXTRA="$1" > $D/make.log 2>&1) || { echo "build failed" >> $L; tail -5 $D/make.log >> $L; }
for rep in $(seq 1 ${2:-4}); do
  echo "== shipped" >> $L
  timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | cut -c14- >> $L
  echo "== $1" >> $L
  WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | cut -c14- >> $L
done
cat $L
