#!/bin/bash
# GPU box: autoencoder parity tests on the shipped library, then same-box timing of the config-4 step for several flag builds ("flagsA|flagsB" in $1)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_abae.log; : > $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_parity.py -k "autoencoder or ae_" 2>&1 | tail -2 >> $L
timeout 1500 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_gpu_fullsize.py -k "c4" 2>&1 | tail -2 >> $L
IFS='|' read -ra FL <<< "$1"
i=0
for f in "${FL[@]}"; do
  D=/tmp/pqb/V$i; rm -rf $D; mkdir -p $D/music_amd $D/include
  cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
  (cd $D/music_amd/csrc && make -j16 EXTRA="$f" > $D/make.log 2>&1) || { echo "build failed: $f" >> $L; tail -5 $D/make.log >> $L; }
  i=$((i+1))
done
for rep in $(seq 1 ${2:-3}); do
  echo "shipped: $(timeout 300 python tools/ae_phases.py 2>/dev/null | tail -1 | cut -c1-400)" >> $L
  i=0
  for f in "${FL[@]}"; do
    echo "[$f]: $(WAVENET_HIP_LIB=/tmp/pqb/V$i/music_amd/libwavenet_hip.so timeout 300 python tools/ae_phases.py 2>/dev/null | tail -1 | cut -c1-400)" >> $L
    i=$((i+1))
  done
done
cat $L
