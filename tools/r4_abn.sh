#!/bin/bash
# GPU box: same-box timing of the backward stack for several flag builds ("flagsA|flagsB|..." in $1), $2 alternations, chain form on and off
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=gpurun_out/r4_abn.log; : > $L
IFS='|' read -ra FL <<< "$1"
i=0
for f in "${FL[@]}"; do
  D=/tmp/pqb/V$i; rm -rf $D; mkdir -p $D/music_amd $D/include
  cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
  (cd $D/music_amd/csrc && make -j16 EXTRA="$f" > $D/make.log 2>&1) || { echo "build failed: $f" >> $L; tail -5 $D/make.log >> $L; }
  i=$((i+1))
done
for rep in $(seq 1 ${2:-3}); do
  for c in 1 0; do
    echo "chain=$c shipped: $(WN_PQ_CHAIN=$c timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | grep -o '"stack_bwd": [0-9.]*')" >> $L
    i=0
    for f in "${FL[@]}"; do
      echo "chain=$c [$f]: $(WN_PQ_CHAIN=$c WAVENET_HIP_LIB=/tmp/pqb/V$i/music_amd/libwavenet_hip.so timeout 300 python tools/kbench.py bwd --reps 30 2>/dev/null | tail -1 | grep -o '"stack_bwd": [0-9.]*')" >> $L
      i=$((i+1))
    done
  done
done
sort $L | uniq -c | sort -k2 | head -0
cat $L
