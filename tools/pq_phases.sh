#!/bin/bash
# Developer aid (GPU box): what a launch of the one-launch backward block is made of.  Builds the library once per timing
# switch (results are wrong with any of them) and prints the backward-stack time of each build.
#   PQ_VARIANTS="NOREC NOPQ" bash tools/pq_phases.sh
REPO=$(pwd)
for v in BASE ${PQ_VARIANTS:-NOREC NOGATE NOWG NOPQ NOSTORE NOFILLDY NOCONV}; do
  D=/tmp/pqb/$v
  rm -rf $D; mkdir -p $D/music_amd $D/include
  cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
  EXTRA=""; [ "$v" != "BASE" ] && EXTRA="-DPQ_T_$v"
  (cd $D/music_amd/csrc && make -j32 EXTRA="$EXTRA" > $D/make.log 2>&1) || { echo "== $v build failed"; tail -5 $D/make.log; continue; }
  echo "== $v"; WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so python3 tools/kbench.py bwd --reps 5 2>/dev/null | tail -1
done
