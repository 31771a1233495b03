#!/bin/bash
# Developer aid (GPU box): build the library with extra compile flags and run a timing + a parity subset against it.
#   bash tools/variant.sh NAME "-DFLAG ..." ["pytest -k expression"]
NAME=$1; FLAGS=$2; KEXPR=${3:-"grads_64 or sweep or fused_train"}
D=/tmp/pqb/$NAME; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j32 EXTRA="$FLAGS" > $D/make.log 2>&1) || { echo "build failed"; tail -5 $D/make.log; exit 1; }
echo "== $NAME ($FLAGS)"
WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so python3 tools/kbench.py bwd --reps 5 2>/dev/null | tail -1
WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -m gpu -x -q -k "$KEXPR" 2>&1 | tail -4
