#!/bin/bash
# Developer aid (GPU box): build the library with extra compile flags and time the decoder at the reference's shipped widths.
#   bash tools/dec_variant.sh NAME "-DFLAG ..."
NAME=$1; FLAGS=$2
D=/tmp/pqb/$NAME; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j32 EXTRA="$FLAGS" > $D/make.log 2>&1) || { echo "build failed"; tail -5 $D/make.log; exit 1; }
echo "== $NAME ($FLAGS)"
WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so python3 tools/dec_speed_shipped.py 2>/dev/null | tail -2
