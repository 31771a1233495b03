D=/tmp/pqb/DBG; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j32 EXTRA="-DPQ_DBG" > $D/make.log 2>&1) || tail -5 $D/make.log
WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so python3 tools/pq_clocks.py $1 2>&1 | tail -5
