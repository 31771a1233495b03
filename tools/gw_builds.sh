# timing builds of the wide GEMM's k-step (GPU box): skip product alone, K = 1920 ... 64
for t in 0 1 2 3 4 5 6 0; do
  D=/tmp/gwb/$t; rm -rf $D; mkdir -p $D/music_amd $D/include
  cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
  (cd $D/music_amd/csrc && make -j32 EXTRA="-DGW_T=$t" > $D/make.log 2>&1) || { echo "build failed"; tail -5 $D/make.log; continue; }
  echo "== GW_T=$t"
  WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so python3 tools/kbench.py skip --reps 10 2>/dev/null | tail -1
done
