#!/bin/bash
# Build container (no GPU needed): a VARIANT of the library - developer switches (timing builds that remove one ingredient and give
# WRONG results, phase clocks, span stamps, in-kernel clock stamps) live in tools/exp/dev_switches.patch, NOT in the shipping sources:
#   bash tools/mkvar.sh NAME "-DFLAG ..." file1.hip [file2.hip ...]   ->   tools/_var_NAME.so
# copies music_amd/csrc to a scratch directory, applies the patch there, compiles the named files with the flags and links them with
# the shipped objects of the rest.  Extra experimental sources (tools/exp/wn_gemm_dma.hip, wn_gemm_w1.hip) can be named too; their hook
# is WN_EXP_GEMM=dma|w1 at run time (see tools/exp/README.md).  The result is git-ignored and travels to the GPU box with the snapshot;
# WAVENET_HIP_LIB=tools/_var_NAME.so selects it.
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
(cd "$ROOT/music_amd/csrc" && make -s -j8 > /dev/null)
D=/tmp/mkvar/$NAME; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r "$ROOT/music_amd/csrc" $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp "$ROOT/include/wavenet_hip.h" $D/include/
(cd $D && patch -s -p1 < "$ROOT/tools/exp/dev_switches.patch")
CC="/opt/rocm/bin/hipcc $FLAGS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=""
for o in "$ROOT"/music_amd/csrc/build/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$b.hip" = "$f" ]; then
      (cd $D/music_amd/csrc && $CC -c $f -o $D/$b.o)
      use=$D/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
for f in "$@"; do                      # experimental sources that are not part of the library
  if [ -f "$ROOT/tools/exp/$f" ]; then
    cp "$ROOT/tools/exp/$f" $D/music_amd/csrc/
    (cd $D/music_amd/csrc && $CC -c $f -o $D/$(basename $f .hip).o)
    OBJS="$OBJS $D/$(basename $f .hip).o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o "$ROOT/tools/_var_$NAME.so"
echo "built tools/_var_$NAME.so ($FLAGS: $*)"
