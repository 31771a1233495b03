#!/bin/bash
# Build container (no GPU needed): a VARIANT of the library with extra flags on some source files, linked against the shipped
# objects of the rest:  bash tools/mkvar.sh NAME "-DFLAG ..." file1.hip [file2.hip ...]  ->  tools/_var_NAME.so
# (git-ignored, travels to the GPU box with the snapshot; WAVENET_HIP_LIB=tools/_var_NAME.so selects it).
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../music_amd/csrc"
make -s -j8 > /dev/null
D=/tmp/mkvar/$NAME; rm -rf $D; mkdir -p $D
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$b.hip" = "$f" ]; then
      /opt/rocm/bin/hipcc $FLAGS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -c $f -o $D/$b.o
      use=$D/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o ../../tools/_var_$NAME.so
echo "built tools/_var_$NAME.so ($FLAGS: $*)"
