#!/bin/bash
mkdir -p gpurun_out/r6c2
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c2
for v in shipped bst1 bst2 bst3 bst4 bst5; do
  if [ $v = shipped ]; then L=music_amd/libwavenet_hip.so; else L=tools/_var_$v.so; fi
  WAVENET_HIP_LIB=$L timeout 300 python tools/gemm_bench.py --rounds 3 > $O/gemm_$v.json 2> $O/gemm_$v.err
  echo "$v: $(grep skipT $O/gemm_$v.json)"
done
