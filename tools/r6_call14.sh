#!/bin/bash
mkdir -p gpurun_out/r6c14
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c14
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "autoencoder or ae or G8" > $O/parity.log 2>&1; echo "ae parity exit $?"; tail -2 $O/parity.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "c4 or autoencoder" > $O/full.log 2>&1; echo "ae fullsize exit $?"; tail -2 $O/full.log
timeout 1500 python tools/ab_vars.py --bench ae --vars "nofb:WN_EPI_FUSED_BWD=0" --reps 3 --tag ae > $O/ab.txt 2>&1; tail -3 $O/ab.txt
