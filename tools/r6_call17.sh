#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd)
rm -rf gpurun_out/btrace; mkdir -p gpurun_out/btrace
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/btrace -- python3 $REPO/bench.py --steps 300 --warmup 5 --settle 30 --no-cpu-baseline --no-extras > $REPO/gpurun_out/btrace/bench.json 2> $REPO/gpurun_out/btrace/err.log)
f=$(find gpurun_out/btrace -name "*kernel_trace.csv" | head -1)
python tools/burst_trace.py $f | grep -v "resblock_bwd_pq_k.*median\|wgrad_big_k.*median" | head -70
echo "== round-5 configuration"
rm -rf gpurun_out/btrace2; mkdir -p gpurun_out/btrace2
(cd /tmp && WN_EPI_FUSED=0 WN_EPI_FUSED_BWD=0 WN_GEMM_BST=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/btrace2 -- python3 $REPO/bench.py --steps 300 --warmup 5 --settle 30 --no-cpu-baseline --no-extras > $REPO/gpurun_out/btrace2/bench.json 2> $REPO/gpurun_out/btrace2/err.log)
f=$(find gpurun_out/btrace2 -name "*kernel_trace.csv" | head -1)
python tools/burst_trace.py $f | grep -v "resblock_bwd_pq_k.*median\|wgrad_big_k.*median" | head -50
find gpurun_out/btrace gpurun_out/btrace2 -type f -size +3M -delete
