#!/bin/bash
# which of the new kernels brings the bursts of slow steps?  200 timed steps per configuration, steps over 5.2 ms counted
mkdir -p gpurun_out/bursts
export PYTHONUNBUFFERED=1
for r in 1 2; do
for v in "old:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0 WN_GEMM_BST=0" "bst1:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0 WN_GEMM_BST=1" "bst2:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0 WN_GEMM_BST=2" "fwd:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=1 WN_GEMM_BST=0" "fwdns:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=1 WN_GEMM_BST=0 WN_EPI_STAGGER=0" "bwd:WN_EPI_FUSED_BWD=1 WN_EPI_FUSED=0 WN_EPI_BWD_ORDER=1" "all1:WN_EPI_BWD_ORDER=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python bench.py --no-cpu-baseline --no-extras --steps 200 --dump-steps > gpurun_out/bursts/$n.$r.json 2> gpurun_out/bursts/$n.$r.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bursts/$n.$r.json"))
a=d["ms_per_step_stats"]["all_in_order"]
slow=[x for x in a if x>5.2]
print("%-6s r$r mean %.3f median %.3f  slow steps %d of %d (max %.2f)  sum of excess %.1f ms" % ("$n", d["ms_per_step"], d["ms_per_step_stats"]["median"], len(slow), len(a), max(a), sum(x-d["ms_per_step_stats"]["median"] for x in slow)))
PY
done; done
