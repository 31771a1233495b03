#!/bin/bash
mkdir -p gpurun_out/r6c19
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c19
for r in 1 2 3; do
for v in "base:" "nodd:AMD_DIRECT_DISPATCH=0" "old:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0 WN_GEMM_BST=0" "oldnodd:AMD_DIRECT_DISPATCH=0 WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0 WN_GEMM_BST=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e python bench.py --no-cpu-baseline --no-extras --steps 300 --dump-steps > $O/$n.$r.json 2> $O/$n.$r.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.$r.json"))
    a=d["ms_per_step_stats"]["all_in_order"]
    med=d["ms_per_step_stats"]["median"]
    slow=[i for i,x in enumerate(a) if x>1.15*med]
    p=d["phase_ms_per_step"]
    print("%-8s r$r mean %.3f median %.3f  slow %d of %d at %s  host enq %.2f | fwd %.3f efwd %.3f ebwd %.3f sbwd %.3f" % ("$n", d["ms_per_step"], med, len(slow), len(a), slow[:12], d["host_enqueue_ms_per_step"], p["stack_fwd"], p["epilogue_fwd"], p["epilogue_bwd"], p["stack_bwd"]))
except Exception as e:
    print("$n r$r failed", e)
PY
done; done
