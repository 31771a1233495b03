#!/bin/bash
# same-box alternation of the default bench under engine switches: prints ms_per_step (mean), median, stack_bwd, epilogue phases
mkdir -p gpurun_out/benchab
export PYTHONUNBUFFERED=1
for r in 1 2 3; do
for v in "d:" "o1:WN_EPI_BWD_ORDER=1" "nofb:WN_EPI_FUSED_BWD=0" "nof:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0" "old:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0 WN_GEMM_BST=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e python bench.py --no-cpu-baseline --no-extras > gpurun_out/benchab/$n.$r.json 2> gpurun_out/benchab/$n.$r.err
  python - <<PY
import json
d=json.load(open("gpurun_out/benchab/$n.$r.json"))
p=d["phase_ms_per_step"]
print("%-5s r$r mean %.3f median %.3f  stack_fwd %.3f epi_fwd %.3f epi_bwd %.3f stack_bwd %.3f slab %.3f roof %.3f" % ("$n", d["ms_per_step"], d["ms_per_step_stats"]["median"], p["stack_fwd"], p["epilogue_fwd"], p["epilogue_bwd"], p["stack_bwd"], p["slab_reduce"], d["roofline"]["frac"]))
PY
done; done
