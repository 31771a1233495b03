#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_switches.py -m gpu -q -p no:cacheprovider -k "skip_epilogue or fused_epilogue or b_stationary" > $O/kernels.log 2>&1; echo "kernels exit $?"; tail -2 $O/kernels.log
timeout 300 python tools/gemm_bench.py --rounds 3 2>/dev/null | grep fused
for r in 1 2 3; do
for v in "old:WN_EPI_FUSED_BWD=0 WN_EPI_FUSED=0 WN_GEMM_BST=0" "new:"; do
  n=${v%%:*}; e=${v#*:}
  env $e python bench.py --no-cpu-baseline --no-extras --steps 200 --dump-steps > $O/$n.$r.json 2> $O/$n.$r.err
  python - <<PY
import json
d=json.load(open("$O/$n.$r.json"))
a=d["ms_per_step_stats"]["all_in_order"]
slow=[x for x in a if x>1.15*d["ms_per_step_stats"]["median"]]
print("%-6s r$r mean %.3f median %.3f  slow steps %d of %d (max %.2f)  sum of excess %.1f ms  roof %.3f / %.3f" % ("$n", d["ms_per_step"], d["ms_per_step_stats"]["median"], len(slow), len(a), max(a), sum(x-d["ms_per_step_stats"]["median"] for x in slow), d["roofline"]["frac"], d["roofline"]["median_step"]["frac"]))
PY
done; done
