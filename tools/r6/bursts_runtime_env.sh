#!/bin/bash
# burst frequency under HIP / HSA runtime switches (300 timed steps each, same box)
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
env | grep -E "^(HIP|HSA|GPU|AMD|ROC)" | head -20
for r in 1 2; do
for v in "base:" "noint:HSA_ENABLE_INTERRUPT=0" "q2:GPU_MAX_HW_QUEUES=2" "q8:GPU_MAX_HW_QUEUES=8" "nodd:AMD_DIRECT_DISPATCH=0" "nosdma:HSA_ENABLE_SDMA=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e python bench.py --no-cpu-baseline --no-extras --steps 300 --dump-steps > $O/$n.$r.json 2> $O/$n.$r.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.$r.json"))
    a=d["ms_per_step_stats"]["all_in_order"]
    med=d["ms_per_step_stats"]["median"]
    slow=[x for x in a if x>1.15*med]
    print("%-6s r$r mean %.3f median %.3f  slow steps %d of %d (max %.2f)  excess %.1f ms  host enqueue %.2f" % ("$n", d["ms_per_step"], med, len(slow), len(a), max(a), sum(x-med for x in slow), d["host_enqueue_ms_per_step"]))
except Exception as e:
    print("$n r$r failed", e)
PY
done; done
