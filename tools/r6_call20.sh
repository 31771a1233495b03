#!/bin/bash
mkdir -p gpurun_out/r6c20
export PYTHONUNBUFFERED=1
O=gpurun_out/r6c20
for r in 1 2 3; do
for v in "a0:WN_BENCH_MAX_AHEAD=0" "a2:WN_BENCH_MAX_AHEAD=2" "a3:WN_BENCH_MAX_AHEAD=3" "a6:WN_BENCH_MAX_AHEAD=6" "a12:WN_BENCH_MAX_AHEAD=12"; do
  n=${v%%:*}; e=${v#*:}
  env $e python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --dump-steps > $O/$n.$r.json 2> $O/$n.$r.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.$r.json"))
    a=d["ms_per_step_stats"]["all_in_order"]
    med=d["ms_per_step_stats"]["median"]
    slow=[i for i,x in enumerate(a) if x>1.15*med]
    print("%-4s r$r mean %.3f median %.3f  slow %d of %d at %s  host enq %.2f roof %.3f" % ("$n", d["ms_per_step"], med, len(slow), len(a), slow[:12], d["host_enqueue_ms_per_step"], d["roofline"]["frac"]))
except Exception as e:
    print("$n r$r failed", e)
PY
done; done
