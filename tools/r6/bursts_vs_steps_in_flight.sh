#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
run() { n=$1; e=$2; shift 2
  env $e python bench.py --gpus 1 "$@" --no-cpu-baseline --no-extras --dump-steps > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    a=d["ms_per_step_stats"]["all_in_order"]
    med=d["ms_per_step_stats"]["median"]
    slow=[i for i,x in enumerate(a) if x>1.15*med]
    print("%-10s mean %.3f median %.3f  slow %d of %d at %s  host enq %.2f" % ("$n", d["ms_per_step"], med, len(slow), len(a), slow[:14], d["host_enqueue_ms_per_step"]))
except Exception as e:
    print("$n failed", e)
PY
}
for r in 1 2; do
run a0_300.$r WN_BENCH_MAX_AHEAD=0 --steps 300 --warmup 10
run a3_300.$r WN_BENCH_MAX_AHEAD=3 --steps 300 --warmup 10
run a8_300.$r WN_BENCH_MAX_AHEAD=8 --steps 300 --warmup 10
run a0_50.$r WN_BENCH_MAX_AHEAD=0
run a3_50.$r WN_BENCH_MAX_AHEAD=3
run a0_drv.$r WN_BENCH_MAX_AHEAD=0 --steps 20 --warmup 5
run a3_drv.$r WN_BENCH_MAX_AHEAD=3 --steps 20 --warmup 5
done
