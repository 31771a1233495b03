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
    print("%-10s mean %.3f median %.3f  slow %d of %d at %s  host enq %.2f roof %.3f" % ("$n", d["ms_per_step"], med, len(slow), len(a), slow[:14], d["host_enqueue_ms_per_step"], d["roofline"]["frac"]))
except Exception as e:
    print("$n failed", e); print(open("$O/$n.err").read()[-600:])
PY
}
for r in 1 2; do
run t0_50.$r WN_MAX_STEPS_IN_FLIGHT=0
run t4_50.$r WN_MAX_STEPS_IN_FLIGHT=4
run t2_50.$r WN_MAX_STEPS_IN_FLIGHT=2
run t4_300.$r WN_MAX_STEPS_IN_FLIGHT=4 --steps 300
run t4_drv.$r WN_MAX_STEPS_IN_FLIGHT=4 --steps 20 --warmup 5
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
