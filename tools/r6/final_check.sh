#!/bin/bash
# the round's closing check on a fresh box: every GPU test, smoke(), and the bench with the driver's arguments
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6/final_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r6/final_tests.log
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/r6/final_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r6/final_smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/final_bench.json 2> gpurun_out/r6/final_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/final_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"].get("ms_per_step_median"))
PY
