mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6/v_tests.log 2>&1; echo "tests rc $?"; tail -1 gpurun_out/r6/v_tests.log
bash tools/gpu_check.sh prof > gpurun_out/r6/v_prof.log 2>&1; tail -1 gpurun_out/r6/v_prof.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/final_bench.json 2> gpurun_out/r6/final_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/final_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"].get("ms_per_step_median"))
PY
