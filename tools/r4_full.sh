#!/bin/bash
# GPU box: the whole GPU suite, the drop-in surface step (flat / torch optimizer, with a kernel trace), a short bench line.
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
REPO=$(pwd); L=gpurun_out/r4_full.log; : > $L
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r4_full_tests.log 2>&1; echo "gpu suite exit $?" >> $L
tail -5 gpurun_out/r4_full_tests.log >> $L
for o in torch flat; do timeout 300 python tools/surface_prof.py --opt $o --steps 20 2>/dev/null | tail -1 >> $L; done
rm -rf gpurun_out/kt_surf
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kt_surf -- python3 $REPO/tools/surface_prof.py --opt flat --steps 10 > $REPO/gpurun_out/kt_surf.log 2>&1)
python3 - >> $L <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt_surf/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("== surface step kernels (15 steps incl. warm-up): total %.2f ms" % (tot / 1e6))
    for r in rows[:28]:
        print("%-70s calls=%5s total=%8.3f ms avg=%8.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find gpurun_out/kt_surf -type f -size +2M -delete
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err; echo "bench exit $?" >> $L
python3 - >> $L <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4_bench_a.json").read().strip().split("\n")[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_stats", "first_window")})
    print("roofline", d["roofline"]["frac"], "stack", d["roofline_stack_fwd_bwd"]["frac"], d["phase_ms_per_step"])
    print("surface", d["extra"]["reference_surface_step"].get("loader_onehot"), "shipped decode", d["extra"]["shipped_params"]["wavenet"].get("decode_single_stream_samples_per_s"))
    print("c4", d["extra"]["c4_autoencoder"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"])
except Exception as e:
    print("bench parse failed", e)
PY
cat $L
