#!/bin/bash
# GPU box: smoke, the default bench line, rocprofv3 stats + PMC passes of the bench command, summaries for profiles/.
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python __graft_entry__.py smoke > gpurun_out/r4_smoke.log 2>&1; echo "smoke exit $?"
timeout 1200 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo "bench exit $?"
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_driver.json 2> gpurun_out/r4_bench_driver.err; echo "bench (driver args) exit $?"
bash tools/gpu_check.sh prof > gpurun_out/r4_prof.log 2>&1; echo "prof exit $?"
cp gpurun_out/prof_summary.md gpurun_out/r4_prof_summary.md 2>/dev/null
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r4_kernel_stats.csv
tail -3 gpurun_out/r4_smoke.log; head -c 1500 gpurun_out/r4_bench.json; echo; head -40 gpurun_out/r4_prof_summary.md
