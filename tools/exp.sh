mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
python tools/kbench.py fwd --reps 10 > gpurun_out/kb_fwd.log 2>&1
cat gpurun_out/kb_fwd.log | tail -3
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_sq -- python3 $REPO/tools/kbench.py fwd --reps 2 > $REPO/gpurun_out/pmc_sq.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_sq2 -- python3 $REPO/tools/kbench.py fwd --reps 2 > $REPO/gpurun_out/pmc_sq2.log 2>&1)
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_sq", "gpurun_out/pmc_sq2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print("no csv in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "resblock" in k or "gemm" in k or "wgrad" in k:
            print(k, {a: round(b / n[(k, a)]) for a, b in v.items()})
PY
tail -3 gpurun_out/pmc_sq.log
