# SQ counters for the backward stack kernels (developer aid): one rocprofv3 --pmc pass per counter group.
mkdir -p gpurun_out; REPO=$(pwd); export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$tag -- python3 $REPO/tools/kbench.py bwd --reps 2 > $REPO/gpurun_out/pmc_$tag.log 2>&1)
  python3 - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
fs = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
if not fs:
    print(tag, "no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if k.startswith(("resblock_bwd_pq_k<true, false", "resblock_fwd_nt_k<F16, 3, 64", "chan_gemm_wide2_k", "wgrad_big_k", "reduce_slabs")):
        print(k, {c: round(sum(v) / len(v)) for c, v in acc[k].items()})
PY
  find gpurun_out/pmc_$tag -type f -size +3M -delete 2>/dev/null
done
