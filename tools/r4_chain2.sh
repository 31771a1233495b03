#!/bin/bash
# GPU box: per-kernel times of the chain / pair forms (rocprofv3 kernel trace of tools/kbench.py bwd), HBM traffic (PMC), and the
# timing build without halo items.
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
REPO=$(pwd); L=gpurun_out/r4_chain2.log; : > $L
summ() {  # kernel-trace csv -> per-kernel averages
python3 - "$1" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f: print("no trace"); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) // 3:]                      # skip the warm-up third
    if sum(v) > 300: print("%-62s n=%4d avg=%7.1f us  min=%7.1f" % (k, len(v), sum(v2) / len(v2), min(v)))
PY
}
for c in 0 1; do
  rm -rf gpurun_out/kt$c
  (cd /tmp && WN_PQ_CHAIN=$c timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/kt$c -- python3 $REPO/tools/kbench.py bwd --reps 10 > $REPO/gpurun_out/kt$c.log 2>&1)
  echo "== kernel trace WN_PQ_CHAIN=$c" >> $L; summ gpurun_out/kt$c >> $L
  find gpurun_out/kt$c -type f -size +2M -delete
done
for cnt in FETCH_SIZE WRITE_SIZE; do
  for c in 0 1; do
    rm -rf gpurun_out/pmc_${cnt}_$c
    (cd /tmp && WN_PQ_CHAIN=$c timeout 600 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_${cnt}_$c -- python3 $REPO/tools/kbench.py bwd --reps 2 > $REPO/gpurun_out/pmc_${cnt}_$c.log 2>&1)
    echo "== $cnt WN_PQ_CHAIN=$c (KiB per launch; FETCH_SIZE is half the bytes of wide reads on gfx950)" >> $L
    python3 - "$cnt" "$c" >> $L <<'PY'
import csv, glob, collections, sys
fs = glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (sys.argv[1], sys.argv[2]), recursive=True)
if not fs: print("no counter file"); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("%-62s n=%4d avg=%10.1f" % (k, len(v), sum(v) / len(v)))
PY
    find gpurun_out/pmc_${cnt}_$c -type f -size +2M -delete
  done
done
# timing build: no halo items (wrong results)
D=/tmp/pqb/nohalo; rm -rf $D; mkdir -p $D/music_amd $D/include
cp -r music_amd/csrc $D/music_amd/csrc; rm -rf $D/music_amd/csrc/build; cp include/wavenet_hip.h $D/include/
(cd $D/music_amd/csrc && make -j16 EXTRA="-DPQ_T_NOHALO" > $D/make.log 2>&1) || { echo "build failed" >> $L; tail -5 $D/make.log >> $L; }
for rep in 1 2; do
  echo "== base chain" >> $L; timeout 300 python tools/kbench.py bwd --reps 20 2>/dev/null | tail -1 >> $L
  echo "== NOHALO" >> $L; WAVENET_HIP_LIB=$D/music_amd/libwavenet_hip.so timeout 300 python tools/kbench.py bwd --reps 20 2>/dev/null | tail -1 >> $L
done
cat $L
