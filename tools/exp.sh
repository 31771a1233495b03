mkdir -p gpurun_out
REPO=$(pwd)
export TMPDIR=/tmp
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/prof.log 2>&1)
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ","")[:44]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 150:
        print(k, "n=%d avg=%.1f us total/step=%.0f us" % (len(v), sum(v)/len(v), sum(v)/7))
PY
find gpurun_out/prof -type f -size +3M -delete
