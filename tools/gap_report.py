#!/usr/bin/env python3
"""Developer tool: idle time of the GPU inside a training step, from a rocprofv3 --kernel-trace CSV.

    python tools/gap_report.py <kernel_trace.csv> [marker kernel name, default softmax256_ce_k]

Takes the last complete step (marker to marker), merges the busy intervals of all streams and prints the span, the busy
time, the idle time and the largest gaps with the kernels on either side."""
import csv
import sys


def main():
    f = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "softmax256_ce_k"
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44])
            for r in csv.DictReader(open(f))]
    rows.sort()
    idx = [i for i, r in enumerate(rows) if marker in r[2]]
    a, b = idx[-2], idx[-1]
    step = rows[a:b]
    span = (rows[b][0] - step[0][0]) / 1e3
    busy, gaps = 0.0, []
    cur_s, cur_e, last = step[0][0], step[0][1], step[0][2]
    for s, e, n in step[1:] + [rows[b]]:
        if s > cur_e:
            busy += (cur_e - cur_s) / 1e3
            gaps.append(((s - cur_e) / 1e3, last, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        last = n
    busy += (cur_e - cur_s) / 1e3
    print("step span %.1f us, busy (union over streams) %.1f us, idle %.1f us in %d gaps (%.1f us mean)"
          % (span, busy, span - busy, len(gaps), (span - busy) / max(len(gaps), 1)))
    for g, p, n in sorted(gaps, reverse=True)[:12]:
        print("  %6.1f us between %-44s and %s" % (g, p, n))
    ksum = sum((e - s) / 1e3 for s, e, _ in step)
    print("sum of kernel durations %.1f us (overlap on side streams: %.1f us)" % (ksum, ksum - busy))


if __name__ == "__main__":
    main()
