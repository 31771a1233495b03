#!/usr/bin/env python3
"""Developer tool: where does a slow step lose its time?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --steps N`, cuts it into steps at
the adam_k launches, and for every step that took more than 1.15 x the median prints the kernels whose duration exceeds 1.3 x their own
median (with the excess) and the idle gaps (no kernel running on any queue) longer than 20 us.

    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 200 --no-cpu-baseline --no-extras)
    python tools/burst_trace.py out/**/*kernel_trace.csv
"""
import csv, sys, statistics, re
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", r["Kernel_Name"])[:60]) for r in rows))
cuts = [e for (s, e, n) in ev if n.startswith("adam_k")]
steps = []
for a, b in zip(cuts[:-1], cuts[1:]):
    steps.append([x for x in ev if a <= x[0] < b])
durs = [(st[-1][1] - st[0][0]) / 1e6 for st in steps if st]
med = statistics.median(durs)
per = defaultdict(list)
for st in steps:
    for s, e, n in st:
        per[n].append((e - s) / 1e3)
kmed = {n: statistics.median(v) for n, v in per.items()}
print("steps %d, median %.3f ms, slow (> 1.15 x) %d" % (len(durs), med, sum(d > 1.15 * med for d in durs)))
shown = 0
for i, st in enumerate(steps):
    d = (st[-1][1] - st[0][0]) / 1e6
    if d <= 1.15 * med or shown >= 6:
        continue
    shown += 1
    print("\nstep %d: %.3f ms (+%.3f)" % (i, d, d - med))
    slow = [(n, (e - s) / 1e3, kmed[n]) for s, e, n in st if (e - s) / 1e3 > 1.3 * kmed[n] and (e - s) / 1e3 - kmed[n] > 15]
    for n, du, km in slow[:14]:
        print("   %-60s %8.1f us (median %7.1f, +%.1f)" % (n, du, km, du - km))
    # idle gaps: sweep over the union of busy intervals
    busy_end, gaps, prev = st[0][1], [], st[0][2]
    for j, (s, e, n) in enumerate(st[1:], 1):
        if s > busy_end + 20000:
            gaps.append(((s - busy_end) / 1e3, j, prev, n))
        if e > busy_end:
            busy_end, prev = e, n
    for g, j, pn, n in gaps[:8]:
        print("   idle %.1f us between kernel %d (%s) and %d (%s) of %d" % (g, j - 1, pn[:40], j, n[:40], len(st)))
    tot_excess = sum(du - km for n, du, km in [(n, (e - s) / 1e3, kmed[n]) for s, e, n in st])
    print("   sum over all kernels of (duration - median): %.1f us" % tot_excess)
