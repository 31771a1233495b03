#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) into a small markdown table.

usage: python tools/prof_summary.py gpurun_out > profiles/rNN_summary.md
Looks for */*kernel_stats.csv, */*kernel_trace.csv and */*counter_collection.csv below
<dir>/prof, <dir>/pmc_fetch, <dir>/pmc_write.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    root = sys.argv[1]
    print("# rocprofv3 summary (%s)\n" % root)
    stats = find(os.path.join(root, "prof"), "*kernel_stats.csv")
    if stats:
        print("## kernel time (rocprofv3 --kernel-trace --stats)\n")
        print("| kernel | calls | total ms | avg us | % |")
        print("|---|---|---|---|---|")
        rows = list(csv.DictReader(open(stats[0])))
        for r in rows[:25]:
            print("| %s | %s | %.3f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
        print()
    pmc = {}
    launches = {}
    # steps of the profiled command: the PMC passes run `bench.py --steps 2 --warmup 1 --settle 0` + its 3 phase-table and 3
    # per-kernel steps (tools/gpu_check.sh); PMC_STEPS overrides.  launches per step = launches / steps.
    n_steps = float(os.environ.get("PMC_STEPS", "9"))
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = find(os.path.join(root, tag), "*counter_collection.csv")
        if not files:
            continue
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(files[0])):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
        print("## %s per launch (raw counter units = KiB; gfx950: double FETCH_SIZE for wide streaming reads)\n" % counter)
        print("| kernel | launches | avg %s (KiB) |" % counter)
        print("|---|---|---|")
        for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:20]:
            print("| %s | %d | %.1f |" % (k, n, v / n))
            pmc.setdefault(k, {})[counter] = v / n
            launches[k] = n
        print()
    # HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950:
    # FETCH_SIZE (KiB) counts 128-B wide reads at 64 B -> x2; WRITE_SIZE (KiB) is exact.
    import json
    allk = {k: {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                "launches_per_step": round(launches.get(k, 0) / n_steps, 3)}
            for k, v in pmc.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    # config 4 (tools/ae_phases.py under the same two PMC passes): entries "c4:<kernel>" with a "source" (bench.py's config-2 figures skip
    # them; its extra.c4_autoencoder.roofline_stacks.traffic_per_launch reads them)
    c4 = {}
    for tag, counter in (("pmc_fetch_ae", "FETCH_SIZE"), ("pmc_write_ae", "WRITE_SIZE")):
        files = find(os.path.join(root, tag), "*counter_collection.csv")
        if not files:
            continue
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(files[0])):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
        print("## config 4: %s per launch (KiB)\n" % counter)
        print("| kernel | launches | avg %s (KiB) |" % counter)
        print("|---|---|---|")
        for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
            print("| %s | %d | %.1f |" % (k, n, v / n))
            c4.setdefault(k, {})[counter] = v / n
            c4[k]["launches"] = n
        print()
    for k, v in c4.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            allk["c4:" + k] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                               "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024, "launches": v["launches"],
                               "source": "config 4 step (tools/ae_phases.py), separate --pmc passes"}
    if allk:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import csrc_sha
        allk["_csrc_sha16"] = csrc_sha()                     # the kernel sources these figures belong to (bench.py checks it)
        # bench.py reads this one (copied to profiles/pmc_kernels.json): HBM bytes per launch of every kernel
        json.dump(allk, open(os.path.join(root, "pmc_kernels.json"), "w"), indent=1)
    for k, v in pmc.items():
        if k.startswith("resblock_fwd") and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            import json
            rec = {"kernel": k, "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                   "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                   "note": "separate --pmc passes; reads doubled (gfx950 FETCH_SIZE counts 128-B requests at 64 B)"}
            json.dump(rec, open(os.path.join(root, "pmc_resblock_fwd.json"), "w"), indent=1)
            print("## traffic\n\n`%s`" % json.dumps(rec))


if __name__ == "__main__":
    main()
