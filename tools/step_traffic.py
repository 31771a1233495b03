#!/usr/bin/env python3
"""Real HBM traffic of the config-2 step, kernel by kernel, against the box's measured copy rate.

Inputs (all committed): a rocprofv3 --kernel-trace --stats table of `python bench.py` (Calls, AverageNs), profiles/pmc_kernels.json
(FETCH_SIZE / WRITE_SIZE per launch from separate --pmc passes, corrected as MI355X_MICROARCH.md prescribes) and the bench line of the same
tree (`roofline.measured_copy_GBs`: a device-to-device copy of 1 GiB timed in that process).  Prints a markdown table: launches per step,
average launch, real bytes per launch, the rate they move at, that rate as a share of the copy rate - i.e. how far each kernel is from what
this memory system gives a kernel that does nothing but move its bytes."""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", default="profiles/r05_v1_kernel_stats.csv")
    ap.add_argument("--pmc", default="profiles/pmc_kernels.json")
    ap.add_argument("--bench", default="profiles/r05_v1_bench.json")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.stats)))
    pmc = json.load(open(a.pmc))
    bench = json.load(open(a.bench))
    copy = bench["roofline"]["measured_copy_GBs"] * 1e9
    steps = next(int(r["Calls"]) for r in rows if r["Name"].startswith("softmax256_ce_k"))
    print("| kernel | launches / step | us / launch | real MB / launch | TB/s | of the copy rate (%.2f TB/s) |" % (copy / 1e12))
    print("|---|---|---|---|---|---|")
    tot_b = tot_t = 0.0
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "").strip()
        hit = pmc.get(name)
        if not isinstance(hit, dict) or name.startswith("__amd_rocclr"):      # (the blit kernel: the prefetched H2D copies and the copy-rate probe)
            continue
        n = int(r["Calls"]) / steps
        us = float(r["AverageNs"]) / 1e3
        b = hit["hbm_bytes_per_launch"]
        tot_b += b * n
        tot_t += us * n
        print("| `%s` | %.1f | %.1f | %.1f | %.2f | %.0f %% |" % (name, n, us, b / 1e6, b / us / 1e6, 100 * b / (us * 1e-6) / copy))
    ms = bench["ms_per_step"]
    print("| **whole step** (kernel times overlap on two streams: wall %.3f ms) | | %.0f summed | %.0f per step | %.2f | **%.0f %%** |"
          % (ms, tot_t, tot_b / 1e6, tot_b / (ms * 1e-3) / 1e12, 100 * tot_b / (ms * 1e-3) / copy))


if __name__ == "__main__":
    main()
