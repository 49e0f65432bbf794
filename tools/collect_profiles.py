#!/usr/bin/env python3
"""Condenses rocprofv3 output dirs (gpurun_out/prof_*) into committed summaries.

  python tools/collect_profiles.py <tag>      e.g. r01_seq
writes profiles/<tag>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats summary, verbatim; from prof_stats/)
       profiles/<tag>_twin_kernel_stats.csv  (the same from prof_stats_twin/, if present)
       profiles/<tag>_counters.csv      (mean of every PMC counter per kernel, one pass per counter set)
and updates profiles/hbm_traffic.json   (per-launch HBM bytes of each psa kernel:
       (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — FETCH_SIZE/WRITE_SIZE are in KiB and gfx950's FETCH_SIZE
       reports half of the fetched bytes, MI355X_MICROARCH.md §HBM; the x2 is calibrated there for
       16 B/lane streams, this kernel reads 4 B/lane, so treat the absolute as +-2x).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out")
    os.makedirs(OUT, exist_ok=True)
    stats = glob.glob(os.path.join(src, "prof_stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(OUT, f"{tag}_kernel_stats.csv"))
    # the default batch (count kernels of consecutive queries overlap on two streams)
    twin = glob.glob(os.path.join(src, "prof_stats_twin", "*", "*_kernel_stats.csv"))
    if twin:
        shutil.copy(twin[0], os.path.join(OUT, f"{tag}_twin_kernel_stats.csv"))
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "prof_*", "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    with open(os.path.join(OUT, f"{tag}_counters.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
        for (k, c), v in sorted(acc.items()):
            if k.startswith("__amd_rocclr"):
                continue
            w.writerow([k, c, len(v), sum(v) / len(v)])
    tpath = os.path.join(OUT, "hbm_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    kernels = {k for (k, c) in acc if c == "FETCH_SIZE" and "psa" in k}
    for k in kernels:
        fetch = acc[(k, "FETCH_SIZE")]
        write = acc.get((k, "WRITE_SIZE"), [0.0])
        short = k.replace("void ", "").replace("psa::", "").split("(")[0]
        traffic[short] = (2 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
