#!/usr/bin/env python3
"""Condenses rocprofv3 output dirs under gpurun_out/ into committed summaries.

  python tools/collect_profiles.py <tag> [<prefix>]     e.g. r02 prof_r02
reads  gpurun_out/<prefix>_stats/      rocprofv3 --kernel-trace --stats  (bench.py --batch-streams 1)
       gpurun_out/<prefix>_twin/       the same with the default two streams
       gpurun_out/<prefix>_pmc*/       rocprofv3 --pmc ... passes (one counter set each)
writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_twin_kernel_stats.csv (verbatim rocprofv3 summaries),
       profiles/<tag>_counters.csv   (kernel, counter, dispatches, mean_per_dispatch — every (kernel, counter) pair
                                      from its own rows of its own pass),
       profiles/<tag>_meta.json      (sha256 of the kernel sources the passes ran: bench.py checks it),
and updates profiles/hbm_traffic.json (per-launch HBM bytes of each psa kernel:
       (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — FETCH_SIZE/WRITE_SIZE are in KiB and gfx950's FETCH_SIZE
       reports half of the fetched bytes, MI355X_MICROARCH.md §HBM; the x2 is calibrated there for
       16 B/lane streams, these kernels read 4-16 B/lane, so treat the absolute as +-2x).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")


def newest(paths):
    return max(paths, key=os.path.getmtime) if paths else None


def main():
    tag = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "prof_" + tag
    src = os.path.join(ROOT, "gpurun_out")
    os.makedirs(OUT, exist_ok=True)
    # nothing to condense (the profiling call did not run): leave the committed summaries alone
    if not glob.glob(os.path.join(src, prefix + "_stats", "*", "*_kernel_stats.csv")) or \
            not glob.glob(os.path.join(src, prefix + "_pmc*", "*", "*_counter_collection.csv")):
        sys.exit(f"collect_profiles: no rocprofv3 output under gpurun_out/{prefix}_*: profiles/ left as it is")
    for sub, name in (("_stats", f"{tag}_kernel_stats.csv"), ("_twin", f"{tag}_twin_kernel_stats.csv")):
        stats = newest(glob.glob(os.path.join(src, prefix + sub, "*", "*_kernel_stats.csv")))
        if stats:
            shutil.copy(stats, os.path.join(OUT, name))
    acc = collections.defaultdict(list)
    # gpurun merges every call's output into gpurun_out/, so a pass directory holds one file per profiling run of the
    # round: only the NEWEST run of each pass counts (earlier ones ran earlier kernels)
    for d in sorted(glob.glob(os.path.join(src, prefix + "_pmc*"))):
        f = newest(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))
        if not f:
            continue
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    with open(os.path.join(OUT, f"{tag}_counters.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
        for (k, c), v in sorted(acc.items()):
            if k.startswith("__amd_rocclr") or "psa" not in k:
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
    # what these summaries are evidence for: the kernel sources they were collected from (bench.py withholds
    # roofline.frac when the tree's hash differs), and the per-kernel resources of that build
    sys.path.insert(0, ROOT)
    import bench
    meta = {"csrc_sha256": bench.csrc_sha256(), "collected": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
            "commands": "tools/profile_round.sh " + tag}
    json.dump(meta, open(os.path.join(OUT, f"{tag}_meta.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
